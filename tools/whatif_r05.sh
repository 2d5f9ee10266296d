#!/bin/bash
o=gpurun_out/r05w; mkdir -p $o
q="--steps 60 --warmup 10 --no-parity --no-cpu-baseline --no-roofline --no-f32-ref"
run() { echo -n "$1: " >> $o/ab.txt; env $2 timeout 200 python bench.py $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $o/ab.txt; }
for rep in 1 2 3; do
run "default (batch 4)" "A=1"
run "wgrad batch 2" "PMF_WGRAD_BATCH=2"
run "wgrad batch 3" "PMF_WGRAD_BATCH=3"
done
run "red batch 16" "PMF_RED_BATCH=16"
run "pack early 4" "PMF_PACK_EARLY=4"
run "pack early 16" "PMF_PACK_EARLY=16"
run "default (batch 4)" "A=1"
cat $o/ab.txt
