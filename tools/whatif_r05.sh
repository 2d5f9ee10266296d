#!/bin/bash
o=gpurun_out/r05x; mkdir -p $o
q="--steps 60 --warmup 10 --no-parity --no-cpu-baseline --no-roofline --no-f32-ref"
run() { echo -n "$1: " >> $o/ab.txt; env $2 timeout 200 python bench.py $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $o/ab.txt; }
for rep in 1 2; do
run "prio 0" "A=1"
run "conv prio 1" "PMF_CONV_PRIO=1"
run "conv prio 3" "PMF_CONV_PRIO=3"
done
cat $o/ab.txt
