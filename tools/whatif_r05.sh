#!/bin/bash
# round-5 measurement batch (GPU box): new weight-gradient cases, full-size parity with the shipped table, what-if step times
# (PMF_DUP_OPS issues launches twice: honest data, honest clocks), per-op profile
o=gpurun_out/r05e; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $o/pytest_ops.log 2>&1; tail -3 $o/pytest_ops.log
timeout 200 python bench.py --steps 10 --warmup 4 --no-parity --no-cpu-baseline --no-f32-ref --profile-out $o/ops.txt > $o/bench.json 2>$o/bench.err
q="--steps 30 --warmup 8 --no-parity --no-cpu-baseline --no-roofline --no-f32-ref"
for s in 0 4 36 3 35 6 7; do
  echo -n "dup $s: " >> $o/whatif.txt
  PMF_DUP_OPS=$s timeout 200 python bench.py $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $o/whatif.txt
done
echo -n "pack in the forward prologue: " >> $o/whatif.txt
PMF_PACK_BEHIND_OPTIM=0 timeout 200 python bench.py $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $o/whatif.txt
echo -n "no stream kernel: " >> $o/whatif.txt
PMF_WGRAD_STREAM=0 timeout 200 python bench.py $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $o/whatif.txt
cat $o/whatif.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu > $o/pytest_fullsize.log 2>&1; tail -5 $o/pytest_fullsize.log
