#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06o; mkdir -p $O
export TMPDIR=/tmp
q="--steps 80 --warmup 10 --no-cpu-baseline --no-f32-ref --no-parity --no-roofline"
for i in 1 2 3; do
for t in 1 0; do
  if [ $t = 1 ]; then export PMF_TIE_LIST=1; else unset PMF_TIE_LIST; fi
  python bench.py $q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r34 tie_list=$t', round(d['ms_per_step'],3))"
done; done
unset PMF_TIE_LIST
timeout 600 python -m pytest tests/test_gpu_graph.py -q 2>&1 | tail -2
