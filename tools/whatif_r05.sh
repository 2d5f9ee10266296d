#!/bin/bash
o=gpurun_out/r05v; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $o/pytest_ops.log 2>&1; tail -2 $o/pytest_ops.log
timeout 300 python bench.py --model epmf --steps 30 --warmup 8 --no-cpu-baseline --no-f32-ref --profile-out $o/ops_epmf.txt > $o/bench_epmf.json 2>$o/err.txt
python -c "
import json
d=json.loads(open('$o/bench_epmf.json').read().strip().splitlines()[-1]); print('epmf', d['ms_per_step'], d['parity']['ok'], d['roofline']['frac'], {k:(v['ms'],v['achieved']) for k,v in d['roofline']['families'].items()})"
grep "downCntx.*OP_WGRAD\|OP_WGRAD_PART.*downCntx" $o/ops_epmf.txt | cut -c1-140
