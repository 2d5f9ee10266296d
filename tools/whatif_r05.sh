#!/bin/bash
# round-5 measurement batch (GPU box)
o=gpurun_out/r05f; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $o/pytest_ops.log 2>&1; tail -3 $o/pytest_ops.log
timeout 300 python bench.py --height 480 --width 640 --steps 10 --warmup 4 --no-cpu-baseline --no-f32-ref --profile-out $o/ops_sb.txt > $o/bench_sb.json 2>$o/bench_sb.err
timeout 200 python bench.py --steps 10 --warmup 4 --no-parity --no-cpu-baseline --no-f32-ref --profile-out $o/ops.txt > $o/bench.json 2>$o/bench.err
timeout 200 python bench.py --backbone resnet50 --nclasses 17 --height 32 --width 1024 --steps 10 --warmup 4 --no-parity --no-cpu-baseline --no-f32-ref --profile-out $o/ops_r50.txt > $o/bench_r50.json 2>$o/bench_r50.err
timeout 200 python bench.py --model epmf --steps 10 --warmup 4 --no-parity --no-cpu-baseline --no-f32-ref --profile-out $o/ops_epmf.txt > $o/bench_epmf.json 2>$o/bench_epmf.err
q="--steps 40 --warmup 8 --no-parity --no-cpu-baseline --no-roofline --no-f32-ref"
for rep in 1 2; do
echo -n "new: " >> $o/ab.txt
timeout 200 python bench.py $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $o/ab.txt
echo -n "round-4-like weight-gradient tails (no stream / ragged / low-res direct): " >> $o/ab.txt
PMF_WGRAD_STREAM=0 PMF_WG_S3N_NORAGGED=1 PMF_WGRAD_DIRECT_S3_MIN_PIX=16384 timeout 200 python bench.py $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $o/ab.txt
echo -n "N-split kernel off too (PMF_WG_S3N=0): " >> $o/ab.txt
PMF_WG_S3N=0 PMF_WGRAD_STREAM=0 PMF_WGRAD_DIRECT_S3_MIN_PIX=16384 timeout 200 python bench.py $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $o/ab.txt
done
cat $o/ab.txt
for f in bench bench_sb bench_r50 bench_epmf; do python -c "
import json,sys
d=json.loads(open('$o/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['frac'], {k:(v['ms'],v['achieved']) for k,v in d['roofline']['families'].items()})"; done
