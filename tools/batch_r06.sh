#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06i; mkdir -p $O
export TMPDIR=/tmp
bash tools/make_tune_cache.sh $O/tuned_gfx950.txt > $O/tune.log 2>&1; echo "tune rc=$?"; tail -2 $O/tune.log
cp $O/tuned_gfx950.txt pmf_amd/tuned/gfx950.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -22 $O/gpu_tests.log | cut -c1-200
q="--steps 60 --warmup 10 --no-cpu-baseline --no-f32-ref --no-parity"
python bench.py $q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r34', round(d['ms_per_step'],3), d['roofline']['frac'])"
