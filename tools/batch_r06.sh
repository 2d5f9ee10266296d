#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06h; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -q -k "splitk or tile_configs or c1x1s2 or split_bf16 or unit_fwd_bwd or train_step or epmf or r50" > $O/units.log 2>&1; echo "units rc=$?"; tail -6 $O/units.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "r50" > $O/r50.log 2>&1; echo "r50 rc=$?"; grep "^\[fullsize\|^\.\[fullsize" $O/r50.log | cut -c1-250; tail -3 $O/r50.log
q="--steps 60 --warmup 10 --no-cpu-baseline --no-f32-ref --no-parity"
for f in 0 1 0 1; do
  PMF_SPLITK_FUSED=$f python bench.py $q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r34 fused=$f', round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['families']['conv_fwd']['ms'], d['roofline']['families']['conv_dgrad']['ms'])"
done
for f in 0 1; do
  PMF_SPLITK_FUSED=$f python bench.py $q --backbone resnet50 --nclasses 17 --height 32 --width 1024 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r50 fused=$f', round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['families']['conv_fwd']['ms'], d['roofline']['families']['conv_dgrad']['ms'])"
  PMF_SPLITK_FUSED=$f python bench.py $q --model epmf 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('epmf fused=$f', round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['families']['conv_fwd']['ms'], d['roofline']['families']['conv_dgrad']['ms'])"
done
