#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06j; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_graph.py -q -x > $O/units.log 2>&1; echo "units rc=$?"; tail -4 $O/units.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "pmf_r34-shipped or epmf or masked" > $O/full.log 2>&1; echo "full rc=$?"; grep "^\[fullsize\|^\.\[fullsize\|^\[masked\|^\.\[masked" $O/full.log | cut -c1-250; tail -3 $O/full.log
q="--steps 60 --warmup 10 --no-cpu-baseline --no-f32-ref --no-parity"
for i in 1 2; do
python bench.py $q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r34', round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['families']['conv_fwd']['ms'], d['roofline']['families']['conv_dgrad']['ms'], d['roofline']['families']['conv_wgrad']['ms'])"
done
python bench.py $q --model epmf 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('epmf', round(d['ms_per_step'],3), d['roofline']['frac'])"
python bench.py $q --backbone resnet50 --nclasses 17 --height 32 --width 1024 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r50', round(d['ms_per_step'],3), d['roofline']['frac'])"
python bench.py --backbone resnet50 --nclasses 17 --height 480 --width 640 --steps 30 --warmup 5 --no-cpu-baseline --no-f32-ref 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['parity']; print('r50sb', round(d['ms_per_step'],3), d['roofline']['frac'], p['ok'], p['logits_rel'], p['logits_rel_vs_float64'])"
