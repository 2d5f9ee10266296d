#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -s -k "masked" > $O/masked.log 2>&1; echo "masked rc=$?"
grep "^\[masked" $O/masked.log; tail -3 $O/masked.log
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "wave_scheduled or tile_configs" > $O/ws.log 2>&1; echo "ws rc=$?"; tail -5 $O/ws.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -s -k "r50_sb-heuristic or S_G" > $O/r50sb.log 2>&1; echo "r50sb rc=$?"; grep "^\[fullsize" $O/r50sb.log; tail -5 $O/r50sb.log
PMF_TUNE_CACHE=$O/tuned_r50sb.txt PMF_TUNE_REPS=20 timeout 900 python bench.py --backbone resnet50 --nclasses 17 --height 480 --width 640 --steps 30 --warmup 5 > $O/bench_r50_sb.json 2> $O/bench_r50_sb.err; echo "bench r50sb rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_r50_sb.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"].get("frac"), d["parity"]["ok"])
except Exception as e:
    print("no line", e)
PY
