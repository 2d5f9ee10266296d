#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
bash tools/make_tune_cache.sh $O/tuned_gfx950.txt > $O/tune.log 2>&1; echo "tune rc=$?"; tail -2 $O/tune.log
cp $O/tuned_gfx950.txt pmf_amd/tuned/gfx950.txt
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -15 $O/gpu_tests.log
grep "^\[masked\|^\.\[masked\|^\[fullsize\|^\.\[fullsize" $O/gpu_tests.log
timeout 600 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench_headline.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("bf16_pipe_frac"), d["parity"]["ok"], d["roofline"]["in_step"]["frac"])
PY
