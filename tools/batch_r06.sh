#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06s; mkdir -p $O
export TMPDIR=/tmp
PMF_TUNE_CACHE=$O/tuned_epmf_native.txt PMF_TUNE_REPS=20 python bench.py --model epmf --height 320 --width 1280 --steps 40 --warmup 8 --no-cpu-baseline --parity-masked > $O/epmf_320x1280.json 2> $O/epmf_320x1280.err; echo "rc=$?"
python - <<PY
import json
d=json.loads(open("$O/epmf_320x1280.json").read().strip().splitlines()[-1]); p=d["parity"]; m=p["masked"]
print(round(d["value"],2), round(d["ms_per_step"],3), d["roofline"]["frac"], p["ok"], m["n_bad"], m["ratio_max"])
PY
wc -l $O/tuned_epmf_native.txt
