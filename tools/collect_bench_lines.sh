#!/bin/bash
# Run ON the GPU box (gpurun -- 'bash tools/collect_bench_lines.sh r04'): the bench JSON line of every mode, one per line,
# into gpurun_out/<tag>_bench_lines.jsonl (copy to profiles/).
TAG=${1:-r04}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${TAG}_bench_lines.jsonl
: > $OUT
cd $ROOT
run() { timeout 900 python bench.py "$@" 2> /dev/null | tail -1 >> $OUT; }
run                                                   # headline (parity block, roofline, cpu_baseline, fp32_mfma_only)
run --mode infer
run --model epmf
run --backbone resnet50 --nclasses 17 --height 32 --width 1024
run --mode loader
run --model salsanext
run --height 256 --width 1024
wc -l $OUT
python - <<PY
import json
for l in open("$OUT"):
    d = json.loads(l)
    print(d["config"].get("workload", "?")[:70], "|", round(d["value"], 2), d["unit"], "| ms", d.get("ms_per_step"),
          "| roofline", (d.get("roofline") or {}).get("frac"), ((d.get("roofline") or {}).get("bf16_pipe") or {}).get("frac"),
          "| cpu", (d.get("cpu_baseline") or {}).get("value"), "| parity", (d.get("parity") or {}).get("ok"))
PY
