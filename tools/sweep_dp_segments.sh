#!/bin/bash
# Run ON an N-GPU box: `bash tools/sweep_dp_segments.sh 8` -- the headline bench at N ranks in the default data-parallel form
# (PMF_DP_MODE=events: one backward range, all-reduces hung behind plan events) and, for A/B, with the backward plan cut into
# PMF_DP_SEGMENTS = 1..6 ranges (PMF_DP_MODE=segments), one JSON line each with `data_parallel.exposed_allreduce_ms_per_step`.
# On a 1-GPU box `bash tools/sweep_dp_segments.sh 1` runs the same sweep through --force-dist (segmentation cost only).
N=${1:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/dp_sweep
mkdir -p $OUT
EXTRA=""; [ "$N" = "1" ] && EXTRA="--force-dist"
for k in 1 2 3 4 6; do
  PMF_DP_MODE=segments PMF_DP_SEGMENTS=$k python $ROOT/bench.py --gpus $N $EXTRA --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity \
    2> $OUT/seg$k.err | tail -1 > $OUT/seg$k.json
  python - $OUT/seg$k.json $k <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
dp = d.get("data_parallel") or {}
print("segments %s: %.2f it/s  %.3f ms/step  exposed all-reduce %s ms" % (sys.argv[2], d["value"], d["ms_per_step"],
      (dp.get("exposed_allreduce_ms_per_step") or {}).get("max")))
PY
done
python $ROOT/bench.py --gpus $N $EXTRA --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity \
  2> $OUT/events.err | tail -1 > $OUT/events.json
python -c "import json;d=json.load(open('$OUT/events.json'));print('events (default):', d['value'], 'it/s', d['ms_per_step'], 'ms/step', (d.get('data_parallel') or {}).get('exposed_allreduce_ms_per_step'))"
