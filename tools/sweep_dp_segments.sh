#!/bin/bash
# Run ON an N-GPU box: `bash tools/sweep_dp_segments.sh 8` -- the headline bench at N ranks for several numbers of backward
# segments (PMF_DP_SEGMENTS; the gradient ranges that are final after a segment are all-reduced under the next ones) and
# for the round-3 cut rule (PMF_DP_CUTS=flops), one JSON line each with `data_parallel.exposed_allreduce_ms_per_step`.
# On a 1-GPU box `bash tools/sweep_dp_segments.sh 1` runs the same sweep through --force-dist (segmentation cost only).
N=${1:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/dp_sweep
mkdir -p $OUT
EXTRA=""; [ "$N" = "1" ] && EXTRA="--force-dist"
for k in 1 2 3 4 6; do
  PMF_DP_SEGMENTS=$k python $ROOT/bench.py --gpus $N $EXTRA --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity \
    2> $OUT/seg$k.err | tail -1 > $OUT/seg$k.json
  python - $OUT/seg$k.json $k <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
dp = d.get("data_parallel") or {}
print("segments %s: %.2f it/s  %.3f ms/step  exposed all-reduce %s ms" % (sys.argv[2], d["value"], d["ms_per_step"],
      (dp.get("exposed_allreduce_ms_per_step") or {}).get("max")))
PY
done
PMF_DP_CUTS=flops python $ROOT/bench.py --gpus $N $EXTRA --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity \
  2> $OUT/flops4.err | tail -1 > $OUT/flops4.json
python -c "import json;d=json.load(open('$OUT/flops4.json'));print('4 equal-flop segments (round 3):', d['value'], 'it/s', (d.get('data_parallel') or {}).get('exposed_allreduce_ms_per_step'))"
