#!/bin/bash
# Run ON the GPU box (gpurun -- 'bash tools/collect_profiles.sh r04 <commit>'): collects the round's rocprofv3 evidence into
# gpurun_out/<tag>/ ; copy the summaries from there into profiles/ (tracked).
TAG=${1:-r04}
COMMIT=${2:-unknown}      # the GPU box has no .git: pass `git rev-parse --short HEAD` from the build container
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export PMF_TUNE_CACHE=/tmp/pmf_tune.txt        # same tile configurations in every pass
python $ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity > $OUT/bench_plain.json 2> /dev/null
# 1. per-kernel statistics of the default bench command
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python $ROOT/bench.py --no-cpu-baseline --no-f32-ref > $OUT/bench_under_rocprof.json 2> /dev/null
python $ROOT/tools/rocpd_stats.py /tmp/p1/r_results.db $OUT/${TAG}_bench_kernel_stats.csv 0.5 > /dev/null
# 2. per-layer kernel durations (one lane: launch order = op order)
rm -rf /tmp/p2; PMF_LANES=0 rocprofv3 --kernel-trace -d /tmp/p2 -o j -- python $ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-f32-ref --no-parity --profile-out $OUT/ops.txt > /dev/null 2>&1
python $ROOT/tools/rocpd_join.py /tmp/p2/j_results.db $OUT/ops.txt 14 $OUT/${TAG}_per_layer_kernel_times.txt
# (the same one-lane run as per-kernel statistics: every kernel alone on the chip -- what bench.py's roofline divides by)
python $ROOT/tools/rocpd_stats.py /tmp/p2/j_results.db $OUT/${TAG}_bench_kernel_stats_one_lane.csv 0.5 > /dev/null
# 2b. who runs next to whom in the replayed step: per-queue timeline of the last full step + concurrency histogram
rm -rf /tmp/p2b; rocprofv3 --kernel-trace -d /tmp/p2b -o l -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-f32-ref --no-roofline --no-parity > /dev/null 2>&1
python $ROOT/tools/rocpd_lanes.py /tmp/p2b/l_results.db $OUT/${TAG}_lanes_segments.txt
# 3. HBM-side traffic of the conv launches: FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c; rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o t -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity > /dev/null 2>&1
done
python $ROOT/tools/pmc_traffic.py /tmp/p_FETCH_SIZE/t_results.db /tmp/p_WRITE_SIZE/t_results.db $OUT/${TAG}_pmc_traffic.json > /dev/null
python - $OUT/${TAG}_pmc_traffic.json $COMMIT <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); d["commit"] = sys.argv[2]; json.dump(d, open(sys.argv[1], "w"), indent=1)
PY
# 4. matrix-pipe utilisation of the conv kernels
rm -rf /tmp/p4; rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_F32 -d /tmp/p4 -o m -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/p4/m_results.db conv > $OUT/${TAG}_pmc_mfma.txt 2>&1
# 5. the roofline of the TIMED step: four-lane trace (2b) + the counter passes (3, 4) over every kernel of one step
python $ROOT/tools/in_step.py /tmp/p2b/l_results.db /tmp/p4/m_results.db /tmp/p_FETCH_SIZE/t_results.db /tmp/p_WRITE_SIZE/t_results.db \
  $OUT/${TAG}_in_step.json $COMMIT $OUT/bench_plain.json > /dev/null 2> $OUT/in_step.err
# 6. per-layer kernel durations of the other BASELINE configurations (one lane): configs[3] PMF-ResNet50 32x1024, configs[4] EPMF
for cfgname in r50 epmf; do
  if [ $cfgname = r50 ]; then ARGS="--backbone resnet50 --nclasses 17 --height 32 --width 1024"; else ARGS="--model epmf"; fi
  rm -rf /tmp/p6; PMF_LANES=0 rocprofv3 --kernel-trace -d /tmp/p6 -o j -- python $ROOT/bench.py $ARGS --steps 10 --warmup 5 --no-cpu-baseline --no-f32-ref --no-parity --profile-out $OUT/ops_$cfgname.txt > /dev/null 2>&1
  python $ROOT/tools/rocpd_join.py /tmp/p6/j_results.db $OUT/ops_$cfgname.txt 14 $OUT/${TAG}_per_layer_kernel_times_$cfgname.txt
done
ls -la $OUT
