#!/bin/bash
o=gpurun_out/r05j; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $o/pytest_ops.log 2>&1; tail -3 $o/pytest_ops.log
(time tools/make_tune_cache.sh $o/tuned.txt) > $o/make.log 2>&1; tail -4 $o/make.log
python - <<'P'
import ast
d=ast.literal_eval(open('gpurun_out/r05j/tuned.txt').read())
print(len(d), 'shapes;', sum(1 for v in d.values() if v>>25&1), 'on the wave-scheduled kernel')
P
q="--steps 40 --warmup 8 --no-parity --no-cpu-baseline --no-roofline --no-f32-ref"
for rep in 1 2; do
echo -n "shipped table (no WS): " >> $o/ab.txt; timeout 200 python bench.py $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $o/ab.txt
echo -n "new table (WS candidates): " >> $o/ab.txt; PMF_TUNE_CACHE=$o/tuned.txt timeout 200 python bench.py $q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $o/ab.txt
done
cat $o/ab.txt
PMF_TUNE_CACHE=$o/tuned.txt timeout 200 python bench.py --steps 10 --warmup 4 --no-parity --no-cpu-baseline --no-f32-ref --profile-out $o/ops.txt > $o/bench.json 2>$o/bench.err
python -c "
import json
d=json.loads(open('$o/bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['roofline']['frac'], {k:(v['ms'],v['achieved']) for k,v in d['roofline']['families'].items()})"
cp $o/tuned.txt $o/tuned_fixed.txt
for kind in epmf pmf_r34 r50 pmf_r34_sb; do
  PMF_TUNE_CACHE=$o/tuned_fixed.txt timeout 500 python tools/bisect_tune.py --kind $kind --fix $o/tuned_fixed.txt > $o/fix_$kind.log 2>&1
  tail -2 $o/fix_$kind.log
done
PMF_TUNE_CACHE=$o/tuned_fixed.txt timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu > $o/pytest_fullsize.log 2>&1; tail -5 $o/pytest_fullsize.log
