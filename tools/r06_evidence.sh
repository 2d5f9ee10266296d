#!/bin/bash
# round-6 evidence batch (GPU box, `gpurun -- "bash tools/r06_evidence.sh <commit>"`): smoke, rocprofv3 summaries (tools/collect_profiles.sh),
# bench lines of every configuration -> gpurun_out/r06z + gpurun_out/r06 (copy to profiles/r06_*)
o=gpurun_out/r06z; mkdir -p $o
python __graft_entry__.py smoke 2>&1 | tail -2
if [ -z "$SKIP_PROFILES" ]; then bash tools/collect_profiles.sh r06 $1 > $o/collect.log 2>&1; head -1 gpurun_out/r06/r06_per_layer_kernel_times.txt; fi
L=$o/r06_bench_lines.jsonl; : > $L
python bench.py --steps 20 --warmup 5 --parity-masked >> $L 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline >> $L 2>/dev/null
python bench.py --height 480 --width 640 --steps 30 --warmup 5 --no-cpu-baseline >> $L 2>/dev/null
python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline >> $L 2>/dev/null
python bench.py --mode infer --height 480 --width 640 --steps 30 --warmup 5 --no-cpu-baseline >> $L 2>/dev/null
python bench.py --backbone resnet50 --nclasses 17 --height 32 --width 1024 --steps 50 --warmup 10 --no-cpu-baseline --parity-masked >> $L 2>/dev/null
python bench.py --backbone resnet50 --nclasses 17 --height 480 --width 640 --steps 30 --warmup 5 --no-cpu-baseline --parity-masked >> $L 2>/dev/null
python bench.py --backbone resnet50 --nclasses 17 --height 512 --width 640 --mode infer --steps 30 --warmup 5 --no-cpu-baseline >> $L 2>/dev/null
python bench.py --model epmf --steps 50 --warmup 10 --no-cpu-baseline --parity-masked >> $L 2>/dev/null
python bench.py --model epmf --height 320 --width 1280 --steps 40 --warmup 8 --no-cpu-baseline --parity-masked >> $L 2>/dev/null   # EPMF's native size (tasks/epmf/config_server_kitti.yaml)
python bench.py --model salsanext --steps 50 --warmup 10 --no-cpu-baseline >> $L 2>/dev/null
python bench.py --height 256 --width 1024 --steps 30 --warmup 5 --no-cpu-baseline >> $L 2>/dev/null
python bench.py --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-f32-ref >> $L 2>/dev/null
python bench.py --mode loader >> $L 2>/dev/null
python -c "
import json
for l in open('$L'):
    l=l.strip()
    if not l.startswith('{'): print('NON-JSON:', l[:80]); continue
    d=json.loads(l); m=((d.get('parity') or {}).get('masked') or {}); print(round(d['value'],2), d['unit'], round(d.get('ms_per_step') or 0,3), (d.get('roofline') or {}).get('frac'), (d.get('parity') or {}).get('ok'), m.get('n_bad'), d['config']['workload'][:70])"
