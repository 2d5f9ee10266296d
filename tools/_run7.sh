mkdir -p gpurun_out/r04g
( time python -m pytest tests/test_gpu_parity.py -x -q -k "split_bf16_vs_float64 or conv_fwd_vs_torch or pack_split" ) > gpurun_out/r04g/s3.txt 2>&1
( time python -m pytest tests/test_gpu_ops.py -x -q -k "c7x7 or tile_configs" ) > gpurun_out/r04g/ops.txt 2>&1
tail -n 4 gpurun_out/r04g/s3.txt gpurun_out/r04g/ops.txt
S3_CFGS="0,$((32|256|65536)),$((32|512|65536)),$((64|256|65536)),$((64|512|65536))" python tools/bench_conv.py s3 stem 2>&1 | grep -v amdgpu.ids
( time python -m pytest tests/test_gpu_parity.py -x -q -k "eval_forward or train_step_matches or precision_probe" ) > gpurun_out/r04g/parity.txt 2>&1
tail -n 4 gpurun_out/r04g/parity.txt
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity"
PMF_NO_STEM_DIRECT=1 $B 2> /dev/null | tail -1 > gpurun_out/r04g/b0.json
$B 2> gpurun_out/r04g/b1.err | tail -1 > gpurun_out/r04g/b1.json
PMF_NO_STEM_DIRECT=1 $B 2> /dev/null | tail -1 > gpurun_out/r04g/b0b.json
$B 2> /dev/null | tail -1 > gpurun_out/r04g/b1b.json
python -c "
import json
for n in ('b0','b1','b0b','b1b'):
    d=json.load(open('gpurun_out/r04g/%s.json'%n)); print(n, d['value'], d['ms_per_step'])"
