mkdir -p gpurun_out/r04f
( time python -m pytest tests/test_gpu_ops.py -x -q -k "tile_configs and direct" ) > gpurun_out/r04f/ops.txt 2>&1
( time python -m pytest tests/test_gpu_parity.py -x -q -k "split_bf16_vs_float64" ) > gpurun_out/r04f/s3.txt 2>&1
tail -n 4 gpurun_out/r04f/ops.txt gpurun_out/r04f/s3.txt
D=$((1<<24))
S3_CFGS="0,$((32|256|65536|D)),$((32|512|65536|D)),$((64|256|65536|D)),$((64|512|65536|D))" python tools/bench_conv.py s3 3x3 > gpurun_out/r04f/conv_3x3.txt 2>&1
S3_CFGS="0,$((32|256|65536|D)),$((32|512|65536|D)),$((64|256|65536|D)),$((64|512|65536|D))" python tools/bench_conv.py s3 2x2 > gpurun_out/r04f/conv_2x2.txt 2>&1
cat gpurun_out/r04f/conv_2x2.txt gpurun_out/r04f/conv_3x3.txt
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity"
PMF_TUNE_DIRECT=0 $B 2> gpurun_out/r04f/b0.err | tail -1 > gpurun_out/r04f/b0.json
$B 2> gpurun_out/r04f/b1.err | tail -1 > gpurun_out/r04f/b1.json
PMF_TUNE_DIRECT=0 $B 2> /dev/null | tail -1 > gpurun_out/r04f/b0b.json
$B 2> /dev/null | tail -1 > gpurun_out/r04f/b1b.json
python -c "
import json
for n in ('b0','b1','b0b','b1b'):
    d=json.load(open('gpurun_out/r04f/%s.json'%n)); print(n, d['value'], d['ms_per_step'])"
