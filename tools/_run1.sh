mkdir -p gpurun_out/r04a
( time python -m pytest tests/test_gpu_fullsize.py -x -q -s ) > gpurun_out/r04a/fullsize.txt 2>&1
python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
python bench.py --mode infer --steps 50 > gpurun_out/r04a/infer.json 2> gpurun_out/r04a/infer.err
python bench.py --force-dist --steps 40 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity > gpurun_out/r04a/forcedist.json 2> gpurun_out/r04a/forcedist.err
PMF_DP_CUTS=flops python bench.py --force-dist --steps 40 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity > gpurun_out/r04a/forcedist_flops.json 2> gpurun_out/r04a/forcedist_flops.err
tail -5 gpurun_out/r04a/fullsize.txt
