mkdir -p gpurun_out/r04b
( time python -m pytest tests/test_gpu_fullsize.py -q -s ) > gpurun_out/r04b/fullsize.txt 2>&1
( time python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py ) > gpurun_out/r04b/gpu_tests.txt 2>&1
python bench.py > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err
python bench.py --mode infer --steps 50 > gpurun_out/r04b/infer.json 2> gpurun_out/r04b/infer.err
python bench.py --force-dist --steps 40 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity 2> gpurun_out/r04b/forcedist.err | tail -1 > gpurun_out/r04b/forcedist.json
PMF_DP_MODE=segments python bench.py --force-dist --steps 40 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity 2> gpurun_out/r04b/forcedist_seg.err | tail -1 > gpurun_out/r04b/forcedist_seg.json
tail -3 gpurun_out/r04b/fullsize.txt gpurun_out/r04b/gpu_tests.txt
