mkdir -p gpurun_out/r04d
( time python -m pytest tests/test_gpu_boundary.py -x -q -k "nus" ) > gpurun_out/r04d/nus.txt 2>&1
( time python -m pytest tests/test_gpu_ops.py -x -q -k "unit" ) > gpurun_out/r04d/ops.txt 2>&1
python tools/bench_elem.py 2>&1 | grep "ONE LAUNCH" > gpurun_out/r04d/bench_elem.txt
python bench.py --no-cpu-baseline --no-f32-ref --no-parity > gpurun_out/r04d/bench.json 2> gpurun_out/r04d/bench.err
tail -n 4 gpurun_out/r04d/nus.txt gpurun_out/r04d/ops.txt; cat gpurun_out/r04d/bench_elem.txt
