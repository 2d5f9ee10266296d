mkdir -p gpurun_out/r04c
python tools/bench_elem.py > gpurun_out/r04c/bench_elem.txt 2>&1
( time python -m pytest tests/test_gpu_ops.py -x -q ) > gpurun_out/r04c/ops.txt 2>&1
( time python -m pytest tests/test_gpu_parity.py -x -q -k "train_step or wholenet or flat_training or epmf or r50 or data_parallel" ) > gpurun_out/r04c/parity.txt 2>&1
python bench.py > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err
PMF_BN_SMALL=0 python bench.py --no-cpu-baseline --no-roofline --no-f32-ref --no-parity > gpurun_out/r04c/bench_nosmall.json 2> gpurun_out/r04c/bench_nosmall.err
tail -n 3 gpurun_out/r04c/ops.txt gpurun_out/r04c/parity.txt
cat gpurun_out/r04c/bench_elem.txt
