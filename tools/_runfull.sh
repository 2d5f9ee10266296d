mkdir -p gpurun_out/r04h
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r04h/gpu_tests.txt 2>&1
tail -n 5 gpurun_out/r04h/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04h/smoke.txt 2>&1; tail -2 gpurun_out/r04h/smoke.txt
bash tools/collect_bench_lines.sh r04 > gpurun_out/r04h/lines.txt 2>&1; tail -9 gpurun_out/r04h/lines.txt
