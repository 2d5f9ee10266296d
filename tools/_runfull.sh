mkdir -p gpurun_out/r04h
( time python -m pytest tests -m gpu -q --durations=15 ) > gpurun_out/r04h/gpu_tests.txt 2>&1
grep -E "passed|failed|^FAILED" gpurun_out/r04h/gpu_tests.txt | tail -8
grep -A18 "slowest" gpurun_out/r04h/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
