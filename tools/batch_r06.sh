#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=25 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -45 $O/gpu_tests.log | cut -c1-200
grep "\[fullsize\|\[masked" $O/gpu_tests.log | grep -v print | cut -c1-260
