#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06p; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=6 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -14 $O/gpu_tests.log | cut -c1-200
SKIP_PROFILES=1 bash tools/r06_evidence.sh 4c81938
