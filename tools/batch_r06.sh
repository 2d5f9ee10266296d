#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
PMF_AUTOTUNE=0 timeout 1200 python tools/masked_tensors.py > $O/masked_tensors_tune0.txt 2> $O/masked_tensors_tune0.err; echo "rc=$?"
grep -B3 -A12 "<--" $O/masked_tensors_tune0.txt | head -80
grep -A8 "^==" $O/masked_tensors_tune0.txt | head -60
tail -5 $O/masked_tensors_tune0.err
