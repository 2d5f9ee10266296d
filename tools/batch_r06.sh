#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06n; mkdir -p $O
export TMPDIR=/tmp
PMF_CONV_F32=1 timeout 1200 python tools/soak_tensors.py --steps 47 --backbone resnet50 --nclasses 17 --height 480 --width 640 --masked > $O/soak_r50sb_f32.txt 2> $O/soak_r50sb_f32.err; echo "rc=$?"
grep "^grad enc.layer4\|^grad enc.layer1.0\|^grad fusion4\|^grad resBlock4" $O/soak_r50sb_f32.txt
grep "^param" $O/soak_r50sb_f32.txt | head -6
