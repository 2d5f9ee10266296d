#!/bin/bash
# sweep tile / split configurations of the conv kernel over the micro-benchmark cases (tools/bench_conv.py)
for cfg in 0,0,0 32,1,1 32,2,1 64,1,1 64,2,1 32,1,2 64,1,2 32,1,4 64,1,4 64,1,8 32,1,8; do
  echo "== BN,MT,KS = $cfg"
  PMF_CONV_FORCE=$cfg python tools/bench_conv.py fwd 2>&1 | grep -v amdgpu.ids
done
