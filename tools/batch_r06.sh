#!/bin/bash
# scratch batch for gpurun (round 6); edited per call
cd "$(dirname "$0")/.."
O=gpurun_out/r06q; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_bf16_probe.hip -o /tmp/mfma_probe && timeout 120 /tmp/mfma_probe > $O/mfma_probe.txt 2>&1; echo "rc=$?"; cat $O/mfma_probe.txt
