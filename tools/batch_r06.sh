#!/bin/bash
# scratch batch for gpurun (round 6): `gpurun -- 'bash tools/batch_r06.sh'`; edited per call during the round, back to its stub
cd "$(dirname "$0")/.."
python -m pytest tests -m gpu -q 2>&1 | tail -5
