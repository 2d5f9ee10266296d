#!/bin/bash
# scratch batch script of round 5 (what-if timings, A/B runs): see tools/r05_evidence.sh for the evidence batch that produced
# profiles/r05_* and docs/rounds/r05.md for the results of the what-ifs
echo "edit me: one gpurun call = one batch of measurements"
