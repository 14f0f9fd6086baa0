#!/bin/bash
# round 5, job ag: two free-running chains with an initial lag (GEMM phase of one beside the spatial-prior phase of the other)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_ag; mkdir -p $O
timeout -k 10 600 python tools/exp/chains_lag.py 24 > $O/lag.txt 2>&1; grep -v amdgpu $O/lag.txt | tail -12
