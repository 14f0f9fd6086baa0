#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_s; mkdir -p $O
run() { DIAG_TAG="$1" timeout -k 10 200 python tools/exp/chains_diag.py 2>&1 | grep "^\[" | tee -a $O/diag.txt; }
DIAG_ROOT=$GRAFT_REPO_ROOT/ab/r03 run "r03 code"
run "head default"
DIAG_GUARD=off run "head guard off"
DIAG_MS=0 run "head multistream off"
MMSA_H8C=0 run "head h8 line planes"
DIAG_CAP=96 run "head cap 96"
DIAG_CAP=64 run "head cap 64"
