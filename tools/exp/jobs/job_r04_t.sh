#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_t; mkdir -p $O
run() { DIAG_TAG="$1" timeout -k 10 200 python tools/exp/chains_diag.py 2>&1 | grep "^\[" | tee -a $O/diag.txt; }
DIAG_ROOT=$GRAFT_REPO_ROOT/ab/r03 run "r03 code"
run "head, r03 creation order"
DIAG_GUARD=off run "head guard off"
