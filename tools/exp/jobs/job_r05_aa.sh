#!/bin/bash
# round 5, job aa: cycle stamps of one tile boundary of the h8c GEMM (tools/epi_stamps.py)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_aa; mkdir -p $O
timeout -k 10 300 python tools/epi_stamps.py ab/libmmsa_estamp.so lin1 lin1none lin1bare qkv proj extout > $O/stamps.txt 2>&1; cat $O/stamps.txt
