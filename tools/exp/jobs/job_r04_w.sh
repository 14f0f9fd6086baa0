#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_w; mkdir -p $O
MMSA_LIB=$GRAFT_REPO_ROOT/ab/libmmsa_wattn_dbg.so timeout -k 10 120 python tools/wattn_bench.py 2 --stamps > $O/stamps_b2.txt 2>&1; cat $O/stamps_b2.txt | cut -c1-400
MMSA_LIB=$GRAFT_REPO_ROOT/ab/libmmsa_wattn_dbg.so timeout -k 10 120 python tools/wattn_bench.py 1 --stamps > $O/stamps_b1.txt 2>&1; tail -3 $O/stamps_b1.txt | cut -c1-400
