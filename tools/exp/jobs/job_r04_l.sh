#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_l; mkdir -p $O
timeout -k 10 900 python tools/ab_env.py --rounds 2 --steps 20 --verify h8line:MMSA_H8C=0 h8c: > $O/ab.txt 2>&1
cat $O/ab.txt
timeout -k 10 200 python tools/dwconv_bench.py 2 > $O/dwconv.txt 2>&1; cat $O/dwconv.txt
