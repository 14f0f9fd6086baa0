#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_f3d; mkdir -p $O
timeout -k 10 1000 python tools/ab_env.py --rounds 2 --steps 20 --verify b3:MMSA_CNX_F16=0 f3: > $O/ab.txt 2>&1; cat $O/ab.txt
