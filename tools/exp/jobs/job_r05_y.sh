#!/bin/bash
# round 5, job y: is the ViT blocks' LayerNorm fold (round 3) still a gain now that the GEMM epilogues are issue-bound?  default | fold_ln = False
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_y; mkdir -p $O
timeout -k 10 1000 python tools/ab_env.py --rounds 2 --steps 20 --verify fold: nofold@fold_ln=False > $O/ab.txt 2>&1; cat $O/ab.txt
