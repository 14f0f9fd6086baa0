#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_j; mkdir -p $O
timeout -k 10 120 tools/exp/bin/h8c_nx0 time 2>&1 | grep -v "grid=12" > $O/micro.txt
MMSA_ABLATE_FMT=h8c timeout -k 10 300 python tools/gemm_ablate.py 2 > $O/real.txt 2>&1
MMSA_GEMM_ROWMAJOR=1 MMSA_ABLATE_FMT=h8c timeout -k 10 300 python tools/gemm_ablate.py 2 > $O/real_rowmajor.txt 2>&1
MMSA_GEMM_ROWMAJOR=1 MMSA_ABLATE_FMT=h8 timeout -k 10 300 python tools/gemm_ablate.py 2 > $O/v2_rowmajor.txt 2>&1
cut -c1-100 $O/micro.txt $O/real.txt $O/real_rowmajor.txt $O/v2_rowmajor.txt
