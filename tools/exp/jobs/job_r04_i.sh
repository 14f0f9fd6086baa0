#!/bin/bash
# round 4, job i: where the h8c GEMM's time goes: full / loop-only, unrolled / rolled epilogue, against gemm_v2's h8; dwconv7 with packed FMAs
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_i; mkdir -p $O
MMSA_ABLATE_FMT=h8 timeout -k 10 300 python tools/gemm_ablate.py 0 2 > $O/ablate_h8.txt 2>&1
MMSA_ABLATE_FMT=h8c timeout -k 10 300 python tools/gemm_ablate.py 0 2 > $O/ablate_h8c.txt 2>&1
MMSA_ABLATE_FMT=h8c MMSA_ABLATE_LIB=libmmsa_knobs_rolled.so timeout -k 10 300 python tools/gemm_ablate.py 0 2 > $O/ablate_h8c_rolled.txt 2>&1
timeout -k 10 120 python tools/dwconv_bench.py 2 > $O/dwconv.txt 2>&1
timeout -k 10 120 python tools/dwconv_bench.py 1 >> $O/dwconv.txt 2>&1
cut -c1-110 $O/ablate_h8.txt $O/ablate_h8c.txt $O/ablate_h8c_rolled.txt; cat $O/dwconv.txt
