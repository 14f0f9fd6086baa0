#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_k; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_planes_gpu.py -m gpu -x -q -k "h8c" > $O/tests.txt 2>&1; tail -n 2 $O/tests.txt
timeout -k 10 120 tools/exp/bin/h8c_nx0 time 2>&1 | grep -v "grid=12" > $O/micro.txt
MMSA_ABLATE_FMT=h8 timeout -k 10 300 python tools/gemm_ablate.py 0 2 > $O/v2.txt 2>&1
MMSA_ABLATE_FMT=h8c timeout -k 10 300 python tools/gemm_ablate.py 0 2 > $O/h8c_keep.txt 2>&1
MMSA_ABLATE_FMT=h8c MMSA_ABLATE_LIB=libmmsa_knobs_nokeep.so timeout -k 10 300 python tools/gemm_ablate.py 0 2 > $O/h8c_nokeep.txt 2>&1
cut -c1-100 $O/micro.txt; for f in v2 h8c_keep h8c_nokeep; do echo "== $f"; grep "dbg" $O/$f.txt | cut -c1-100; done
