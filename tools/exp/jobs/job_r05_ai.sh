#!/bin/bash
# round 5, job ai: f3 operands on the 4-wave GEMM flavour where 256-row tiles under-fill the chip (ConvNeXt stage 3): parity, sites, step
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_ai; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_planes_gpu.py -m gpu -x -q -k "flavours or f3 or convnext" > $O/t1.txt 2>&1; tail -n 3 $O/t1.txt
timeout -k 10 600 python tools/gemm_sites.py --rounds 5 --only cnx3pw1,cnx3pw2,cnx2pw2,cnx1pw1 ab/libmmsa_knobs.so:MMSA_GEMM_F3_NW4_TILES=0 ab/libmmsa_knobs.so > $O/sites.txt 2>&1; cat $O/sites.txt
timeout -k 10 900 python tools/ab_env.py --rounds 2 --steps 20 --verify off:MMSA_LIB=$R/ab/libmmsa_knobs.so+MMSA_GEMM_F3_NW4_TILES=0 on:MMSA_LIB=$R/ab/libmmsa_knobs.so > $O/ab.txt 2>&1; cut -c1-200 $O/ab.txt
