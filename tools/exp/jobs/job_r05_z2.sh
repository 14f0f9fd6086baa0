#!/bin/bash
# round 5, job z2: how much the golden-probe figure moves with changes far below the formats' rounding: GELU polynomial of degree 6 / 7 / 8 and the previous (A&S) form
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_z2; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 1000 python tools/ab_env.py --rounds 2 --steps 10 --verify deg6: deg7:MMSA_LIB=$R/ab/libmmsa_gelud7.so deg8:MMSA_LIB=$R/ab/libmmsa_gelud8.so as7126:MMSA_LIB=$R/ab/libmmsa_gelu0.so > $O/ab.txt 2>&1; cat $O/ab.txt
