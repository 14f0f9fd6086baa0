#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_aa; mkdir -p $O
L=$GRAFT_REPO_ROOT/ab/libmmsa_conv_knobs.so
timeout -k 10 1000 python tools/ab_env.py --rounds 2 --steps 20 old_convs:MMSA_LIB=$L+MMSA_DWCONV7_BLK=0+MMSA_DWCONV3_STRIP=0 new_convs:MMSA_LIB=$L > $O/ab.txt 2>&1; cat $O/ab.txt
