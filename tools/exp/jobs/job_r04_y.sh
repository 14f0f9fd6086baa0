#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_y; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "dwconv or conv or tiny" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
MMSA_LIB=$GRAFT_REPO_ROOT/ab/libmmsa_conv_knobs.so MMSA_DWCONV7_BLK=0 timeout -k 10 200 python tools/dwconv_bench.py 2 > $O/old.txt 2>&1; cat $O/old.txt
timeout -k 10 200 python tools/dwconv_bench.py 2 > $O/new.txt 2>&1; cat $O/new.txt
