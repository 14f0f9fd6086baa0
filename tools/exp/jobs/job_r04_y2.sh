#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_y2; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "dwconv7" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout -k 10 200 python tools/dwconv_bench.py 2 > $O/new.txt 2>&1; cat $O/new.txt
