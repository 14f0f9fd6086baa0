#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_f3g; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -12 $O/tests.txt
timeout -k 10 300 python tools/exp/peaky_detail.py 2>&1 | tail -1
