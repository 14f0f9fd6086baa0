#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_f3f; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -6 $O/tests.txt
