#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_m; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
tail -n 12 $O/gpu_tests.txt
