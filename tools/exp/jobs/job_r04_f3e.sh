#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_f3e; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -8 $O/tests.txt
TAG="f3 + h8c lo compensation" timeout -k 10 200 python tools/exp/probe_detail.py 2>&1 | tail -1 | tee -a $O/detail.txt
timeout -k 10 400 python tools/error_budget.py $O/error_budget.json 2>&1 | tail -24
