#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/c2/tests.txt
cat gpurun_out/c2/tests.txt
timeout 900 python tools/gemm_ablate.py 2>&1 | tee gpurun_out/c2/ablate.txt
