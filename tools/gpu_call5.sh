#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
timeout 900 python -m pytest tests/test_inference_gpu.py -x -q -m gpu 2>&1 | grep -v "^E    .*tensor(\[" | tail -40 > gpurun_out/c5/tests.txt
cat gpurun_out/c5/tests.txt
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_planes_gpu.py -x -q -m gpu 2>&1 | tail -5
