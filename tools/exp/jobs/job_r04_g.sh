#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_planes_gpu.py -m gpu -x -q -k "h8c or gemm_h8 or flavour" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
tail -n 40 $O/tests.txt
