#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_f3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_planes_gpu.py -x -q -m gpu -k "f3" > $O/tests_f3.txt 2>&1; tail -15 $O/tests_f3.txt
timeout -k 10 900 python tools/ab_env.py --rounds 1 --steps 20 --verify b3:MMSA_CNX_F16=0 f3: > $O/ab.txt 2>&1; cat $O/ab.txt
