#!/bin/bash
# round 5, job n: adapter-token LayerNorm fold + padded offset projections -- its test, the backbone suite, same-box A/B of the step with and without the fold
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_backbone_gpu.py -m gpu -x -q -k "adapter_layernorm_fold" > $O/t1.txt 2>&1; tail -n 12 $O/t1.txt
timeout -k 10 1200 python -m pytest tests/test_backbone_gpu.py tests/test_inference_gpu.py tests/test_bench_gpu.py -m gpu -x -q > $O/t2.txt 2>&1; tail -n 6 $O/t2.txt
timeout -k 10 900 python tools/ab_env.py --rounds 2 --steps 20 --verify nofold@fold_adapter_ln=False fold: > $O/ab.txt 2>&1; cat $O/ab.txt
