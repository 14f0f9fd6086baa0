#!/bin/bash
# round 5, job a: register-resident GEMM epilogue -- parity of the new variants + the GEMM suites, then per-shape timing in the model against the round-4 library (same box)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "gemm or fold or f3 or convnext" > $O/tests.txt 2>&1; tail -n 15 $O/tests.txt
MMSA_LIB=$PWD/ab/libmmsa_r04.so timeout -k 10 300 python tools/gemm_shapes.py > $O/shapes_r04.txt 2>&1; head -12 $O/shapes_r04.txt
timeout -k 10 300 python tools/gemm_shapes.py > $O/shapes_new.txt 2>&1; head -12 $O/shapes_new.txt
