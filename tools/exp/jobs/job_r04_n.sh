#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "dwconv or dwpair or convnext or gconv" > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
timeout -k 10 200 python tools/dwconv_bench.py 2 > $O/dwconv.txt 2>&1; timeout -k 10 200 python tools/dwconv_bench.py 1 >> $O/dwconv.txt 2>&1; grep -v amdgpu $O/dwconv.txt
timeout -k 10 200 python tools/dwpair_bench.py > $O/dwpair.txt 2>&1; grep -v amdgpu $O/dwpair.txt
