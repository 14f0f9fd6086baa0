#!/bin/bash
# round 5, job o: window attention by instruction count (scale folded into the exponential, maxima by threes, unconditional re-index writes)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bookkeeping_gpu.py tests/test_planes_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "window or bookkeeping or attention" > $O/t.txt 2>&1; tail -n 4 $O/t.txt
for i in 1 2; do
MMSA_LIB=$PWD/ab/libmmsa_prev.so timeout -k 10 200 python tools/wattn_bench.py 2 > $O/prev_$i.txt 2>&1; echo "prev: $(grep -h 'us per launch' $O/prev_$i.txt | head -3 | tr '\n' ' ')"
timeout -k 10 200 python tools/wattn_bench.py 2 > $O/new_$i.txt 2>&1; echo "new:  $(grep -h 'us per launch' $O/new_$i.txt | head -3 | tr '\n' ' ')"
done
