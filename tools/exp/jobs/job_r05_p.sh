#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_p; mkdir -p $O
for i in 1 2; do for v in prev wa_fold wa_unc; do
MMSA_LIB=$PWD/ab/libmmsa_$v.so timeout -k 10 200 python tools/wattn_bench.py 2 > $O/${v}_$i.txt 2>&1; echo "$v: $(grep -h 'us per launch' $O/${v}_$i.txt | head -1)"
done; done
