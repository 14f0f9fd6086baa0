#!/bin/bash
# round 4, job b: h8c loop after the MFMA-order fix: check, interval stamps, ablations (1 no MFMA, 2 no LDS reads, 4 no DMA)
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_b; mkdir -p $O
timeout -k 10 120 tools/exp/bin/h8c_nx0 check > $O/check.txt 2>&1
for v in nx0 st a1 a2 a4 a6 a5 a3; do timeout -k 10 120 tools/exp/bin/h8c_$v time 2>&1 | grep -E "lin1 |lin2 |qkv|ext out|stamps" > $O/time_$v.txt; done
tail -n 3 $O/check.txt; cat $O/time_*.txt
