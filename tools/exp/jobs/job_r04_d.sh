#!/bin/bash
# round 4, job d: h8c loop with pinned scheduling (straight-line steady-state pairs): check, stamps, ablations
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_d; mkdir -p $O
timeout -k 10 120 tools/exp/bin/h8c_nx0 check > $O/check.txt 2>&1 || { tail -5 $O/check.txt; exit 1; }
for v in nx0 st fst a1 a4 a6 a5 a3; do timeout -k 10 120 tools/exp/bin/h8c_$v time 2>&1 | grep -E "lin1  |lin2  |qkv|ext out|stamps" | grep -v "grid=12[38]" > $O/time_$v.txt; done
tail -n 2 $O/check.txt; cat $O/time_*.txt | cut -c1-400
