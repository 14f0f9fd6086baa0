#!/bin/bash
# round 4, job c: fine stamps inside the phases of the h8c loop (full, no MFMA, no DMA)
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_c; mkdir -p $O
for v in fst fst_a1 fst_a4; do timeout -k 10 120 tools/exp/bin/h8c_$v time 2>&1 | grep -E "lin1  |fine" > $O/$v.txt; done
cat $O/*.txt
