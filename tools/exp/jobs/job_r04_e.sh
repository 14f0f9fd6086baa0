#!/bin/bash
# round 4, job e: h8c loop, DMA split between the two read phases (NX of the 6 HI pieces issued in phase X)
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_e; mkdir -p $O
for nx in 1 2 3; do timeout -k 10 120 tools/exp/bin/h8c_nx$nx check > $O/check_nx$nx.txt 2>&1 || { tail -5 $O/check_nx$nx.txt; exit 1; }; done
for v in nx0 nx1 nx2 nx3 nx0; do timeout -k 10 120 tools/exp/bin/h8c_$v time 2>&1 | grep -v "grid=12[38]" >> $O/time.txt; done
cat $O/time.txt
