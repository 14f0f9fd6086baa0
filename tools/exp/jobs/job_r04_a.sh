#!/bin/bash
# round 4, job a: h8c loop microbenchmark (check + timing, three DMA splits) beside gemm_v2's h8 loop (full kernel / loop only) on the same box
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_a; mkdir -p $O
timeout -k 10 120 tools/exp/bin/h8c_loop_nx0 check > $O/h8c_check_nx0.txt 2>&1
timeout -k 10 120 tools/exp/bin/h8c_loop_nx1 check > $O/h8c_check_nx1.txt 2>&1
timeout -k 10 120 tools/exp/bin/h8c_loop_nx2 check > $O/h8c_check_nx2.txt 2>&1
for nx in 0 1 2; do timeout -k 10 120 tools/exp/bin/h8c_loop_nx$nx time > $O/h8c_time_nx$nx.txt 2>&1; done
MMSA_ABLATE_FMT=h8 timeout -k 10 300 python tools/gemm_ablate.py 0 2 > $O/gemm_v2_h8_ablate.txt 2>&1
tail -n 30 $O/*.txt
