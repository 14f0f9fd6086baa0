#!/bin/bash
# round 5, job t: two-workgroups-per-CU h8c microbenchmark (tools/exp/h8c_2wg.hip) against the library kernel on the same box
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_t; mkdir -p $O
timeout -k 10 120 tools/exp/bin/h8c_2wg_e0 check > $O/check.txt 2>&1 || { tail -8 $O/check.txt; exit 1; }
tail -n 3 $O/check.txt
for e in 0 1 2; do timeout -k 10 120 tools/exp/bin/h8c_2wg_e$e time > $O/time_e$e.txt 2>&1; grep "h8c 2wg" $O/time_e$e.txt | cut -c1-140; done
timeout -k 10 600 python tools/gemm_sites.py --rounds 3 --only lin1,qkv,lin2,proj,extout,ffnfc2 multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_knobs.so:MMSA_GEMM_DEBUG=2 > $O/sites.txt 2>&1; cat $O/sites.txt
