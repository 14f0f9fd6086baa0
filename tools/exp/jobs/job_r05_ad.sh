#!/bin/bash
# round 5, job ad: two workgroups per CU with wave priorities (k loop at s_setprio 3, epilogue at 0) -- does the epilogue then fill the issue slots the MFMA phases leave?
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_ad; mkdir -p $O
timeout -k 10 120 tools/exp/bin/h8c_2wg_e0p1 check > $O/check.txt 2>&1 || { tail -5 $O/check.txt; exit 1; }
tail -n 1 $O/check.txt
for v in e0p1 e1p0 e1p1 e2p0 e2p1; do timeout -k 10 120 tools/exp/bin/h8c_2wg_$v time > $O/time_$v.txt 2>&1; echo "== $v"; grep "grid=512\|grid=448" $O/time_$v.txt | cut -c1-120; done
timeout -k 10 400 python tools/gemm_sites.py --rounds 3 --only lin1,qkv,lin2,proj,extout,ffnfc2 multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so > $O/sites.txt 2>&1; cat $O/sites.txt
