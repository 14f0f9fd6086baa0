#!/bin/bash
# round 5, job s: "rows" form extended to per-column scales and 96-column tiles (ConvNeXt pw2 and the neck's scaled 1x1 convs)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "gemm or fold or f3 or convnext" > $O/t.txt 2>&1; tail -n 4 $O/t.txt
timeout -k 10 600 python tools/gemm_sites.py --rounds 5 --only cnx2pw2,cnx1pw2,cnx3pw2,extout,lin1 ab/libmmsa_prev.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so > $O/sites.txt 2>&1; cat $O/sites.txt
cp multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_new.so
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_prev.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
