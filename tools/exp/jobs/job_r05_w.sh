#!/bin/bash
# round 5, job w: one-transcendental GELU (exp2 of a degree-6 polynomial for erfc): parity suites, sites, step A/B against the previous build
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_w; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_planes_gpu.py -m gpu -x -q > $O/t1.txt 2>&1; tail -n 4 $O/t1.txt
timeout -k 10 600 python tools/gemm_sites.py --rounds 5 --only lin1,cnx2pw1,cnx1pw1,cnx3pw1 ab/libmmsa_gelu0.so ab/libmmsa_new.so > $O/sites.txt 2>&1; cat $O/sites.txt
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_gelu0.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
timeout -k 10 900 python -m pytest tests/test_backbone_gpu.py -m gpu -x -q > $O/t2.txt 2>&1; tail -n 4 $O/t2.txt
