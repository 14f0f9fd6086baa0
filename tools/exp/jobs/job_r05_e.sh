#!/bin/bash
# round 5, job e: slack stagger of the persistent GEMM grids -- GEMM suites, site timing and step A/B against the same build without it
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_e; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_planes_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "gemm or fold or f3" > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
timeout -k 10 600 python tools/gemm_sites.py --rounds 3 ab/libmmsa_nostagger.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so > $O/sites.txt 2>&1; cat $O/sites.txt
cp multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_new.so
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_nostagger.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
