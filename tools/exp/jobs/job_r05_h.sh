#!/bin/bash
# round 5, job h: clamp watch -- its test, the suites of the entry points it touched, then the full suite
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_backbone_gpu.py -m gpu -x -q -k "clamp_flag" > $O/t1.txt 2>&1; tail -n 12 $O/t1.txt
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -n 8 $O/tests.txt
cp multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_new.so
AB_NO_HEAD=0 timeout -k 10 600 python tools/ab_step.py ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
