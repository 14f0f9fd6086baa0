#!/bin/bash
# round 5, job c: full GPU suite on the register-epilogue build, then same-box A/B of the step against the round-4 library
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_c; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -x -q -m gpu --durations=8 > $O/tests.txt 2>&1; tail -n 14 $O/tests.txt
cp multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so ab/libmmsa_new.so
AB_NO_HEAD=0 timeout -k 10 900 python tools/ab_step.py ab/libmmsa_r04.so ab/libmmsa_new.so > $O/ab.txt 2>&1; cat $O/ab.txt
cp ab/libmmsa_new.so multimodal-sam-adapter_amd/mmsa/libmmsa_hip.so
