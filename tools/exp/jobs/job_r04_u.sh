#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "conv or attention or msda or gram or gfe or neck or tiny or dwpair" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout -k 10 900 python tools/ab_env.py --rounds 2 --steps 20 --verify noxcd:MMSA_LIB=$GRAFT_REPO_ROOT/ab/libmmsa_noxcd.so xcd: > $O/ab.txt 2>&1; cat $O/ab.txt
bash tools/pmc_traffic.sh r04u > $O/pmc.txt 2>&1; tail -3 $O/pmc.txt
cp profiles/r04u_hbm_kernels.json profiles/r04u_gemm_traffic.json $O/
rm -rf gpurun_out/pmc_r04u_*
# SLP hazard: conv_pair.hip built WITH the vectoriser, the one-pixel kernel forced, beside a second encoder instance
MMSA_LIB=$GRAFT_REPO_ROOT/ab/libmmsa_slp_pair.so MMSA_DWPAIR_STRIP=0 ITERS=80 timeout -k 10 300 python tools/stress_concurrent.py model > $O/slp_old_kernel.txt 2>&1; tail -3 $O/slp_old_kernel.txt
MMSA_LIB=$GRAFT_REPO_ROOT/ab/libmmsa_slp_pair.so ITERS=80 timeout -k 10 300 python tools/stress_concurrent.py model > $O/slp_strip_kernel.txt 2>&1; tail -3 $O/slp_strip_kernel.txt
