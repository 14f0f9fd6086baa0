#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c13
for rnd in 1 2; do
for f in b3 h8; do
  echo "## $f" | tee -a gpurun_out/c13/ablate.txt
  MMSA_ABLATE_FMT=$f timeout 300 python tools/gemm_ablate.py 0 2 2>&1 | tee -a gpurun_out/c13/ablate.txt
done
done
