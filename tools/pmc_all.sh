#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_mfma.sh r02 2>&1 | grep -E "rc=|mfma_util_pct" | head
bash tools/pmc_traffic.sh r02 2>&1 | grep -E "per_launch|rc="
mkdir -p gpurun_out/r02_pmc && cp profiles/r02_mfma_util.json profiles/r02_gemm_traffic.json gpurun_out/r02_pmc/
rm -rf gpurun_out/pmc_r02_*
