#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_mfma.sh ${1:-r03} 2>&1 | grep -E "rc=|mfma_util_pct" | head
bash tools/pmc_traffic.sh ${1:-r03} 2>&1 | grep -E "per_launch|rc=|hbm kernels"
mkdir -p gpurun_out/${1:-r03}_pmc && cp profiles/${1:-r03}_mfma_util.json profiles/${1:-r03}_gemm_traffic.json profiles/${1:-r03}_hbm_kernels.json gpurun_out/${1:-r03}_pmc/
rm -rf gpurun_out/pmc_${1:-r03}_*
