#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_mfma.sh r02 2>&1 | tail -40
bash tools/pmc_traffic.sh r02 2>&1 | tail -25
mkdir -p gpurun_out/r02_pmc && cp profiles/r02_mfma_util.json profiles/r02_gemm_traffic.json gpurun_out/r02_pmc/ 2>/dev/null
rm -rf gpurun_out/pmc_r02_*
