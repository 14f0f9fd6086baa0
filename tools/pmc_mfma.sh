#!/bin/bash
# MFMA utilisation of one bench step per kernel (rocprofiler-sdk derived metric MfmaUtil =
# sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM) * 100) and the MFMA op counters
# (SQ_INSTS_VALU_MFMA_MOPS_{BF16,F16,F8,F32}: ops / 512; the h8 GEMMs issue F16 and block-scaled F8 MFMAs).  Counters in their own passes, kernel trace only beside them.
# Run on the GPU box:  bash tools/pmc_mfma.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
rm -rf gpurun_out/pmc_${TAG}_mfma_*
timeout 500 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_mfma_util -- python bench.py --steps 1 --warmup 1 --no-graph --no-verify --no-cpu-baseline --no-roofline > /dev/null 2>&1
echo "pass MfmaUtil rc=$?"
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_mfma_ops -- python bench.py --steps 1 --warmup 1 --no-graph --no-verify --no-cpu-baseline --no-roofline > /dev/null 2>&1
echo "pass MOPS rc=$?"
python tools/pmc_mfma.py $TAG
