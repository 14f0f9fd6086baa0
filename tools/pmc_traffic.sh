#!/bin/bash
# HBM-side traffic of the GEMM kernels (MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes,
# FETCH_SIZE doubled on gfx950 for 16-B/lane streaming reads; kernel trace beside the counters for the per-kernel durations of
# profiles/<tag>_hbm_kernels.json).  Run on the GPU box:  bash tools/pmc_traffic.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -- python bench.py --steps 1 --warmup 1 --no-graph --no-verify --no-cpu-baseline --no-roofline --no-head > /dev/null 2>&1
  echo "pass $C rc=$?"
done
python tools/pmc_traffic.py $TAG
