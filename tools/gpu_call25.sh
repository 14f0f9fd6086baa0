#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c25
bash tools/pmc_mfma.sh r02 2>&1 | grep -E "rc=|util" | head -20
bash tools/pmc_traffic.sh r02 2>&1 | grep -E "per_launch|rc="
cp profiles/r02_mfma_util.json profiles/r02_gemm_traffic.json gpurun_out/c25/
rm -rf gpurun_out/pmc_r02_*
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/c25/bench.err | tail -1 > gpurun_out/c25/bench_$i.json
python -c "import json; d=json.load(open('gpurun_out/c25/bench_$i.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], d['encoder_only'], r['achieved'], r['frac'], r['kernel_ms_per_step'], r['traffic'], r['mfma_counters'])"
done
timeout 300 python tools/gemm_shapes.py 2>&1 | grep "GEMM launches"
