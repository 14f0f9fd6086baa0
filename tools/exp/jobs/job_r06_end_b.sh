#!/bin/bash
TAG=r06_z
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 600 python bench.py --chains 1 --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/$TAG/bench_chains1.json
python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_chains1.json')); print('chains=1:', d['value'], d['ms_per_step'], d['encoder_only'])"
MMSA_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline 2> gpurun_out/$TAG/dist.err | tail -1 > gpurun_out/$TAG/bench_dist1.json
python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_dist1.json')); print('RCCL 1 rank:', d['value'], d['ms_per_step'], d['config']['collective'])" || tail -5 gpurun_out/$TAG/dist.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --chains 1 > /dev/null 2>&1
cp $(ls gpurun_out/$TAG/prof/*/*kernel_stats.csv | head -1) gpurun_out/$TAG/kernel_stats.csv
rm -rf gpurun_out/$TAG/prof
python tools/kstats.py gpurun_out/$TAG/kernel_stats.csv auto 60 > gpurun_out/$TAG/kstats.txt; head -14 gpurun_out/$TAG/kstats.txt
timeout 300 python tools/gemm_shapes.py > gpurun_out/$TAG/shapes.txt 2>&1
head -3 gpurun_out/$TAG/shapes.txt
timeout 300 python tools/frame_bench.py > gpurun_out/$TAG/frame.txt 2>&1; tail -1 gpurun_out/$TAG/frame.txt | cut -c1-300
timeout 400 python bench.py --config vith1024 --no-cpu-baseline --steps 10 --warmup 3 2> gpurun_out/$TAG/vith.err | tail -1 > gpurun_out/$TAG/vith.json
cat gpurun_out/$TAG/vith.json | cut -c1-300
