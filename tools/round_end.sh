#!/bin/bash
# Round-end measurement on the GPU box (bash tools/round_end.sh <tag> -> gpurun_out/<tag>/): full GPU suite, bench line (+cpu baseline), rocprof kernel stats, shape table, frame bench, ViT-H line, RCCL path
TAG=${1:-r02_c}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
# (verbose, straight into a file under gpurun_out/: a run that prints nothing for 7 minutes is taken to be hung)
timeout -k 10 1500 python -m pytest tests -x -v -m gpu --durations=10 > gpurun_out/$TAG/tests_full.txt 2>&1
grep -v "^E    .*tensor(\[" gpurun_out/$TAG/tests_full.txt | tail -16 > gpurun_out/$TAG/tests.txt
cat gpurun_out/$TAG/tests.txt
timeout 600 python bench.py 2> gpurun_out/$TAG/bench.err | tail -1 > gpurun_out/$TAG/bench.json
cat gpurun_out/$TAG/bench.json | cut -c1-1200; tail -2 gpurun_out/$TAG/bench.err
timeout 600 python bench.py --chains 1 --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/$TAG/bench_chains1.json
python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_chains1.json')); print('chains=1:', d['value'], d['ms_per_step'], d['encoder_only'])"
MMSA_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline 2> gpurun_out/$TAG/dist.err | tail -1 > gpurun_out/$TAG/bench_dist1.json
python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_dist1.json')); print('RCCL 1 rank:', d['value'], d['ms_per_step'], d['config']['collective'])" || tail -5 gpurun_out/$TAG/dist.err
# (one chain: under the tracer kernels of concurrent chains no longer overlap; --no-roofline: no single-stream pass, no worst-case-precision leg -- the table is the default step only)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --chains 1 > /dev/null 2>&1
cp $(ls gpurun_out/$TAG/prof/*/*kernel_stats.csv | head -1) gpurun_out/$TAG/kernel_stats.csv
rm -rf gpurun_out/$TAG/prof
python tools/kstats.py gpurun_out/$TAG/kernel_stats.csv auto 60 > gpurun_out/$TAG/kstats.txt; head -14 gpurun_out/$TAG/kstats.txt
timeout 300 python tools/gemm_shapes.py > gpurun_out/$TAG/shapes.txt 2>&1
head -3 gpurun_out/$TAG/shapes.txt
timeout 300 python tools/frame_bench.py > gpurun_out/$TAG/frame.txt 2>&1; tail -1 gpurun_out/$TAG/frame.txt | cut -c1-300
timeout 400 python bench.py --config vith1024 --no-cpu-baseline --steps 10 --warmup 3 2> gpurun_out/$TAG/vith.err | tail -1 > gpurun_out/$TAG/vith.json
cat gpurun_out/$TAG/vith.json | cut -c1-300
