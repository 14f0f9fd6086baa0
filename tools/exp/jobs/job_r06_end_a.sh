#!/bin/bash
TAG=r06_z
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests -x -v -m gpu --durations=10 > gpurun_out/$TAG/tests_full.txt 2>&1
grep -v "^E    .*tensor(\[" gpurun_out/$TAG/tests_full.txt | tail -16 > gpurun_out/$TAG/tests.txt
tail -3 gpurun_out/$TAG/tests.txt
timeout 600 python bench.py 2> gpurun_out/$TAG/bench.err | tail -1 > gpurun_out/$TAG/bench.json
cat gpurun_out/$TAG/bench.json | cut -c1-900; tail -2 gpurun_out/$TAG/bench.err
