#!/bin/bash
# round 4, job f: full GPU suite + bench line after the guard / boundary-cleanup commit
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_f; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
tail -n 15 $O/gpu_tests.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 3000 $O/bench.json; tail -n 5 $O/bench.err
