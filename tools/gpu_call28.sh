#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/chains_bench.py 2 2 2>&1 | grep "chain"
MMSA_GEMM_MAX_GRID=128 timeout 300 python tools/split_batch_bench.py 1 2 2>&1 | grep "images/s"
MMSA_GEMM_MAX_GRID=128 timeout 300 python tools/chains_bench.py 2 2 2>&1 | grep "chain"
