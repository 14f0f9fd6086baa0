#!/bin/bash
# round 5, job ab: full GPU suite on the tree with the one-transcendental GELU (degree 7) and the packed h8 split
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_ab; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -n 5 $O/tests.txt
