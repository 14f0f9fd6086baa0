#!/bin/bash
# round 5, job z: golden probes of the replayed graph (rel-L2 and max-abs parts) with the previous GELU build and the new one, same box
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_z; mkdir -p $O
timeout -k 10 1000 python tools/ab_env.py --rounds 1 --steps 5 --verify new: oldgelu:MMSA_LIB=$GRAFT_REPO_ROOT/ab/libmmsa_gelu0.so > $O/ab.txt 2>&1; cat $O/ab.txt
