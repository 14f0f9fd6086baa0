#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_f3b; mkdir -p $O
TAG="default (cnx f3)" timeout -k 10 200 python tools/exp/probe_detail.py 2>&1 | tail -1 | tee -a $O/detail.txt
TAG="cnx b3" MMSA_CNX_F16=0 timeout -k 10 200 python tools/exp/probe_detail.py 2>&1 | tail -1 | tee -a $O/detail.txt
TAG="cnx f3 + attention b3" MMSA_ATTN=b3 timeout -k 10 200 python tools/exp/probe_detail.py 2>&1 | tail -1 | tee -a $O/detail.txt
TAG="cnx f3 + no h8 sites" MMSA_H8=none timeout -k 10 200 python tools/exp/probe_detail.py 2>&1 | tail -1 | tee -a $O/detail.txt
TAG="cnx f3 + no h8 + attention b3" MMSA_H8=none MMSA_ATTN=b3 timeout -k 10 200 python tools/exp/probe_detail.py 2>&1 | tail -1 | tee -a $O/detail.txt
