#!/bin/bash
# usage: tools/pmc_mem.sh M N K tag  -- L2 hit rate, TA busy, TCP->TCC read latency of the GEMM kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
M=$1; N=$2; K=$3; TAG=$4
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pm1_$TAG -- python tools/gemm_one.py $M $N $K 3 > /dev/null 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pm2_$TAG -- python tools/gemm_one.py $M $N $K 3 > /dev/null 2>&1
python - <<PY
import csv,collections,glob
for d in ("pm1_$TAG","pm2_$TAG"):
    fs=glob.glob("gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True)
    if not fs: print(d,"no output"); continue
    agg=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("$TAG", {k: round(v/n[k],1) for k,v in agg.items()})
PY
