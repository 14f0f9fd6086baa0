#!/bin/bash
# usage: tools/pmc_gemm.sh M N K tag   (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
M=$1; N=$2; K=$3; TAG=$4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG -- python tools/gemm_one.py $M $N $K 3 > /dev/null 2>&1
python - <<PY
import csv,collections,glob
f=glob.glob("gpurun_out/pmc_$TAG/**/*counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(f)):
    if "gemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
g=agg["GRBM_GUI_ACTIVE"]/n["GRBM_GUI_ACTIVE"]/8
print("$TAG $M x $N x $K: cycles/XCD %.0f  mfma_util %.3f  wait_any %.3f wait_inst %.3f active %.3f lds_idx %.3f" % (g,
  agg["SQ_VALU_MFMA_BUSY_CYCLES"]/n["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/g,
  agg["SQ_WAIT_ANY"]/agg["SQ_WAVE_CYCLES"], agg["SQ_WAIT_INST_ANY"]/agg["SQ_WAVE_CYCLES"], agg["SQ_ACTIVE_INST_ANY"]/agg["SQ_WAVE_CYCLES"],
  agg["SQ_LDS_IDX_ACTIVE"]/n["SQ_LDS_IDX_ACTIVE"]/256/g))
f=glob.glob("gpurun_out/pmc_$TAG/**/*kernel_trace.csv", recursive=True)[0]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if "gemm_v2" in r["Kernel_Name"] or "gemm_split3" in r["Kernel_Name"]]
print("kernel us:", ["%.1f"%x for x in d])
PY
