"""Summarise tools/pmc_mfma.sh into profiles/<tag>_mfma_util.json: per kernel the duration-weighted MfmaUtil (%), the MFMA
flops the hardware counted (MOPS x 512) and the resulting MFMA TFLOP/s; the same for the GEMM family and the whole step."""
import collections, csv, glob, json, sys
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "multimodal-sam-adapter_amd"))
import build as _build   # multimodal-sam-adapter_amd/build.py
STAMP = {"source_digest": _build.source_digest(), "commit": _os.environ.get("MMSA_COMMIT", "n/a")}   # bench.py attaches a profile only to the sources it was measured on
tag = sys.argv[1]


def load(sub, names):
    d = f"gpurun_out/pmc_{tag}_mfma_{sub}"
    dur = {}
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    val = collections.defaultdict(dict)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in names:
                val[r["Dispatch_Id"]][r["Counter_Name"]] = val[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return dur, val


dur, val = load("util", ("MfmaUtil",))
per = collections.defaultdict(lambda: {"launches": 0, "ns": 0, "util_ns": 0.0, "mfma_flop": 0.0, "ops_ns": 0, "by_type": collections.defaultdict(float)})
for k, (name, ns) in dur.items():
    if k in val and "MfmaUtil" in val[k]:
        p = per[name]; p["launches"] += 1; p["ns"] += ns; p["util_ns"] += val[k]["MfmaUtil"] * ns
MOPS = ("SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_F8", "SQ_INSTS_VALU_MFMA_MOPS_F32")
dur2, val2 = load("ops", MOPS)
for k, (name, ns) in dur2.items():
    if k in val2:
        per[name]["mfma_flop"] += 512.0 * sum(val2[k].values()); per[name]["ops_ns"] += ns
        for c, v in val2[k].items():
            per[name]["by_type"][c.rsplit("_", 1)[1]] += 512.0 * v


def fold(keys):
    ns = sum(per[k]["ns"] for k in keys); ons = sum(per[k]["ops_ns"] for k in keys)
    fl = sum(per[k]["mfma_flop"] for k in keys)
    bt = collections.defaultdict(float)
    for k in keys:
        for c, v in per[k]["by_type"].items():
            bt[c] += v
    return {"launches": sum(per[k]["launches"] for k in keys), "ms": ns / 1e6, "mfma_util_pct": sum(per[k]["util_ns"] for k in keys) / max(ns, 1),
            "mfma_flop_counted": fl, "mfma_tflops": fl / max(ons, 1) / 1e3, "mfma_flop_by_type": dict(bt)}


gem = [k for k in per if "gemm" in k or "mlp_fused" in k]   # the contraction kernels
# the library's own kernels only: the pass also contains torch / hipBLASLt / runtime-copy kernels of the FIRST forward (weight packing,
# nothing else since round 4: the logit range is measured by the attention kernels themselves), which are not part of a step
FOREIGN = ("Cijk_", "void at::native", "__amd_rocclr", "void rocprim", "void at::cuda")
own = [k for k in per if not k.startswith(FOREIGN)]
res = {**STAMP, "note": "eager forwards of bench.py (ViT-L 1024^2 RGB+LiDAR, batch 2, encoder + head; warm-up + one step) under rocprofv3 --pmc; whole_step = this library's kernels only; kernels run serialised "
               "under counter collection, so ms is the sum of kernel durations, not the step time; util is duration-weighted MfmaUtil; "
               "mfma_flop_counted = 512 x (MOPS_BF16 + MOPS_F16 + MOPS_F8 + MOPS_F32): 3 x the algorithmic flops for bf16 hi/lo contractions, "
               "1 (F16) + 2 (F8, at twice the rate) for h8 contractions",
       "whole_step": fold(own), "whole_pass_incl_first_forward_torch_kernels": fold(list(per)), "gemm_family": fold(gem),
       "per_kernel": {k: fold([k]) for k in sorted(per, key=lambda k: -per[k]["ns"])[:25]}}
json.dump(res, open(f"profiles/{tag}_mfma_util.json", "w"), indent=1)
print(json.dumps({"whole_step": res["whole_step"], "gemm_family": res["gemm_family"]}, indent=1))
for k, v in list(res["per_kernel"].items())[:12]:
    print(f"{k[:50]:50s} {v['ms']:8.3f} ms  util {v['mfma_util_pct']:6.2f} %  {v['mfma_tflops']:8.1f} TF")
