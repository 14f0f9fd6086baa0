"""Per-shape breakdown of the GEMM launches of one ViT-L forward (GPU box): which shapes eat the time."""
import collections, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
from mmsa import lib
from tests.configs import CONFIGS, make_input
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "vitl1024"]
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
x = make_input(cfg, batch=2).cuda()
m.multistream = False
m(x); torch.cuda.synchronize()
ops = mmsa.ops
shapes = []
orig = ops.gemm
def rec(a, w, *args, **kw):
    mm = kw.get("m") or (a.p.shape[0] if isinstance(a, ops.Planes) else a.shape[0])
    shapes.append((mm, w.n, w.kpad, kw.get("batch", 1), isinstance(a, ops.Planes), kw.get("out_planes") is not None, kw.get("act", "none")))
    return orig(a, w, *args, **kw)
prof = []
ops.GEMM_PROFILE = prof
ops.gemm = rec
import mmsa.backbone as bb
bb.ops.gemm = rec
m.multistream = False
m(x); torch.cuda.synchronize()
ops.GEMM_PROFILE = None
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for s, (f, e0, e1, _) in zip(shapes, prof):
    t = ctypes.c_float(); lib.call("mmsa_event_elapsed_ms", e0, e1, ctypes.byref(t))
    a = agg[s]; a[0] += 1; a[1] += t.value; a[2] += f
tot = sum(v[1] for v in agg.values())
print(f"total gemm ms {tot:.2f} launches {len(shapes)}")
for s, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"M={s[0]:7d} N={s[1]:5d} K={s[2]:5d} b={s[3]} Apl={int(s[4])} Cpl={int(s[5])} act={s[6]:7s} n={v[0]:3d} ms={v[1]:7.3f} ({100*v[1]/tot:4.1f}%) {v[2]/v[1]/1e9:7.1f} TF")
