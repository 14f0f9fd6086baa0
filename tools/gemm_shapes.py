"""Per-shape table of the split3 GEMM launches of one bench step (single stream, HIP events per launch).
python tools/gemm_shapes.py"""
import collections, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch  # noqa: E402
import mmsa  # noqa: E402
from mmsa import lib  # noqa: E402
from tests.configs import CONFIGS, make_input  # noqa: E402

cfg = CONFIGS["vitl1024"]
torch.manual_seed(1234)
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
m.multistream = False
x = make_input(cfg, batch=2, seed=1234).to("cuda:0")
for _ in range(2):
    m(x)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for rep in range(3):
    prof, shapes = [], []
    mmsa.ops.GEMM_PROFILE, mmsa.ops.GEMM_SHAPES = prof, shapes
    m(x)
    torch.cuda.synchronize()
    mmsa.ops.GEMM_PROFILE = mmsa.ops.GEMM_SHAPES = None
    for (f, e0, e1, _), sh in zip(prof, shapes):
        t = ctypes.c_float()
        lib.call("mmsa_event_elapsed_ms", e0, e1, ctypes.byref(t))
        a = agg.setdefault(sh, [0, 0.0, 0.0])
        a[0] += 1; a[1] += t.value; a[2] += f
tot = sum(a[1] for a in agg.values()) / 3
print(f"GEMM launches/step {sum(a[0] for a in agg.values()) // 3}, total {tot:.2f} ms/step")
for sh, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"M={sh[0]:6d} N={sh[1]:5d} K={sh[2]:5d} b={sh[3]:2d} act={sh[4]:5s} resid={int(sh[5])} out={sh[6]:2s} A={sh[7]:6s}  n/step {a[0] / 3:5.1f}  ms/step {a[1] / 3:6.3f}  avg us {a[1] / a[0] * 1e3:7.1f}  {a[2] / a[1] / 1e9:6.1f} TF")
