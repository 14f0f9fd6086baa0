"""Per-operator time of each neck level (single stream, eager, events around every C-ABI call): which operators make up the
part of the SPM that sits on the step's critical path (levels 1..3)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
from mmsa import lib
from tests.configs import CONFIGS, make_input
cfg = CONFIGS["vitl1024"]
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
x = make_input(cfg, batch=2).cuda()
m.multistream = False
for _ in range(2):
    m(x)
torch.cuda.synchronize()
B, H, W = 2, 1024, 1024
D = cfg["kwargs"]["embed_dim"]
sizes = [(H // 4, W // 4), (H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
Nc = sum(s[0] * s[1] for s in sizes[1:])
ws, pk, chans = m._ws, m._packed, m.channels
tcat = [ws.get(f"tcat{i}", B * sizes[i][0] * sizes[i][1], 2 * chans[i]) for i in range(4)]
tcat_p = [ws.planes(f"tcat{i}", B * sizes[i][0] * sizes[i][1], 2 * chans[i]) for i in range(4)]
cbuf = ws.get("c", B * Nc, D); c1 = ws.get("c1", B * sizes[0][0] * sizes[0][1], D)
offs = [0, 0, sizes[1][0] * sizes[1][1], sizes[1][0] * sizes[1][1] + sizes[2][0] * sizes[2][1]]
rec = []
orig = lib.call
def timed_call(name, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *args); e1.record()
    rec.append((name, args, e0, e1))
import mmsa.ops as ops_mod, mmsa.backbone as bb_mod
for mod in (lib, ops_mod.lib, bb_mod.ops.lib):
    mod.call = timed_call
for i in range(4):
    for rep in range(3):
        rec.clear()
        if i == 0:
            m._neck_level(0, pk["neck"][0], tcat[0], B, sizes[0][0], sizes[0][1], chans[0], c1, 0, tcat_p[0])
        else:
            m._neck_level(i, pk["neck"][i], tcat[i], B, sizes[i][0], sizes[i][1], chans[i], cbuf[offs[i]:], Nc * D, tcat_p[i])
        torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for name, args, e0, e1 in rec:
        key = name.replace("mmsa_", "")
        if key == "gconv_nhwc":
            key += f" k={args[12]} cin_g={args[10]}"
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
    tot = sum(a[1] for a in agg.values())
    print(f"== level {i}: {len(rec)} calls, {tot:.3f} ms")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {k:34s} x{a[0]:2d}  {a[1] * 1e3:8.1f} us")
