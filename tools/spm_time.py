"""Wall time of the spatial-prior module alone vs the whole forward (HIP-graph replays, side streams on) -- GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
from tests.configs import CONFIGS, make_input
cfg = CONFIGS["vitl1024"]
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
x = make_input(cfg, batch=2).cuda()
for _ in range(2):
    m(x)
torch.cuda.synchronize()
B, H, W = 2, 1024, 1024
D = cfg["kwargs"]["embed_dim"]
Nc = (H // 8) ** 2 + (H // 16) ** 2 + (H // 32) ** 2
ws = m._ws
c1 = ws.get("c1", B * (H // 4) * (W // 4), D); cbuf = ws.get("c", B * Nc, D)

def graph_time(fn, n=10):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def _spm_joined():
    e = m._spm(x.contiguous().float(), B, H, W, c1, cbuf, Nc)
    torch.cuda.current_stream().wait_event(e)
t_spm = graph_time(_spm_joined)
t_all = graph_time(lambda: m(x))
m.multistream = False
t_spm1 = graph_time(_spm_joined)
t_all1 = graph_time(lambda: m(x))
print(f"SPM {t_spm:.2f} ms of {t_all:.2f} ms (side streams on);  SPM {t_spm1:.2f} ms of {t_all1:.2f} ms (single stream)")

# --- parts: the two ConvNeXt streams alone, and the four neck levels alone (side streams on)
m.multistream = True
sizes = [(H // 4, W // 4), (H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
chans = m.channels
tcat = [ws.get(f"tcat{i}", B * sizes[i][0] * sizes[i][1], 2 * chans[i]) for i in range(4)]
tcat_p = [ws.planes(f"tcat{i}", B * sizes[i][0] * sizes[i][1], 2 * chans[i]) for i in range(4)]
xf = x.contiguous().float()
pk = m._packed

def conv_only():
    m._twin_batched(xf, B, sizes, tcat, [], tcat_p)

offs = [0, 0, sizes[1][0] * sizes[1][1], sizes[1][0] * sizes[1][1] + sizes[2][0] * sizes[2][1]]
def neck_only(levels=(0, 1, 2, 3)):
    main = torch.cuda.current_stream()
    f = torch.cuda.Event(); f.record(main); joins = []
    for i in levels:
        sn = next(iter(m._sides.values()))[i]; sn.wait_event(f)
        with torch.cuda.stream(sn):
            out = c1 if i == 0 else cbuf[offs[i]:]
            m._neck_level(i, pk["neck"][i], tcat[i], B, sizes[i][0], sizes[i][1], chans[i], out, 0 if i == 0 else Nc * D, tcat_p[i])
            e = torch.cuda.Event(); e.record(sn); joins.append(e)
    for e in joins:
        main.wait_event(e)

print(f"ConvNeXt (both streams batched) {graph_time(conv_only):.2f} ms")
print(f"neck all levels {graph_time(neck_only):.2f} ms; " + ", ".join(f"L{i} {graph_time(lambda i=i: neck_only((i,))):.2f}" for i in range(4)))

def eager_time(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
m.multistream = True
print(f"eager: SPM {eager_time(_spm_joined):.2f} ms, conv {eager_time(conv_only):.2f} ms, neck {eager_time(neck_only):.2f} ms")

def both_independent():
    main = torch.cuda.current_stream()
    f = torch.cuda.Event(); f.record(main)
    sn = next(iter(m._sides.values()))[0]; sn.wait_event(f)
    with torch.cuda.stream(sn):
        m._neck_level(0, pk["neck"][0], tcat[0], B, sizes[0][0], sizes[0][1], chans[0], c1, 0, tcat_p[0])
        e = torch.cuda.Event(); e.record(sn)
    m._twin_batched(xf, B, sizes, tcat, [], tcat_p)
    main.wait_event(e)
print(f"conv + neck L0 forked at the root: graph {graph_time(both_independent):.2f} ms, eager {eager_time(both_independent):.2f} ms")
