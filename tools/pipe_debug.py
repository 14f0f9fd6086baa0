import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch
import mmsa
from tests.configs import CONFIGS, make_input
name = sys.argv[1] if len(sys.argv) > 1 else "vitl1024"
cfg = CONFIGS[name]
torch.manual_seed(1234)
m = mmsa.build_backbone(dict(type="SAMAdapterbimodalMixModNewInTwinConvNEW", **cfg["kwargs"]))
x = make_input(cfg, batch=2, seed=1234).to("cuda:0")
ref = [f.clone() for f in m(x)[0]]
if os.environ.get("PD_TWICE", "1") == "1":
    ref2 = [f.clone() for f in m(x)[0]]
    print("forward twice equal:", [torch.equal(a, b) for a, b in zip(ref, ref2)])
torch.cuda.synchronize()
assert m.forward_pipelined(x) is None
for it in range(int(os.environ.get("PD_ITERS", "4"))):
    outs = m.forward_pipelined(x)[0]
    torch.cuda.synchronize()
    print(it, [f"{((a - b).abs().max() / b.abs().max()).item():.2e}" for a, b in zip(outs, ref)], [torch.equal(a, b) for a, b in zip(outs, ref)])
# which stage is the victim?  SPM outputs of every overlapped call against those of the priming (non-overlapped) call
m._pl_cur = None
assert m.forward_pipelined(x) is None
torch.cuda.synchronize()
c_ref, c1_ref = m._pl_cur["c"].clone(), m._pl_cur["c1"].clone()
for it in range(int(os.environ.get("PD_ITERS", "4"))):
    outs = m.forward_pipelined(x)[0]
    torch.cuda.synchronize()
    c, c1 = m._pl_cur["c"], m._pl_cur["c1"]
    print("spm", it, "c equal", torch.equal(c, c_ref), "c1 equal", torch.equal(c1, c1_ref), "outs equal", [torch.equal(a, b) for a, b in zip(outs, ref)],
          f"c diff {((c - c_ref).abs().max() / c_ref.abs().max()).item():.2e} c1 diff {((c1 - c1_ref).abs().max() / c1_ref.abs().max()).item():.2e}")
    if not torch.equal(c, c_ref):
        d = (c != c_ref).view(2, -1, c.shape[1])
        n2, n3, n4 = 16384, 4096, 1024
        for b in range(2):
            for name, lo, hi in (("c2", 0, n2), ("c3", n2, n2 + n3), ("c4", n2 + n3, n2 + n3 + n4)):
                dd = d[b, lo:hi]
                rows = dd.any(1).nonzero().flatten()
                cols = dd.any(0).nonzero().flatten()
                if rows.numel():
                    print(f"   image {b} {name}: {int(dd.sum())} elements differ, rows {int(rows.min())}..{int(rows.max())} ({rows.numel()} rows), cols {int(cols.min())}..{int(cols.max())} ({cols.numel()} cols)")
# first differing intermediate of the neck (buffers keep the values of the last SPM run)
names = sorted(n for n in m._ws.bufs if n.startswith("nk"))
m._pl_cur = None
assert m.forward_pipelined(x) is None
torch.cuda.synchronize()
snap = {n: m._ws.bufs[n].clone() for n in names}
found = 0
for it in range(int(os.environ.get("PD_ITERS", "4")) * 2):
    m.forward_pipelined(x)
    torch.cuda.synchronize()
    bad = [n for n in names if not torch.equal(m._ws.bufs[n].view(torch.int32) if m._ws.bufs[n].dtype == torch.float32 else m._ws.bufs[n], snap[n].view(torch.int32) if snap[n].dtype == torch.float32 else snap[n])]
    bad = [n for n in bad if not any(t in n for t in ("nk_att", "nk_pool", "nk_y1", "nk_zop"))]
    if bad:
        found += 1
        print("iter", it, "differing neck buffers:", bad)
        for n in bad:
            a_, b_ = m._ws.bufs[n].double(), snap[n].double()
            idx = (a_ != b_).nonzero().flatten()
            print(f"   {n}: {idx.numel()} of {a_.numel()} differ, first idx {int(idx[0])}, last {int(idx[-1])}, max abs diff {float((a_ - b_).abs().max()):.3e}")
            if n.endswith("nk_hg.pl"):
                A_, B_ = m._ws.bufs[n], snap[n]
                ii = (A_ != B_).nonzero().flatten()
                row0 = int(ii[0]) // 128 * 128
                def dec(t):
                    seg = t[row0:row0 + 128].view(2, 2, 32)   # [k-block][hi|lo][32]
                    f = lambda u: (u.to(torch.int32) << 16).view(torch.float32)
                    return (f(seg[:, 0]) + f(seg[:, 1])).flatten()
                va, vb = dec(A_), dec(B_)
                print("      idx", [int(v) - row0 for v in ii[:8]], "... values now / snapshot (first 8 differing channels):")
                ch = (va != vb).nonzero().flatten()
                print("      channels", ch.tolist())
                print("      now ", [f"{float(va[c]):.6e}" for c in ch[:6]])
                print("      snap", [f"{float(vb[c]):.6e}" for c in ch[:6]])
        if found >= 3:
            break
