"""Time the windowed-attention kernel alone at the ViT-L shape (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
ops = mmsa.ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
vf = os.environ.get("MMSA_WATTN_VF", "1") == "1"    # h8 planes throughout: every contraction on the fp16 MFMA (the production form)
H, W, heads, hd, ws = 64, 64, 16, 64, 14
D = heads * hd
dev = "cuda"
x, brow = torch.randn(B * H * W, 3 * D, device=dev), torch.randn(1, 3 * D, device=dev)
pf = ops.FMT_H8 if vf else ops.FMT_B3
qkv = ops.split_planes(x, fmt=pf)
bias = ops.split_planes(brow, kpad=3 * D, fmt=pf)
relp = ops.window_relpos_planes(torch.randn(27, hd, device=dev) * 0.3, torch.randn(27, hd, device=dev) * 0.3, ws, fmt=pf)
out = ops.alloc_planes(B * H * W, D, dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
best = 1e9
for rnd in range(3):
    for _ in range(3):
        ops.window_attention(qkv, bias, relp, out, B, H, W, heads, hd, ws, hd ** -0.5)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        ops.window_attention(qkv, bias, relp, out, B, H, W, heads, hd, ws, hd ** -0.5)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / n * 1e3)
print(f"wattn B={B} vf={vf}: {best:.1f} us per launch")
if "--stamps" in sys.argv:
    # debug-knob build only (MMSA_LIB = a library built with tools/build_variant.sh ... wattn.hip -DMMSA_DEBUG_KNOBS): shader-clock stamps of
    # workgroup 0, waves 0 and 6, per item: 0 loop top | 1 Q / K landed | 2 rel-pos terms done | 3 past barrier 1 | 4 scores done | 5 past
    # barrier 2 | 6 softmax maxima done | 7 V landed | 8 past barrier 3 | 9 P V + stores done
    import ctypes
    L = ctypes.CDLL(os.environ["MMSA_LIB"])
    st = torch.zeros(2 * 8 * 16, dtype=torch.int64, device=dev)
    L.mmsa_debug_wattn_stamps(ctypes.c_void_p(st.data_ptr()))
    ops.window_attention(qkv, bias, relp, out, B, H, W, heads, hd, ws, hd ** -0.5)
    torch.cuda.synchronize()
    L.mmsa_debug_wattn_stamps(ctypes.c_void_p(0))
    t = st.cpu().view(2, 8, 16)
    names = ["wait Q/K", "rel-pos", "barrier 1", "scores", "barrier 2", "maxima", "wait V", "barrier 3", "PV+store", "next Q .. top"]
    for wv, wname in ((0, "wave 0"), (1, "wave 6")):
        for i in range(8):
            r = t[wv, i]
            if r[9] == 0:
                continue
            d = [int(r[k + 1] - r[k]) for k in range(9)]
            nxt = int(t[wv, i + 1][0] - r[9]) if i + 1 < 8 and t[wv, i + 1][0] else 0
            print(f"{wname} item {i}: " + " | ".join(f"{n} {v}" for n, v in zip(names, d + [nxt])) + f" | total {int(r[9] - r[0]) + nxt}")
