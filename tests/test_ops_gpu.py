"""GPU parity tests, one per HIP kernel family, each against the oracle / a plain fp32 torch statement of the
same reference op computed on the CPU.  All calls go through the C ABI (mmsa.ops -> libmmsa_hip.so).
Tolerance: tests.util.REL_TOL = 1e-3 (north_star); integer/index work is checked bit-exactly."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_encoder as R
from tests.weights import seeded_state_dict
from tests.util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import mmsa
    return mmsa.ops


def g(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 96, 64), (1000, 576, 1024), (4096, 1024, 768), (77, 27, 96), (8192, 256, 4096)])
def test_gemm_plain(ops, M, N, K):
    a = torch.randn(M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / K ** 0.5
    b = torch.randn(N, generator=g(3))
    ref = F.linear(a.double(), w.double(), b.double()).float()
    pl = ops.split_planes(w.to(DEV))
    out = torch.empty(M, N, device=DEV)
    ops.gemm(a.to(DEV), pl, out, bias=b.to(DEV))
    r, _ = assert_close(out, ref, what=f"gemm {M}x{N}x{K}")
    assert r < 1e-4  # split3 keeps ~16 mantissa bits


def test_gemm_epilogues(ops):
    M, N, K = 515, 192, 96
    a = torch.randn(M, K, generator=g(4))
    w = torch.randn(N, K, generator=g(5)) / K ** 0.5
    b = torch.randn(N, generator=g(6))
    cs = torch.randn(N, generator=g(7))
    res = torch.randn(M, N, generator=g(8))
    pl = ops.split_planes(w.to(DEV))
    lin = F.linear(a, w, b)
    for act, fn in (("gelu", F.gelu), ("relu", F.relu), ("relu6", F.relu6), ("hswish", lambda t: t * F.relu6(t + 3) / 6),
                    ("sigmoid", torch.sigmoid), ("none", lambda t: t)):
        ref = 0.7 * res + cs * 1.3 * fn(lin)
        out = torch.empty(M, N, device=DEV)
        ops.gemm(a.to(DEV), pl, out, bias=b.to(DEV), act=act, alpha=1.3, colscale=cs.to(DEV), resid=res.to(DEV), beta=0.7)
        assert_close(out, ref, what=f"epilogue {act}")
    # in-place residual, strided views, broadcast residual
    big = torch.randn(M, 2 * N, generator=g(9)).to(DEV)
    ref = big[:, N:].cpu() + lin
    ops.gemm(a.to(DEV), pl, big[:, N:], bias=b.to(DEV), resid=big[:, N:])
    assert_close(big[:, N:], ref, what="in-place strided")
    pos = torch.randn(103, N, generator=g(10))
    out = torch.empty(M, N, device=DEV)
    ops.gemm(a.to(DEV), pl, out, bias=b.to(DEV), resid=pos.to(DEV), resid_mod=103)
    assert_close(out, lin + pos[torch.arange(M) % 103], what="resid_mod")


def test_gemm_batched_and_pixel_shuffle(ops):
    B, M, N, K = 3, 200, 64, 64
    a = torch.randn(B, M, K, generator=g(11))
    w = torch.randn(B, N, K, generator=g(12)) / 8
    ref = torch.bmm(a, w.transpose(1, 2))
    pls = ops.split_planes(w.reshape(B * N, K).to(DEV))
    pls.n = N
    out = torch.empty(B * M, N, device=DEV)
    ops.gemm(a.reshape(B * M, K).to(DEV), pls, out, batch=B, m=M, stride_a=M * K, stride_w=N * 2 * K, stride_c=M * N)
    assert_close(out.view(B, M, N), ref, what="batched")
    # ConvTranspose2d(C, C, 2, 2) as GEMM + pixel shuffle (BK:55,324)
    Bc, C, H, W = 2, 32, 5, 7
    ct = torch.nn.ConvTranspose2d(C, C, 2, 2)
    x = torch.randn(Bc, C, H, W, generator=g(13))
    add = torch.randn(Bc, C, 2 * H, 2 * W, generator=g(14))
    ref = (ct(x) + add).detach()
    wp = ct.weight.detach().permute(2, 3, 1, 0).reshape(4 * C, C)
    pl = ops.split_planes(wp.contiguous().to(DEV))
    tok = x.permute(0, 2, 3, 1).reshape(Bc * H * W, C).contiguous().to(DEV)
    c1 = add.permute(0, 2, 3, 1).reshape(Bc * 4 * H * W, C).contiguous().to(DEV)
    ops.gemm(tok, pl, c1, bias=ct.bias.detach().repeat(4).to(DEV), resid=c1, batch=Bc, m=H * W, stride_a=H * W * C,
             stride_r=4 * H * W * C, stride_c=4 * H * W * C, pixel_shuffle=(H, W, C))
    assert_close(c1.view(Bc, 2 * H, 2 * W, C).permute(0, 3, 1, 2), ref, what="pixel shuffle")


def test_gemm_rejects_bad_arguments(ops):
    pl = ops.split_planes(torch.randn(8, 32).to(DEV))
    with pytest.raises(RuntimeError):
        ops.gemm(torch.randn(4, 32), pl, torch.empty(4, 8, device=DEV))  # CPU tensor
    with pytest.raises(RuntimeError):
        ops.gemm(torch.randn(4, 16).to(DEV), pl, torch.empty(4, 8, device=DEV))  # K mismatch


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("C", [32, 96, 192, 768, 1024, 1536, 4096])
def test_layernorm(ops, C):
    x = torch.randn(333, C, generator=g(20)) * 3 + 1
    w, b = torch.randn(C, generator=g(21)), torch.randn(C, generator=g(22))
    for eps in (1e-6, 1e-5):
        out = torch.empty(333, C, device=DEV)
        out2 = torch.empty(333, C, device=DEV)
        ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), eps, out, out2=out2)
        ref = F.layer_norm(x, (C,), w, b, eps)
        assert_close(out, ref, tol=1e-5, what="layernorm")
        assert_close(out2, ref + x, tol=1e-5, what="layernorm + x")


def test_layernorm_patchify(ops):
    B, H, W, C = 2, 6, 8, 64
    x = torch.randn(B, H, W, C, generator=g(23))
    w, b = torch.randn(C, generator=g(24)), torch.randn(C, generator=g(25))
    y = F.layer_norm(x, (C,), w, b, 1e-6)
    # im2col of a 2x2 s2 conv with K order (kh, kw, c)
    ref = y.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B * (H // 2) * (W // 2), 4 * C)
    out = torch.empty_like(ref, device=DEV)
    ops.layernorm(x.view(-1, C).to(DEV), w.to(DEV), b.to(DEV), 1e-6, out, patchify=(H, W))
    assert_close(out, ref, tol=1e-5, what="patchify")


def test_colstats_ffrm_lnhw(ops):
    """GFFM LayerNorm(H*W) (AM:265) followed by FFRM (AM:158-162), via colstats + ffrm_finalize + lnhw_apply."""
    B, H, W, C = 2, 24, 20, 64
    HW = H * W
    f = torch.randn(B, C, H, W, generator=g(30)) * 2 + 0.5
    ln = torch.nn.LayerNorm(HW)
    ln.weight.data = torch.randn(HW, generator=g(31)) * 0.2 + 1
    ln.bias.data = torch.randn(HW, generator=g(32)) * 0.1
    ffrm = R.FFRM(C)
    sd = seeded_state_dict(ffrm, 3)
    ffrm.load_state_dict(sd)
    with torch.no_grad():
        ref = ffrm(ln(f.view(B, C, HW)).view(B, C, H, W))
    x = f.permute(0, 2, 3, 1).reshape(B * HW, C).contiguous().to(DEV)
    st = torch.empty(B * 3, C, dtype=torch.float64, device=DEV)
    ops.colstats(x, HW * C, B, HW, st, wrow=ln.weight.data.to(DEV))
    s = st.view(B, 3, C).cpu()
    assert torch.allclose(s[:, 0], f.double().sum((2, 3)), rtol=1e-6, atol=1e-6)
    assert torch.allclose(s[:, 1], f.double().pow(2).sum((2, 3)), rtol=1e-6)
    mean, rstd, mult = (torch.empty(B, C, device=DEV) for _ in range(3))
    ops.ffrm_finalize(st, B, HW, C, float(ln.weight.double().mean()), float(ln.bias.double().mean()),
                      sd["conv_atten.conv.weight"].reshape(C, C).contiguous().to(DEV), sd["conv_atten.gn.weight"].to(DEV),
                      sd["conv_atten.gn.bias"].to(DEV), mean, rstd, mult, torch.empty(2 * B, C, device=DEV))
    out = torch.empty(B * HW, C, device=DEV)
    ops.lnhw_apply(x, mean, rstd, mult, ln.weight.data.to(DEV), ln.bias.data.to(DEV), out, B, HW)
    assert_close(out.view(B, H, W, C).permute(0, 3, 1, 2), ref, tol=1e-4, what="lnhw+ffrm")


# ------------------------------------------------------------------------------------------------ MSDA
def test_msda_reference_known_answer(ops, golden_dir):
    """The reference's own parity contract (ops/test.py:53-75): fp32, rtol 1e-2, atol 1e-3 -- and much tighter."""
    gd = np.load(os.path.join(golden_dir, "msda.npz"))
    args = [torch.from_numpy(gd[k]).to(DEV) for k in ("t_value", "t_shapes", "t_lsi", "t_loc", "t_aw")]
    out = ops.msda_forward(*args, im2col_step=2)
    ref = torch.from_numpy(gd["t_out"])
    assert torch.allclose(out.cpu(), ref, rtol=1e-2, atol=1e-3)
    assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=1e-7)
    with pytest.raises(RuntimeError):  # batch % im2col_step check (ms_deform_attn_cuda.cu:52)
        big = [torch.cat([a] * 3) if a.dtype == torch.float32 else a for a in args]
        ops.msda_forward(*big, im2col_step=2)


@pytest.mark.parametrize("tag", ["inj", "ext"])
def test_msda_golden(ops, golden_dir, tag):
    gd = np.load(os.path.join(golden_dir, "msda.npz"))
    args = [torch.from_numpy(gd[f"{tag}_{k}"]).to(DEV) for k in ("value", "shapes", "lsi", "loc", "aw")]
    out = ops.msda_forward(*args)
    assert_close(out, torch.from_numpy(gd[f"{tag}_out"]), tol=1e-5, what=f"msda {tag}")


def test_msda_fused_module(ops):
    """Whole MSDeformAttn.forward (ops/modules/ms_deform_attn.py:83-130) through value/offset GEMMs + fused gather."""
    B, D, M, Pn, L = 2, 128, 4, 4, 3
    shapes = [(16, 12), (8, 6), (4, 3)]
    S = sum(h * w for h, w in shapes)
    Lq = 8 * 6
    mod = R.MSDeformAttn(D, L, M, Pn, 0.5)
    sd = seeded_state_dict(mod, 5)
    sd["sampling_offsets.weight"] *= 4  # push samples across the borders
    mod.load_state_dict(sd)
    q = torch.randn(B, Lq, D, generator=g(40))
    feat = torch.randn(B, S, D, generator=g(41))
    ss = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    ref_pts = R.get_reference_points([(8, 6)], torch.float32)  # [1, Lq, 1, 2]
    with torch.no_grad():
        ref = mod(q, ref_pts, feat, ss, lsi)
    dv = D // 2
    oa = ops.split_planes(torch.cat([sd["sampling_offsets.weight"], sd["attention_weights.weight"]]).to(DEV))
    oab = torch.cat([sd["sampling_offsets.bias"], sd["attention_weights.bias"]]).to(DEV)
    val = torch.empty(B * S, dv, device=DEV)
    ops.gemm(feat.view(-1, D).to(DEV), ops.split_planes(sd["value_proj.weight"].to(DEV)), val, bias=sd["value_proj.bias"].to(DEV))
    raw = torch.empty(B * Lq, oa.n, device=DEV)
    ops.gemm(q.view(-1, D).to(DEV), oa, raw, bias=oab)
    samp = torch.empty(B * Lq, dv, device=DEV)
    ops.msda_fused(val, ss.to(DEV), lsi.to(DEV), raw, ref_pts.view(Lq, 2).contiguous().to(DEV), samp, B, S, M, dv // M, L, Lq, Pn)
    out = torch.empty(B * Lq, D, device=DEV)
    ops.gemm(samp, ops.split_planes(sd["output_proj.weight"].to(DEV)), out, bias=sd["output_proj.bias"].to(DEV))
    assert_close(out.view(B, Lq, D), ref, what="MSDeformAttn module")


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("H,W,heads,hd,ws,table", [
    (16, 16, 2, 32, 14, 27),    # 16 -> 28 padding, 4 windows
    (14, 14, 2, 32, 14, 27),    # no padding
    (20, 20, 2, 64, 14, 27),    # 20 -> 28 padding
    (64, 64, 2, 64, 14, 27),    # the ViT-L window geometry (64 -> 70)
    (16, 16, 2, 32, 0, 31),     # global
    (14, 14, 2, 32, 0, 31),     # global with rel-pos interpolation (31 -> 27)
    (20, 12, 3, 64, 0, 39),     # global, non-square, odd key count
    (64, 64, 2, 64, 0, 127),    # the ViT-L global geometry (4096 keys)
])
def test_attention(ops, H, W, heads, hd, ws, table):
    import mmsa.backbone as bb
    B, D = 2, heads * hd
    att = R.Attention(D, heads, (table // 2 + 1, table // 2 + 1))
    sd = seeded_state_dict(att, 7)
    sd["rel_pos_h"] = torch.randn(table, hd, generator=g(50)) * 0.3
    sd["rel_pos_w"] = torch.randn(table, hd, generator=g(51)) * 0.3
    sd["qkv.bias"] = torch.randn(3 * D, generator=g(52)) * 0.5  # pad tokens attend with k = v = bias
    att.load_state_dict(sd)
    x = torch.randn(B, H, W, D, generator=g(53))
    with torch.no_grad():
        if ws:
            xw, pad_hw = R.window_partition(x, ws)
            ref = R.window_unpartition(att(xw), ws, pad_hw, (H, W))
        else:
            ref = att(x)
        ref_noproj = None
    # device path: qkv GEMM on real tokens only, rel-pos prepass, fused attention, proj GEMM
    T = H * W
    qkv = torch.empty(B * T, 3 * D, device=DEV)
    ops.gemm(x.view(-1, D).to(DEV), ops.split_planes(sd["qkv.weight"].to(DEV)), qkv, bias=sd["qkv.bias"].to(DEV))
    if ws:
        rh, rw = bb._rel_table(ws, sd["rel_pos_h"].to(DEV)), bb._rel_table(ws, sd["rel_pos_w"].to(DEV))
        kk = 2 * ws
    else:
        rh, rw = bb._rel_table(H, sd["rel_pos_h"].to(DEV)), bb._rel_table(W, sd["rel_pos_w"].to(DEV))
        kk = H + W
    rp = torch.empty(B * heads * T, kk, device=DEV)
    ops.relpos_bias(qkv, rh, rw, rp, B, H, W, heads, hd, ws)
    ao = torch.empty(B * T, D, device=DEV)
    ops.attention(qkv, sd["qkv.bias"].to(DEV), rp, ao, B, H, W, heads, hd, ws, hd ** -0.5)
    out = torch.empty(B * T, D, device=DEV)
    ops.gemm(ao, ops.split_planes(sd["proj.weight"].to(DEV)), out, bias=sd["proj.bias"].to(DEV))
    assert_close(out.view(B, H, W, D), ref, what=f"attention {H}x{W} ws={ws}")


def test_relpos_terms_exact(ops):
    """rel_h / rel_w einsums (IE:616-617) in fp32 on the device vs torch."""
    import mmsa.backbone as bb
    B, H, W, heads, hd = 1, 16, 16, 2, 32
    qkv = torch.randn(B * H * W, 3 * heads * hd, generator=g(60))
    tab_h, tab_w = torch.randn(31, hd, generator=g(61)), torch.randn(31, hd, generator=g(62))
    Rh, Rw = R.get_rel_pos(H, H, tab_h), R.get_rel_pos(W, W, tab_w)
    q = qkv[:, :heads * hd].view(B, H, W, heads, hd).permute(0, 3, 1, 2, 4)
    rel_h = torch.einsum("bnhwc,hkc->bnhwk", q, Rh)
    rel_w = torch.einsum("bnhwc,wkc->bnhwk", q, Rw)
    ref = torch.cat([rel_h, rel_w], -1).reshape(B * heads * H * W, H + W)
    rp = torch.empty(B * heads * H * W, H + W, device=DEV)
    ops.relpos_bias(qkv.to(DEV), bb._rel_table(H, tab_h.to(DEV)), bb._rel_table(W, tab_w.to(DEV)), rp, B, H, W, heads, hd, 0)
    assert_close(rp, ref, tol=1e-5, what="relpos")


# ------------------------------------------------------------------------------------------------ convs
def test_dwconv_and_gconv(ops):
    B, H, W = 2, 13, 10
    for C, k, bias, act, fn in ((96, 7, True, "none", lambda t: t), (64, 3, False, "relu6", F.relu6), (16, 3, True, "gelu", F.gelu)):
        conv = torch.nn.Conv2d(C, C, k, padding=k // 2, groups=C, bias=bias)
        x = torch.randn(B, C, H, W, generator=g(70))
        ref = fn(conv(x)).detach()
        w = conv.weight.detach().reshape(C, k * k).t().contiguous().to(DEV)
        out = torch.empty(B * H * W, C, device=DEV)
        ops.dwconv(x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV), w, conv.bias.detach().to(DEV) if bias else None,
                   out, B, H, W, k, act=act)
        assert_close(out.view(B, H, W, C).permute(0, 3, 1, 2), ref, tol=1e-5, what=f"dwconv k={k}")
    for cin, cout, groups, k in ((96, 288, 32, 1), (288, 288, 32, 3), (128, 128, 64, 3)):
        conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2, groups=groups, bias=False)
        x = torch.randn(B, cin, H, W, generator=g(71))
        ref = conv(x).detach()
        cig, cog = cin // groups, cout // groups
        w = conv.weight.detach().reshape(groups, cog, cig, k * k).permute(0, 3, 2, 1).contiguous().to(DEV)
        out = torch.empty(B * H * W, cout, device=DEV)
        ops.gconv(x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(DEV), w, None, out, B, H, W, groups, cig, cog, k)
        assert_close(out.view(B, H, W, cout).permute(0, 3, 1, 2), ref, tol=1e-5, what=f"gconv g={groups} k={k}")


def test_patchify_convs_as_gemm(ops):
    """PatchEmbed 16x16 s16 (IE:658-663) and the ConvNeXt stem 4x4 s4 on the aux channels (TC:307-316)."""
    x = torch.randn(2, 6, 64, 96, generator=g(80))
    for c0, p, cout in ((0, 16, 64), (3, 4, 32)):
        conv = torch.nn.Conv2d(3, cout, p, stride=p)
        ref = conv(x[:, c0:c0 + 3]).detach()
        pl = ops.split_planes(conv.weight.detach().reshape(cout, -1).contiguous().to(DEV))
        hp, wp = 64 // p, 96 // p
        a = torch.empty(2 * hp * wp, pl.kpad, device=DEV)
        ops.im2col_nchw(x.to(DEV), c0, 3, p, a)
        out = torch.empty(2 * hp * wp, cout, device=DEV)
        ops.gemm(a, pl, out, bias=conv.bias.detach().to(DEV))
        assert_close(out.view(2, hp, wp, cout).permute(0, 3, 1, 2), ref, what=f"patchify p={p}")


# ------------------------------------------------------------------------------------------------ neck pieces
def test_gram_tn(ops):
    B, P, c = 2, 1500, 80
    x = torch.randn(B, P, 2 * c, generator=g(90))
    ref = torch.einsum("bpi,bpj->bij", x[..., :c].double(), x[..., c:].double()).float()
    xd = x.view(B * P, 2 * c).to(DEV)
    out = torch.empty(B * c, c, device=DEV, dtype=torch.float64)
    ops.gram_tn(xd[:, :c], xd[:, c:], P * 2 * c, out, B, P, nblk=1)
    assert_close(out.view(B, c, c).float(), ref, tol=1e-5, what="gram full")
    # the per-slice scratch is the caller's: too small a buffer is refused; a sufficient one gives the same bits as the op's own
    need = ops.gram_tn_scratch_bytes(B, P, c)
    assert need == B * 1 * ((P + 255) // 256) * 36 * 64 * 16
    with pytest.raises(RuntimeError):
        ops.gram_tn(xd[:, :c], xd[:, c:], P * 2 * c, out.clone(), B, P, nblk=1, scratch=torch.empty(need // 4 - 64, device=DEV))
    out2 = torch.full_like(out, float("nan"))
    ops.gram_tn(xd[:, :c], xd[:, c:], P * 2 * c, out2, B, P, nblk=1, scratch=torch.empty(need // 4 + 5, device=DEV))
    assert torch.equal(out2, out)
    ops.gram_tn(xd[:, :c], xd[:, c:], P * 2 * c, out, B, P, nblk=8)  # only diagonal head blocks are defined
    ch = c // 8
    for h in range(8):
        sl = slice(h * ch, (h + 1) * ch)
        assert_close(out.view(B, c, c)[:, sl, sl].float(), ref[:, sl, sl], tol=1e-5, what="gram diag block")


@pytest.mark.parametrize("c,H,W", [(32, 12, 10), (96, 16, 16)])
def test_gfe_module(ops, c, H, W):
    """GFE (AM:133-145): LN -> grouped qkv convs -> L2-normalised channel attention -> proj, as the backbone runs it."""
    B, HW = 2, H * W
    mod = R.GFE(c)
    sd = seeded_state_dict(mod, 11)
    mod.load_state_dict(sd)
    x = torch.randn(B, c, H, W, generator=g(91))
    with torch.no_grad():
        ref = mod(x)
    X = x.permute(0, 2, 3, 1).reshape(B * HW, c).contiguous().to(DEV)
    y, s = torch.empty_like(X), torch.empty_like(X)
    ops.layernorm(X, sd["norm1.body.weight"].to(DEV), sd["norm1.body.bias"].to(DEV), 1e-5, y, out2=s)
    G = 32
    q1w = sd["attn.qkv1.weight"].reshape(G, 3 * c // G, c // G, 1).permute(0, 3, 2, 1).contiguous().to(DEV)
    q2w = sd["attn.qkv2.weight"].reshape(G, 3 * c // G, 3 * c // G, 9).permute(0, 3, 2, 1).contiguous().to(DEV)
    q1 = torch.empty(B * HW, 3 * c, device=DEV)
    q2 = torch.empty(B * HW, 3 * c, device=DEV)
    ops.gconv(y, q1w, None, q1, B, H, W, G, c // G, 3 * c // G, 1)
    ops.gconv(q1, q2w, None, q2, B, H, W, G, 3 * c // G, 3 * c // G, 3)
    st = torch.empty(B * 3, 3 * c, dtype=torch.float64, device=DEV)
    ops.colstats(q2, HW * 3 * c, B, HW, st)
    gm = torch.empty(B * c, c, device=DEV, dtype=torch.float64)
    ops.gram_tn(q2[:, :c], q2[:, c:2 * c], HW * 3 * c, gm, B, HW, nblk=8)
    pl = ops.Planes(torch.zeros(B * c, 2 * c, dtype=torch.int16, device=DEV), c, c, c)
    base = st.data_ptr()
    ops.chanattn_build(gm, base + 8 * 3 * c, 9 * c, base + 8 * 4 * c, 9 * c, sd["attn.scale"].reshape(8).contiguous().to(DEV),
                       sd["attn.proj.weight"].reshape(c, c).contiguous().to(DEV), pl, B, c, 8)
    out = torch.empty(B * HW, c, device=DEV)
    ops.gemm(q2[:, 2 * c:], pl, out, alpha=float(sd["attn.scale2"]), resid=s, batch=B, m=HW, stride_a=HW * 3 * c,
             stride_w=c * 2 * c, stride_r=HW * c, stride_c=HW * c)
    assert_close(out.view(B, H, W, c).permute(0, 3, 1, 2), ref, what="GFE")


def test_gffm_gemms(ops):
    B, c, H, W = 2, 64, 10, 12
    HW = H * W
    gx, gy = 0.7, -0.4
    gin = torch.randn(B, 2 * c, H, W, generator=g(92)) * 0.3
    x, y = gin[:, :c].reshape(B, c, HW), gin[:, c:].reshape(B, c, HW)
    ax = F.softmax(torch.bmm(x, y.transpose(1, 2)), -1)
    ay = F.softmax(torch.bmm(y, x.transpose(1, 2)), -1)
    ref = torch.cat((gx * torch.bmm(ax, y) + x, gy * torch.bmm(ay, x) + y), 1)  # before the LayerNorm
    gc = gin.permute(0, 2, 3, 1).reshape(B * HW, 2 * c).contiguous().to(DEV)
    e = torch.empty(B * c, c, device=DEV, dtype=torch.float64)
    ops.gram_tn(gc[:, :c], gc[:, c:], HW * 2 * c, e, B, HW)
    mk = lambda: ops.Planes(torch.zeros(B * c, 2 * c, dtype=torch.int16, device=DEV), c, c, c)
    px, py = mk(), mk()
    ops.gffm_build(e, px, py, B, c)
    f = torch.empty(B * HW, 2 * c, device=DEV)
    kw = dict(batch=B, m=HW, stride_a=HW * 2 * c, stride_w=c * 2 * c, stride_r=HW * 2 * c, stride_c=HW * 2 * c)
    ops.gemm(gc[:, c:], px, f[:, :c], alpha=gx, resid=gc[:, :c], **kw)
    ops.gemm(gc[:, :c], py, f[:, c:], alpha=gy, resid=gc[:, c:], **kw)
    assert_close(f.view(B, HW, 2 * c).permute(0, 2, 1), ref, what="GFFM energies")


def test_coordinate_attention_and_gate(ops):
    B, C, H, W = 2, 64, 9, 11
    ca = R.CA(C)
    sd = seeded_state_dict(ca, 13)
    ca.load_state_dict(sd)
    ca.eval()
    x = torch.randn(B, C, H, W, generator=g(93))
    with torch.no_grad():
        ref = ca(x)
    z = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV)
    pooled = torch.empty(B * (H + W), C, device=DEV)
    ops.pool_hw(z, pooled, B, H, W)
    p = "coord_atten."
    mip = sd[p + "conv1.weight"].shape[0]
    inv = sd[p + "bn1.weight"] / torch.sqrt(sd[p + "bn1.running_var"] + 1e-5)
    w1 = sd[p + "conv1.weight"].reshape(mip, C) * inv[:, None]
    b1 = (sd[p + "conv1.bias"] - sd[p + "bn1.running_mean"]) * inv + sd[p + "bn1.bias"]
    y1 = torch.zeros(B * (H + W), 32, device=DEV)
    ops.gemm(pooled, ops.split_planes(w1.to(DEV)), y1, bias=b1.to(DEV), act="hswish")
    att = torch.empty(B * (H + W), C, device=DEV)
    ops.gemm(y1, ops.split_planes(sd[p + "conv_h.weight"].reshape(C, mip).to(DEV)), att, bias=sd[p + "conv_h.bias"].to(DEV),
             act="sigmoid", batch=B, m=H, stride_a=(H + W) * 32, stride_c=(H + W) * C)
    ops.gemm(y1[H:], ops.split_planes(sd[p + "conv_w.weight"].reshape(C, mip).to(DEV)), att[H:], bias=sd[p + "conv_w.bias"].to(DEV),
             act="sigmoid", batch=B, m=W, stride_a=(H + W) * 32, stride_c=(H + W) * C)
    out = torch.empty_like(z)
    ops.ca_apply(z, att, out, B, H, W)
    assert_close(out.view(B, H, W, C).permute(0, 3, 1, 2), ref, what="CoordinateAttention")
    # gated MLP elementwise part
    hd = torch.randn(50, 2 * C, generator=g(94))
    o = torch.empty(50, C, device=DEV)
    ops.gelu_gate(hd.to(DEV), o, C)
    assert_close(o, F.gelu(hd[:, :C]) * hd[:, C:], tol=1e-5, what="gelu gate")


# ------------------------------------------------------------------------------------------------ tail
@pytest.mark.parametrize("scale", [4, 2, 1, 0.5])
def test_tail_fuse(ops, scale):
    B, C, Hx, Wx = 2, 96, 8, 6
    Hc, Wc = int(Hx * scale), int(Wx * scale)
    xt = torch.randn(B, C, Hx, Wx, generator=g(100))
    cm = torch.randn(B, C, Hc, Wc, generator=g(101))
    bn = torch.nn.BatchNorm2d(C).eval()
    bn.weight.data, bn.bias.data = torch.randn(C, generator=g(102)), torch.randn(C, generator=g(103))
    bn.running_mean, bn.running_var = torch.randn(C, generator=g(104)), torch.rand(C, generator=g(105)) + 0.5
    up = xt if scale == 1 else F.interpolate(xt, scale_factor=scale, mode="bilinear", align_corners=False)
    with torch.no_grad():
        ref = bn(cm + up)
    inv = bn.weight.data / torch.sqrt(bn.running_var + 1e-5)
    out = torch.empty(B, C, Hc, Wc, device=DEV)
    ops.tail_fuse(cm.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV), Hc * Wc * C,
                  xt.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV), inv.to(DEV),
                  (bn.bias.data - bn.running_mean * inv).to(DEV), out, B, Hc, Wc, Hx, Wx)
    assert_close(out, ref, tol=1e-5, what=f"tail x{scale}")


@pytest.mark.parametrize("cg,H,W", [(9, 19, 33), (18, 16, 16), (36, 21, 17), (72, 32, 32), (3, 7, 5), (40, 9, 9)])
def test_gconv3_on_the_matrix_pipe(ops, cg, H, W):
    """Grouped 3x3 conv (groups 32, cin_g = cout_g) as an fp32-MFMA implicit GEMM: the GFE qkv2 widths of the four neck levels,
    partial tiles, channel counts that are not multiples of the 8-channel chunk or of the 16-column n-tile."""
    B, groups = 2, 32
    conv = torch.nn.Conv2d(cg * groups, cg * groups, 3, padding=1, groups=groups, bias=False)
    x = torch.randn(B, cg * groups, H, W, generator=g(140))
    ref = conv(x.double().float()).detach()
    ref64 = F.conv2d(x.double(), conv.weight.detach().double(), padding=1, groups=groups).float()
    w = conv.weight.detach().reshape(groups, cg, cg, 9).permute(0, 3, 2, 1).contiguous().to(DEV)
    out = torch.full((B * H * W, cg * groups), float("nan"), device=DEV)
    ops.gconv(x.permute(0, 2, 3, 1).reshape(-1, cg * groups).contiguous().to(DEV), w, None, out, B, H, W, groups, cg, cg, 3)
    assert_close(out.view(B, H, W, -1).permute(0, 3, 1, 2), ref64, tol=1e-5, what=f"gconv 3x3 cin_g=cout_g={cg}")
    again = torch.empty_like(out)
    ops.gconv(x.permute(0, 2, 3, 1).reshape(-1, cg * groups).contiguous().to(DEV), w, None, again, B, H, W, groups, cg, cg, 3)
    assert torch.equal(out, again)


@pytest.mark.parametrize("fmt_name", ["b3", "f3"])
@pytest.mark.parametrize("C,M", [(96, 1000), (96, 128), (96, 70000), (96, 33)])
def test_convnext_mlp_fused(C, M, fmt_name):
    """mmsa_convnext_mlp_fused (TC:107-132 pointwise_conv1 -> GELU -> pointwise_conv2 -> gamma -> + residual, two batched streams)
    against fp64 torch and against the two separate GEMM launches it replaces."""
    import mmsa
    ops = mmsa.ops
    g = torch.Generator().manual_seed(90 + C)
    b = 2
    a = torch.randn(b * M, C, generator=g)
    w1 = torch.randn(b * 4 * C, C, generator=g) / C ** 0.5
    w2 = torch.randn(b * C, 4 * C, generator=g) / (4 * C) ** 0.5
    b1, b2, gam = torch.randn(b * 4 * C, generator=g), torch.randn(b * C, generator=g), torch.randn(b * C, generator=g)
    x0 = torch.randn(b * M, C, generator=g)
    ref = torch.empty_like(x0)
    for s in range(b):
        h = F.gelu(a[s * M:(s + 1) * M].double() @ w1[s * 4 * C:(s + 1) * 4 * C].double().t() + b1[s * 4 * C:(s + 1) * 4 * C].double())
        y = h @ w2[s * C:(s + 1) * C].double().t() + b2[s * C:(s + 1) * C].double()
        ref[s * M:(s + 1) * M] = (x0[s * M:(s + 1) * M].double() + gam[s * C:(s + 1) * C].double() * y).float()
    fmt = ops.FMT_F3 if fmt_name == "f3" else ops.FMT_B3      # f3: fp16 hi/lo pairs, 22 significant bits (round 4: the model's default here)
    ap = ops.split_planes(a.to(DEV), kpad=ops.pad32(C), fmt=fmt)
    w1p = ops.split_planes(w1.to(DEV), fmt=fmt); w1p = ops.Planes(w1p.p, 4 * C, C, w1p.kpad, fmt)
    w2p = ops.split_planes(w2.to(DEV), fmt=fmt); w2p = ops.Planes(w2p.p, C, 4 * C, w2p.kpad, fmt)
    x = x0.to(DEV).clone()
    ops.convnext_mlp_fused(ap, w1p, w2p, b1.to(DEV), b2.to(DEV), gam.to(DEV), x, M, batch=b, stride_a=M * 2 * ap.kpad,
                           stride_w1=4 * C * 2 * w1p.kpad, stride_w2=C * 2 * w2p.kpad, stride_x=M * C)
    assert_close(x, ref, tol=3e-5 if fmt == ops.FMT_B3 else 3e-6, what="fused ConvNeXt MLP vs fp64")
    # the pair of launches it replaces
    hb = ops.alloc_planes(b * M, 4 * C, DEV, fmt=fmt)
    x2 = x0.to(DEV).clone()
    ops.gemm(ap, w1p, bias=b1.to(DEV), act="gelu", out_planes=hb, batch=b, m=M, stride_a=M * 2 * ap.kpad, stride_w=4 * C * 2 * w1p.kpad,
             stride_bias=4 * C, stride_cp=M * 2 * hb.kpad)
    ops.gemm(hb, w2p, x2, bias=b2.to(DEV), colscale=gam.to(DEV), resid=x2, batch=b, m=M, stride_a=M * 2 * hb.kpad, stride_w=C * 2 * w2p.kpad,
             stride_bias=C, stride_r=M * C, stride_c=M * C)
    assert_close(x, x2, tol=5e-6, what="fused vs the two GEMM launches")


def test_gemm_f3_weights_fp32_activations():
    """The ConvNeXt stem (TC:297-304 as a patch-matrix GEMM): fp32 A split to fp16 hi/lo while staged, f3 weight planes, batched per stream."""
    import mmsa
    ops = mmsa.ops
    g = torch.Generator().manual_seed(191)
    M, K, N, b = 1000, 48, 96, 2
    a = torch.randn(b * M, 64, generator=g); a[:, K:] = 0
    w = torch.randn(b * N, K, generator=g) / K ** 0.5
    bias = torch.randn(b * N, generator=g)
    wp = ops.split_planes(w.to(DEV), kpad=64, fmt=ops.FMT_F3); wp = ops.Planes(wp.p, N, K, wp.kpad, ops.FMT_F3)
    out = torch.empty(b * M, N, device=DEV)
    ops.gemm(a.to(DEV), wp, out, bias=bias.to(DEV), batch=b, m=M, stride_a=M * 64, stride_w=N * 2 * 64, stride_bias=N, stride_c=M * N)
    ref = torch.cat([a[i * M:(i + 1) * M, :K].double() @ w[i * N:(i + 1) * N].double().t() + bias[i * N:(i + 1) * N].double() for i in range(b)], 0).float()
    assert_close(out, ref, tol=2e-6, what="fp32-A gemm on f3 weights")


def test_zero_bytes(ops):
    """mmsa_zero_bytes: exactly the named range is cleared, on the current stream; non-contiguous views are refused."""
    t = torch.full((4, 1000), 7.5, dtype=torch.float64, device=DEV)
    ops.zero_(t[1, 10:900])
    torch.cuda.synchronize()
    assert (t[1, 10:900] == 0).all() and (t[1, :10] == 7.5).all() and (t[1, 900:] == 7.5).all() and (t[0] == 7.5).all() and (t[2:] == 7.5).all()
    with pytest.raises(RuntimeError):
        ops.zero_(t[:, 3])
