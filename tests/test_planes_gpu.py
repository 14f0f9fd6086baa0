"""GPU parity of the activation-planes paths (bf16 hi/lo pairs written by producers, consumed by the GEMM and
attention kernels) against fp32 torch / the oracle.  Same tolerance as the fp32-input paths."""
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


def planes_to_float(p):
    import mmsa
    return mmsa.ops.planes_to_float(p)


def test_split_planes_roundtrip(ops):
    x = torch.randn(37, 50, generator=g(1)) * 3
    p = ops.split_planes(x.to(DEV))
    assert p.p.shape == (37, 128) and p.kpad == 64 and p.k == 50
    back = ops.planes_to_float(p, cols=64).cpu()
    assert torch.all(back[:, 50:] == 0)
    assert ((back[:, :50] - x).abs() <= x.abs() * 2 ** -15).all()  # ~16 mantissa bits
    # interleaved layout: hi of element k at (k>>5)*64 + (k&31), lo 32 further
    raw = p.p.cpu()
    hi0 = (raw[:, 0].to(torch.int32) << 16).view(torch.float32)
    lo0 = (raw[:, 32].to(torch.int32) << 16).view(torch.float32)
    assert torch.equal(hi0 + lo0, back[:, 0])
    hi33 = (raw[:, 64 + 1].to(torch.int32) << 16).view(torch.float32)
    assert ((hi33 - x[:, 33]).abs() <= x[:, 33].abs() * 2 ** -8).all()


@pytest.mark.parametrize("M,N,K", [(300, 96, 64), (1000, 576, 1024), (4096, 1024, 768), (77, 28, 96)])
def test_gemm_planes_in_and_out(ops, M, N, K):
    a = torch.randn(M, K, generator=g(2))
    w = torch.randn(N, K, generator=g(3)) / K ** 0.5
    b = torch.randn(N, generator=g(4))
    res = torch.randn(M, N, generator=g(5))
    ref = F.gelu(F.linear(a.double(), w.double(), b.double())).float() + res
    ap = ops.split_planes(a.to(DEV), kpad=K)
    pl = ops.split_planes(w.to(DEV))
    out = torch.empty(M, N, device=DEV)
    outp = ops.alloc_planes(M, N, DEV)
    ops.gemm(ap, pl, out, bias=b.to(DEV), act="gelu", resid=res.to(DEV), out_planes=outp)
    assert_close(out, ref, what="planes-in gemm fp32 out")
    assert_close(planes_to_float(outp), ref, what="planes-in gemm planes out")
    # planes-only output from an fp32 A
    outp2 = ops.alloc_planes(M, N, DEV)
    ops.gemm(a.to(DEV), pl, bias=b.to(DEV), act="gelu", resid=res.to(DEV), out_planes=outp2)
    assert_close(planes_to_float(outp2), ref, what="fp32-in gemm planes out")


def test_layernorm_planes_and_patchify(ops):
    x = torch.randn(2 * 6 * 8, 64, generator=g(6)) * 2 + 0.3
    w, b = torch.randn(64, generator=g(7)), torch.randn(64, generator=g(8))
    y = F.layer_norm(x, (64,), w, b, 1e-6)
    p = ops.alloc_planes(x.shape[0], 64, DEV)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, out_planes=p)
    assert_close(planes_to_float(p), y, tol=5e-5, what="LN planes")
    ref = y.view(2, 3, 2, 4, 2, 64).permute(0, 1, 3, 2, 4, 5).reshape(2 * 3 * 4, 256)
    pp = ops.alloc_planes(24, 256, DEV)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, out_planes=pp, patchify=(6, 8))
    assert_close(planes_to_float(pp), ref, tol=5e-5, what="LN patchify planes")


def _ref_max_logit(att, xin, hw, live=None):
    """max |scale * q.k + rel-pos terms| of the oracle's Attention on xin [N, h, w, D] (IE:488-495), rows of `live` == False excluded."""
    h, w = hw
    heads = att.num_heads
    with torch.no_grad():
        qkv = att.qkv(xin).reshape(xin.shape[0], h * w, 3, heads, -1).permute(2, 0, 3, 1, 4)
        q, k, _ = qkv.reshape(3, xin.shape[0] * heads, h * w, -1).unbind(0)
        lg = (q * att.scale) @ k.transpose(-2, -1)
        lg = R.add_decomposed_rel_pos(lg, q, att.rel_pos_h, att.rel_pos_w, (h, w), (h, w)).abs()
        if live is not None:   # [N, h*w] bool
            lg = lg * live[:, None, :, None].expand(-1, heads, -1, -1).reshape(-1, h * w, 1)
        return lg.max().item()


def _live_queries(B, H, W, ws, pad_hw):
    Hp_, Wp_ = pad_hw
    live = torch.zeros(Hp_, Wp_, dtype=torch.bool)
    live[:H, :W] = True
    live = live.view(Hp_ // ws, ws, Wp_ // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    return live[None].expand(B, -1, -1).reshape(-1, ws * ws)


@pytest.mark.parametrize("H,W,heads,hd,ws,table", [
    (16, 16, 2, 32, 14, 27), (20, 20, 2, 64, 14, 27), (64, 64, 2, 64, 14, 27),
    (14, 14, 2, 32, 0, 31), (20, 12, 3, 64, 0, 39), (64, 64, 2, 64, 0, 127)])
@pytest.mark.parametrize("vf", [False, True])
def test_attention_planes(ops, H, W, heads, hd, ws, table, vf):
    import mmsa.backbone as bb
    B, D = 2, heads * hd
    att = R.Attention(D, heads, (table // 2 + 1, table // 2 + 1))
    sd = seeded_state_dict(att, 7)
    sd["rel_pos_h"] = torch.randn(table, hd, generator=g(50)) * 0.3
    sd["rel_pos_w"] = torch.randn(table, hd, generator=g(51)) * 0.3
    sd["qkv.bias"] = torch.randn(3 * D, generator=g(52)) * 0.5
    att.load_state_dict(sd)
    x = torch.randn(B, H, W, D, generator=g(53))
    with torch.no_grad():
        if ws:
            xw, pad_hw = R.window_partition(x, ws)
            ref = R.window_unpartition(att(xw), ws, pad_hw, (H, W))
        else:
            ref = att(x)
    T = H * W
    qkv = ops.alloc_planes(B * T, 3 * D, DEV, fmt=ops.FMT_F3, split=2 * D if vf else 0)     # the kernels' hi/lo form: fp16 pairs (f3 planes)
    ops.gemm(x.view(-1, D).to(DEV), ops.split_planes(sd["qkv.weight"].to(DEV)), bias=sd["qkv.bias"].to(DEV), out_planes=qkv)
    if vf:   # the GEMM's split output against a host-side split of the same matrix: q | k bf16 hi/lo, v fp16 hi + e5m2 lo
        qkv_f = (x.view(-1, D) @ sd["qkv.weight"].t() + sd["qkv.bias"]).to(DEV)
        assert_close(planes_to_float(qkv), qkv_f, tol=5e-5, what="qkv planes with the v columns as h8 planes")
    if ws:
        rh, rw = bb._rel_table(ws, sd["rel_pos_h"].to(DEV)), bb._rel_table(ws, sd["rel_pos_w"].to(DEV))
        kk = 2 * ws
    else:
        rh, rw = bb._rel_table(H, sd["rel_pos_h"].to(DEV)), bb._rel_table(W, sd["rel_pos_w"].to(DEV))
        kk = H + W
    rp = torch.empty(B * heads * T, kk, device=DEV)
    ops.relpos_bias(qkv, rh, rw, rp, B, H, W, heads, hd, ws)
    ao = ops.alloc_planes(B * T, D, DEV)
    bias_row = sd["qkv.bias"].reshape(1, -1).contiguous().to(DEV)
    biasp = ops.split_planes_qkv(bias_row, D) if vf else ops.split_planes(bias_row, kpad=3 * D, fmt=ops.FMT_F3)
    gw = torch.zeros(1, device=DEV)
    ops.attention(qkv, biasp, rp, ao, B, H, W, heads, hd, ws, hd ** -0.5, max_logit=gw)
    if not vf:   # bf16 hi/lo planes have the same layout and would be misread: refused
        with pytest.raises(RuntimeError):
            ops.attention(qkv, ops.split_planes(bias_row, kpad=3 * D), rp, ao, B, H, W, heads, hd, ws, hd ** -0.5)
    out = torch.empty(B * T, D, device=DEV)
    ops.gemm(ao, ops.split_planes(sd["proj.weight"].to(DEV)), out, bias=sd["proj.bias"].to(DEV))
    assert_close(out.view(B, H, W, D), ref, what=f"attention planes {H}x{W} ws={ws} vf={vf}")
    # logit guard word (include/mmsa.h): max |logit| over live queries and existing keys, natural units
    want = _ref_max_logit(att, xw, (ws, ws), _live_queries(B, H, W, ws, pad_hw)) if ws else _ref_max_logit(att, x, (H, W))
    assert abs(gw.item() - want) <= 2e-5 * max(want, 1.0), (gw.item(), want)


def test_msda_and_dwconv_planes_outputs(ops):
    B, M, D, L, Pn, Lq = 1, 4, 32, 1, 4, 100
    S = 10 * 10
    ss = torch.tensor([(10, 10)], dtype=torch.long).to(DEV)
    lsi = torch.zeros(1, dtype=torch.long, device=DEV)
    val = torch.randn(B * S, M * D, generator=g(60)).to(DEV)
    raw = torch.randn(B * Lq, M * L * Pn * 3, generator=g(61)).to(DEV)
    ref_pts = torch.rand(Lq, 2, generator=g(62)).to(DEV)
    o32 = torch.empty(B * Lq, M * D, device=DEV)
    for pf, tol in ((ops.FMT_B3, 5e-5), (ops.FMT_F3, 2e-6), (ops.FMT_H8, 1.5e-4)):   # (f3 since round 6: an interaction that followed its blocks onto fp16 pairs)
        op = ops.alloc_planes(B * Lq, M * D, DEV, fmt=pf)
        ops.msda_fused(val, ss, lsi, raw, ref_pts, o32, B, S, M, D, L, Lq, Pn, out_planes=op)
        assert_close(planes_to_float(op), o32.cpu(), tol=tol, what=f"msda planes (format {pf})")
    C, H, W = 32, 9, 7
    conv = torch.nn.Conv2d(C, C, 3, padding=1, groups=C)
    x = torch.randn(2, C, H, W, generator=g(63))
    ref = F.gelu(conv(x)).detach().permute(0, 2, 3, 1).reshape(-1, C)
    for pf, tol in ((ops.FMT_B3, 5e-5), (ops.FMT_F3, 2e-6)):
        p = ops.alloc_planes(2 * H * W, C, DEV, fmt=pf)
        ops.dwconv(x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV), conv.weight.detach().reshape(C, 9).t().contiguous().to(DEV),
                   conv.bias.detach().to(DEV), None, 2, H, W, 3, act="gelu", out_planes=p)
        assert_close(planes_to_float(p), ref, tol=tol, what=f"dwconv planes (format {pf})")


@pytest.mark.parametrize("M,D,L,shapes", [(4, 32, 1, [(10, 10)]), (2, 32, 3, [(12, 12), (6, 6), (3, 3)]), (3, 64, 2, [(9, 7), (5, 4)]), (4, 8, 1, [(6, 5)])])
def test_msda_gather_on_fp16_value_planes(ops, M, D, L, shapes):
    """mmsa_msda_fused_planes (round 6): the deformable-attention gather with `value` as H8 activation planes -- the 8 fp16 hi values of a lane's channels per
    corner (lo_bytes = False: the value rounded to 11 significant bits) or hi + e5m2 lo bytes (~14 bits) -- against mmsa_msda_fused on the fp32 values:
    with the SAME rounded values handed to the fp32 kernel the two kernels agree to accumulation order; offsets large enough that samples fall outside
    the map (zero taps), several levels, head widths 8 / 32 / 64."""
    B, Pn, Lq = 2, 4, 77
    S = sum(h * w for h, w in shapes)
    ss = torch.tensor(shapes, dtype=torch.long).to(DEV)
    lsi = torch.tensor([sum(h * w for h, w in shapes[:i]) for i in range(L)], dtype=torch.long).to(DEV)
    val = (torch.randn(B * S, M * D, generator=g(160)) * 3.0).to(DEV)
    raw = torch.randn(B * Lq, M * L * Pn * 3, generator=g(161))
    raw[:, :M * L * Pn * 2] *= 4.0                      # offsets of several pixels: a visible share of the samples leaves the map
    raw = raw.to(DEV)
    ref_pts = torch.rand(Lq, 2, generator=g(162)).to(DEV)
    vp = ops.split_planes(val, fmt=ops.FMT_H8)
    full = ops.planes_to_float(vp)[:, :M * D].contiguous()                                  # hi + lo / 2^11
    hi_only = val.clamp(-57344.0, 57344.0).half().float()
    o_ref = torch.empty(B * Lq, M * D, device=DEV)
    for lo_bytes, rounded in ((False, hi_only), (True, full)):
        ops.msda_fused(rounded.contiguous(), ss, lsi, raw, ref_pts, o_ref, B, S, M, D, L, Lq, Pn)
        o32 = torch.empty(B * Lq, M * D, device=DEV)
        op = ops.alloc_planes(B * Lq, M * D, DEV, fmt=ops.FMT_H8C if (M * D) % 64 == 0 else ops.FMT_B3)
        ops.msda_fused(vp, ss, lsi, raw, ref_pts, o32, B, S, M, D, L, Lq, Pn, out_planes=op, lo_bytes=lo_bytes)
        assert_close(o32, o_ref, tol=2e-6, what=f"msda on value planes (lo_bytes={lo_bytes}) vs the fp32 kernel on the same rounded values")
        assert_close(planes_to_float(op)[:, :M * D], o32.cpu(), tol=5e-5, what="msda on value planes: planes output")
    # what the format costs against the unrounded values: fp16 ~ 2^-12 per value, hi + lo ~ 2^-15
    ops.msda_fused(val, ss, lsi, raw, ref_pts, o_ref, B, S, M, D, L, Lq, Pn)
    for lo_bytes, tol in ((False, 6e-4), (True, 8e-5)):   # (2^-11 = 4.9e-4 is the worst case of one fp16 rounding; few samples average at these sizes)
        o32 = torch.empty(B * Lq, M * D, device=DEV)
        ops.msda_fused(vp, ss, lsi, raw, ref_pts, o32, B, S, M, D, L, Lq, Pn, lo_bytes=lo_bytes)
        assert_close(o32, o_ref, tol=tol, what=f"msda on value planes (lo_bytes={lo_bytes}) vs fp32 values")
    with pytest.raises(RuntimeError):
        ops.msda_fused(ops.split_planes(val, fmt=ops.FMT_B3), ss, lsi, raw, ref_pts, o32, B, S, M, D, L, Lq, Pn)


@pytest.mark.parametrize("H,W,heads,ws", [(16, 16, 2, 14), (20, 20, 2, 14), (64, 64, 2, 14), (30, 22, 3, 7), (14, 14, 1, 14),
                                          (9, 33, 2, 5)])
@pytest.mark.parametrize("vf", [False, True])
def test_window_attention_fused_relpos(ops, H, W, heads, ws, vf):
    """K/V-resident windowed kernel (rel-pos fused, no mmsa_relpos_bias pass) vs the oracle's Attention on partitioned windows.
    vf: qkv (GEMM output), bias row and rel-pos table as h8 planes and an fp16 selector -> every contraction on the fp16 MFMA (v_fmt = 2)."""
    hd, B = 64, 2
    D = heads * hd
    L = 2 * ws - 1
    att = R.Attention(D, heads, (ws, ws))
    sd = seeded_state_dict(att, 17)
    sd["rel_pos_h"] = torch.randn(L, hd, generator=g(70)) * 0.3
    sd["rel_pos_w"] = torch.randn(L, hd, generator=g(71)) * 0.3
    sd["qkv.bias"] = torch.randn(3 * D, generator=g(72)) * 0.5
    att.load_state_dict(sd)
    x = torch.randn(B, H, W, D, generator=g(73))
    with torch.no_grad():
        xw, pad_hw = R.window_partition(x, ws)
        ref = R.window_unpartition(att(xw), ws, pad_hw, (H, W))
    T = H * W
    pf = ops.FMT_H8 if vf else ops.FMT_F3      # hi/lo form: fp16 pairs (f3 planes) since round 4
    qkv = ops.alloc_planes(B * T, 3 * D, DEV, fmt=pf)
    ops.gemm(x.view(-1, D).to(DEV), ops.split_planes(sd["qkv.weight"].to(DEV)), bias=sd["qkv.bias"].to(DEV), out_planes=qkv)
    relp = ops.window_relpos_planes(sd["rel_pos_h"].to(DEV), sd["rel_pos_w"].to(DEV), ws, fmt=pf)
    bias_row = sd["qkv.bias"].reshape(1, -1).contiguous().to(DEV)
    biasp = ops.split_planes(bias_row, kpad=3 * D, fmt=pf)
    ao = ops.alloc_planes(B * T, D, DEV)
    ops.window_attention(qkv, biasp, relp, ao, B, H, W, heads, hd, ws, hd ** -0.5)
    out = torch.empty(B * T, D, device=DEV)
    ops.gemm(ao, ops.split_planes(sd["proj.weight"].to(DEV)), out, bias=sd["proj.bias"].to(DEV))
    assert_close(out.view(B, H, W, D), ref, what=f"window attention {H}x{W} ws={ws} vf={vf}")
    with pytest.raises(RuntimeError):
        ops.window_attention(qkv, biasp, relp, ao, B, H, W, heads, hd, 15, hd ** -0.5)
    # logit guard (include/mmsa.h): the launch folds max |logit| -- scale * q.k + rel-pos terms over live queries and existing keys -- into a device word
    gw = torch.zeros(1, device=DEV)
    ao_g = ops.alloc_planes(B * T, D, DEV)
    ops.window_attention(qkv, biasp, relp, ao_g, B, H, W, heads, hd, ws, hd ** -0.5, max_logit=gw)
    assert torch.equal(ao_g.p, ao.p), "the guard must not change the result"
    want = _ref_max_logit(att, xw, (ws, ws), _live_queries(B, H, W, ws, pad_hw))
    assert abs(gw.item() - want) <= (2e-3 if vf else 2e-5) * max(want, 1.0), (gw.item(), want)
    ops.window_attention(qkv, biasp, relp, ao_g, B, H, W, heads, hd, ws, hd ** -0.5, max_logit=gw)   # never lowered, and idempotent
    assert abs(gw.item() - want) <= (2e-3 if vf else 2e-5) * max(want, 1.0)
    if vf:   # qkv, bias and rel-pos planes must agree on the format; the split form belongs to the entry with a rel-pos prepass
        with pytest.raises(RuntimeError):
            ops.window_attention(qkv, ops.split_planes(bias_row, kpad=3 * D), relp, ao, B, H, W, heads, hd, ws, hd ** -0.5)
        with pytest.raises(RuntimeError):
            ops.window_attention(qkv, biasp, ops.window_relpos_planes(sd["rel_pos_h"].to(DEV), sd["rel_pos_w"].to(DEV), ws), ao, B, H, W, heads, hd, ws, hd ** -0.5)
        with pytest.raises(RuntimeError):
            ops.window_attention(ops.alloc_planes(B * T, 3 * D, DEV, split=2 * D), ops.split_planes_qkv(bias_row, D),
                                 ops.window_relpos_planes(sd["rel_pos_h"].to(DEV), sd["rel_pos_w"].to(DEV), ws), ao, B, H, W, heads, hd, ws, hd ** -0.5)


def test_layernorm_row_groups_and_wrap(ops):
    """Grouped LayerNorm of the batched TwinConvNeXt chain: per-group weights, column offset + row wrap (channel concat)."""
    P, C = 300, 96
    x = torch.randn(2 * P, C, generator=g(80)) * 2 + 0.5
    w, b = torch.randn(2, C, generator=g(81)), torch.randn(2, C, generator=g(82))
    ref = torch.cat([F.layer_norm(x[:P], (C,), w[0], b[0], 1e-6), F.layer_norm(x[P:], (C,), w[1], b[1], 1e-6)], 1)   # [P, 2C]
    out = torch.zeros(P, 2 * C, device=DEV)
    outp = ops.alloc_planes(P, 2 * C, DEV, zero=True)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, out, out_planes=outp, group_rows=P, w_gstride=C, y_gcol=C, y_wrap=True)
    assert_close(out, ref, tol=5e-5, what="grouped LN fp32 (wrap)")
    assert_close(planes_to_float(outp), ref, tol=5e-5, what="grouped LN planes (wrap)")
    stacked = torch.cat([F.layer_norm(x[:P], (C,), w[0], b[0], 1e-6), F.layer_norm(x[P:], (C,), w[1], b[1], 1e-6)], 0)
    out2 = torch.empty(2 * P, C, device=DEV)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, out2, group_rows=P, w_gstride=C)
    assert_close(out2, stacked, tol=5e-5, what="grouped LN stacked")


@pytest.mark.parametrize("B,H,W,C", [(1, 16, 48, 96), (2, 32, 32, 64), (1, 48, 16, 160)])
def test_dwconv7_block_kernel_equals_the_tiled_kernel_and_torch(ops, B, H, W, C):
    """The 16 x 16 x 32 block kernel (fp32 output only, H, W % 16 == 0, C % 32 == 0: the ConvNeXt stages of the shipped configs) against torch's grouped
    conv, and BITWISE against the tiled kernel -- the same call with a planes output beside the fp32 one runs that one (same terms, same order, same
    packed FMAs).  Two image groups with their own weights; maps with tiles on every border and in the interior."""
    convs = [torch.nn.Conv2d(C, C, 7, padding=3, groups=C) for _ in range(2)]
    x = torch.randn(2 * B, C, H, W, generator=g(84))
    ref = torch.cat([convs[0](x[:B]), convs[1](x[B:])], 0).detach().permute(0, 2, 3, 1).reshape(-1, C)
    wt = torch.stack([c.weight.detach().reshape(C, 49).t().contiguous() for c in convs], 0).contiguous().to(DEV)   # [2, 49, C]
    bs = torch.stack([c.bias.detach() for c in convs], 0).contiguous().to(DEV)
    xd = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV)
    out = torch.full((2 * B * H * W, C), float("nan"), device=DEV)
    ops.dwconv(xd, wt, bs, out, 2 * B, H, W, 7, imgs_per_group=B)
    assert_close(out, ref, tol=5e-5, what="dwconv7 block kernel vs torch")
    if C % 64 == 0 and H % 8 == 0 and W % 8 == 0:
        out2 = torch.full_like(out, float("nan"))
        pl = ops.alloc_planes(2 * B * H * W, C, DEV)
        ops.dwconv(xd, wt, bs, out2, 2 * B, H, W, 7, imgs_per_group=B, out_planes=pl)
        assert torch.equal(out, out2), "block kernel and tiled kernel must agree bitwise"


def test_dwconv7_image_groups(ops):
    B, H, W, C = 2, 12, 20, 64
    convs = [torch.nn.Conv2d(C, C, 7, padding=3, groups=C) for _ in range(2)]
    x = torch.randn(2 * B, C, H, W, generator=g(83))
    ref = torch.cat([convs[0](x[:B]), convs[1](x[B:])], 0).detach().permute(0, 2, 3, 1).reshape(-1, C)
    wt = torch.stack([c.weight.detach().reshape(C, 49).t().contiguous() for c in convs], 0).contiguous()   # [2, 49, C]
    bs = torch.stack([c.bias.detach() for c in convs], 0).contiguous()
    out = torch.empty(2 * B * H * W, C, device=DEV)
    ops.dwconv(x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV), wt.to(DEV), bs.to(DEV), out, 2 * B, H, W, 7, imgs_per_group=B)
    assert_close(out, ref, tol=5e-5, what="dwconv7 with per-group weights")


def test_gemm_batched_per_batch_bias_and_colscale(ops):
    M, N, K = 384, 160, 96
    a = torch.randn(2 * M, K, generator=g(84))
    w = torch.randn(2, N, K, generator=g(85)) / K ** 0.5
    bias, gam, res = torch.randn(2, N, generator=g(86)), torch.randn(2, N, generator=g(87)), torch.randn(2 * M, N, generator=g(88))
    ref = torch.cat([res[i * M:(i + 1) * M] + gam[i] * F.linear(a[i * M:(i + 1) * M].double(), w[i].double(), bias[i].double()).float()
                     for i in range(2)], 0)
    ap = ops.split_planes(a.to(DEV), kpad=K)
    wp = ops.split_planes(w.reshape(2 * N, K).to(DEV))
    wp0 = ops.Planes(wp.p[:N], N, K, wp.kpad)
    out = torch.empty(2 * M, N, device=DEV)
    ops.gemm(ap, wp0, out, bias=bias.to(DEV), colscale=gam.to(DEV), resid=res.to(DEV), batch=2, m=M, stride_a=M * 2 * ap.kpad,
             stride_w=N * 2 * wp.kpad, stride_bias=N, stride_r=M * N, stride_c=M * N)
    assert_close(out, ref, what="batched gemm, per-batch weights / bias / colscale")


def test_dwpair_gate_and_ca_apply_planes(ops):
    B, H, W, C = 2, 10, 14, 64
    conv = torch.nn.Conv2d(2 * C, 2 * C, 3, padding=1, groups=C, bias=False)
    x = torch.randn(B, 2 * C, H, W, generator=g(89))
    y = conv(x).detach()
    ref = (F.gelu(y[:, :C]) * y[:, C:]).permute(0, 2, 3, 1).reshape(-1, C)
    wt = conv.weight.detach().reshape(C, 2, 2, 9).permute(3, 0, 2, 1).contiguous()   # [tap][group][ci][co]
    out = torch.empty(B * H * W, C, device=DEV)
    outp = ops.alloc_planes(B * H * W, C, DEV)
    xin = x.permute(0, 2, 3, 1).reshape(-1, 2 * C).contiguous().to(DEV)
    ops.dwpair_gate(xin, wt.to(DEV), out, B, H, W, C, out_planes=outp)
    assert_close(out, ref, tol=5e-5, what="dwpair_gate fp32")
    assert_close(planes_to_float(outp), ref, tol=5e-5, what="dwpair_gate planes")
    z = torch.randn(B * H * W, C, generator=g(90))
    att = torch.rand(B * (H + W), C, generator=g(91))
    ah = att.view(B, H + W, C)[:, :H].reshape(B, H, 1, C)
    aw = att.view(B, H + W, C)[:, H:].reshape(B, 1, W, C)
    zr = z.view(B, H, W, C)
    refz = (zr + zr * aw * ah).reshape(-1, C)
    zp = ops.alloc_planes(B * H * W, C, DEV)
    ops.ca_apply(z.to(DEV), att.to(DEV), None, B, H, W, out_planes=zp)
    assert_close(planes_to_float(zp), refz, tol=5e-5, what="ca_apply planes")


@pytest.mark.parametrize("vf", [False, True])
@pytest.mark.parametrize("H", [64, 32])
def test_global_attention_fused_relpos(ops, H, vf):
    """Global flash kernel with the rel-pos terms computed in its prologue (no prepass) vs the oracle's Attention."""
    import mmsa.backbone as bb
    W, heads, hd, B = 64, 2, 64, 2
    D = heads * hd
    att = R.Attention(D, heads, (H, W))
    sd = seeded_state_dict(att, 27)
    sd["rel_pos_h"] = torch.randn(2 * H - 1, hd, generator=g(92)) * 0.3
    sd["rel_pos_w"] = torch.randn(2 * W - 1, hd, generator=g(93)) * 0.3
    sd["qkv.bias"] = torch.randn(3 * D, generator=g(94)) * 0.5
    att.load_state_dict(sd)
    x = torch.randn(B, H, W, D, generator=g(95))
    with torch.no_grad():
        ref = att(x)
    T = H * W
    pf = ops.FMT_H8 if vf else ops.FMT_F3     # vf: h8 planes throughout, every contraction of the kernel on the fp16 MFMA (v_fmt = 2)
    qkv = ops.alloc_planes(B * T, 3 * D, DEV, fmt=pf)
    ops.gemm(x.view(-1, D).to(DEV), ops.split_planes(sd["qkv.weight"].to(DEV)), bias=sd["qkv.bias"].to(DEV), out_planes=qkv)
    relg = ops.global_relpos_planes(sd["rel_pos_h"].to(DEV), sd["rel_pos_w"].to(DEV), fmt=pf)
    bias_row = sd["qkv.bias"].reshape(1, -1).contiguous().to(DEV)
    biasp = ops.split_planes(bias_row, kpad=3 * D, fmt=pf)
    ao = ops.alloc_planes(B * T, D, DEV)
    gw = torch.zeros(1, device=DEV)
    ops.global_attention(qkv, biasp, relg, ao, B, H, W, heads, hd, hd ** -0.5, max_logit=gw)
    want = _ref_max_logit(att, x, (H, W))
    assert abs(gw.item() - want) <= (2e-3 if vf else 2e-5) * max(want, 1.0), (gw.item(), want)
    out = torch.empty(B * T, D, device=DEV)
    ops.gemm(ao, ops.split_planes(sd["proj.weight"].to(DEV)), out, bias=sd["proj.bias"].to(DEV))
    assert_close(out.view(B, H, W, D), ref, what=f"global attention fused rel-pos {H}x{W} vf={vf}")


@pytest.mark.parametrize("fmt", ["b3", "h8"])
@pytest.mark.parametrize("M,N,K,outk", [
    (8192, 1024, 4096, "C"),      # lin2: one tile per workgroup, 128 k-tiles (124 of them straight-line steps)
    (4096, 4096, 1024, "P"),      # lin1 of one image: 4 tiles per workgroup on 128 of the CUs' worth of tiles, planes output
    (21504, 1024, 256, "C"),      # extractor ConvFFN fc2: 8 k-tiles per tile (4 straight-line), 5.25 tiles per workgroup, residual
    (3000, 640, 384, "P"),        # ragged bottom edge, 12 k-tiles
])
def test_gemm_straight_line_steps_race_screen(ops, M, N, K, outk, fmt):
    """Race screen of the GEMM main loop's straight-line k-tile steps (gemm_v2.hip K_STEP_PP_FAST: the same barriers and counted waits
    as the general step, both groups executing both waits) and of the h8 fp8-first MFMA order: 60 launches per shape, the odd ones
    while a second stream keeps the memory system and half of the CUs busy, every result bit-identical to the first and within the
    format's tolerance of fp64."""
    h8 = fmt == "h8"
    f = ops.FMT_H8 if h8 else ops.FMT_B3
    a = torch.randn(M, K, generator=g(130)) * 0.5
    w = torch.randn(N, K, generator=g(131)) / K ** 0.5
    ad, wd = a.to(DEV), w.to(DEV)
    bias = torch.randn(N, generator=g(132)).to(DEV)
    resid = torch.randn(M, N, generator=g(133)).to(DEV) if outk == "C" else None
    ref = ad.double() @ wd.double().t() + bias.double()
    ref = (ref + resid.double() if resid is not None else torch.nn.functional.gelu(ref)).float()
    ap = ops.split_planes(ad, kpad=K, fmt=f)
    wp = ops.split_planes(wd, fmt=f, weight=h8)
    side = torch.cuda.Stream()
    junk_a = torch.randn(8192, 2048, device=DEV)
    junk_w = ops.split_planes(torch.randn(2048, 2048, device=DEV))
    junk_o = torch.empty(8192, 2048, device=DEV)
    first = None
    for rep in range(60):
        if rep & 1:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    ops.gemm(junk_a, junk_w, junk_o)
                    junk_o.mul_(0.5)
        if outk == "C":
            out = torch.full((M, N), float("nan"), device=DEV)
            ops.gemm(ap, wp, out, bias=bias, resid=resid)
            got = out
        else:
            outp = ops.alloc_planes(M, N, DEV, fmt=f)
            ops.gemm(ap, wp, bias=bias, act="gelu", out_planes=outp)
            got = outp.p
        if first is None:
            first = got.clone()
            val = got if outk == "C" else planes_to_float(outp)
            assert_close(val, ref, tol=2e-4 if h8 else 3e-5, what=f"gemm {M}x{N}x{K} {fmt} {outk}")
        else:
            assert torch.equal(got, first), f"launch {rep} of {M}x{N}x{K} {fmt} {outk} differs from the first"
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,mode", [
    (70000, 200, 96, "C"),        # 548 tiles on 256 persistent workgroups (3 per workgroup), 3 k-tiles, ragged M and N
    (9000, 1100, 160, "P"),       # 324 tiles, 5 k-tiles, planes-only output, ragged right edge
    (33000, 130, 32, "C"),        # one k-tile per output tile: every k-tile is a tile boundary
    (600, 4100, 64, "CP"),        # 3 x 33 tiles, two k-tiles, both outputs
    (66000, 384, 1536, "Cres"),   # deep K, residual, 774 tiles of 128 columns -> 1032 tiles of 96 columns (fewer idle CUs in the last round)
    (16384, 384, 1536, "Cres"),   # the ConvNeXt stage-2 pw2 shape: 192 tiles of 128 columns -> 256 tiles of 96
    (43008, 192, 1024, "C"),      # MSDA offsets + weights projection: 336 tiles either way, 96-column tiles are 3/4 of the work
    (8192, 576, 1024, "C"),       # injector projection: 160 -> 192 tiles
    (5000, 300, 64, "C"),         # 96-column tiles with a ragged right edge (300 = 3 x 96 + 12) and a ragged bottom
])
def test_gemm_persistent_tile_stream(ops, M, N, K, mode):
    """The LDS-DMA kernel's continuous (tile, k-tile) stream: several output tiles per workgroup, few k-tiles per tile, ragged
    edges -- against fp64, and bit-identical when repeated (the ping-pong schedule has no timing-dependent result)."""
    a = torch.randn(M, K, generator=g(120)) * 0.5
    w = torch.randn(N, K, generator=g(121)) / K ** 0.5
    b = torch.randn(N, generator=g(122))
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    ref = (ad.double() @ wd.double().t() + bd.double())
    res = None
    if mode == "Cres":
        res = torch.randn(M, N, generator=g(123)).to(DEV)
        ref = ref + res.double()
    else:
        ref = torch.relu(ref)
    ref = ref.float()
    ap = ops.split_planes(ad, kpad=K)
    pl = ops.split_planes(wd)
    outs = []
    for rep in range(3):
        out = torch.full((M, N), float("nan"), device=DEV) if "C" in mode else None
        outp = ops.alloc_planes(M, N, DEV) if "P" in mode else None
        ops.gemm(ap, pl, out, bias=bd, act="none" if mode == "Cres" else "relu", resid=res, out_planes=outp)
        got = out if out is not None else planes_to_float(outp)
        outs.append(got.clone())
        if out is not None and outp is not None:
            assert_close(planes_to_float(outp), ref, tol=3e-5, what="planes output")
    assert_close(outs[0], ref, tol=3e-5, what=f"gemm {M}x{N}x{K} {mode}")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("M,N,K,mode", [
    (16384, 1536, 384, "Pgelu"),     # ConvNeXt stage-2 pw1 (the shape the 4-wave flavour was built for)
    (9000, 1100, 160, "P"),
    (70000, 200, 96, "C"),
    (5000, 300, 64, "Cres"),
    (700, 130, 1024, "Cres"),        # deep K through the 2-slot ring
    (33000, 130, 32, "C"),           # one k-tile per tile
])
def test_gemm_workgroup_flavours_agree_bitwise(ops, M, N, K, mode):
    """The two workgroup flavours of the LDS-DMA GEMM (8 waves: 256-row ping-pong tiles, one workgroup per CU; 4 waves: 128-row
    tiles, two workgroups per CU with a 2-slot ring) accumulate every output element over k in the same order: their results are
    bit-identical, so which one a shape is routed to can never change a value.  Both against fp64 too."""
    from mmsa import lib
    a = torch.randn(M, K, generator=g(140)) * 0.5
    w = torch.randn(N, K, generator=g(141)) / K ** 0.5
    b = torch.randn(N, generator=g(142))
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    ref = ad.double() @ wd.double().t() + bd.double()
    res = torch.randn(M, N, generator=g(143)).to(DEV) if mode == "Cres" else None
    act = "gelu" if mode == "Pgelu" else "none"
    if act == "gelu":
        ref = F.gelu(ref)
    if res is not None:
        ref = ref + res.double()
    ap, pl = ops.split_planes(ad, kpad=K), ops.split_planes(wd)
    got = {}
    try:
        for nw in (8, 4, 8, 4):
            ops.GEMM_FLAVOUR = nw
            out = torch.full((M, N), float("nan"), device=DEV) if mode[0] == "C" else None
            outp = ops.alloc_planes(M, N, DEV) if mode[0] == "P" else None
            ops.gemm(ap, pl, out, bias=bd, act=act, resid=res, out_planes=outp)
            o = out if out is not None else planes_to_float(outp)
            assert_close(o, ref.float(), tol=3e-5, what=f"flavour {nw}")
            if nw in got:
                assert torch.equal(got[nw], o)
            got[nw] = o.clone()
    finally:
        ops.GEMM_FLAVOUR = 0
    assert torch.equal(got[4], got[8]), "the 4-wave and the 8-wave flavour differ"


@pytest.mark.parametrize("M,N,K,batch,mode", [
    (8192, 1024, 1024, 1, "CPres_rs"),   # proj at two images: exactly one 256 x 128 tile per CU in the 8-wave form -> the site the 4-wave flavour is dispatched for
    (4096, 1024, 512, 1, "CPres_rs"),    # the K = 512 producer of the stream (one image)
    (8192, 512, 1024, 1, "C"),           # value projection on the ViT tokens
    (8192, 640, 1024, 1, "Crn"),         # offsets / attention-weights projection on the ViT tokens, LayerNorm-folded form
    (1000, 300, 128, 1, "Cres"),         # ragged in M and N: clamped operand rows, staged epilogue
    (130, 128, 64, 1, "C"),              # one k-tile pair per tile
    (2048, 256, 192, 2, "Pres"),         # two batches, planes-only output, three pairs
    (43008, 256, 1024, 1, "C"),          # many tiles per workgroup (forced flavour): the drained tile boundary, again and again
])
def test_gemm_h8c_four_wave_flavour_agrees_bitwise(ops, M, N, K, batch, mode):
    """gemm_h8c4_kernel (round 6: 128 x 128 tiles, 4 waves, two workgroups per CU -- dispatched for launches of at most one 256-row tile per CU, K <= 1024,
    plain epilogues) against gemm_h8c_kernel (8 waves, 256 x 128 tiles): the same MFMAs in the same order per accumulator and the same epilogue include, so every
    output -- fp32 rows, operand planes, strip sums -- is bit-identical whichever flavour a site is routed to.  Both against float64 on the kernel's operands."""
    a = torch.randn(batch * M, K, generator=g(610)) * 0.7
    w = torch.randn(batch * N, K, generator=g(611)) / K ** 0.5
    bias = torch.randn(batch * N, generator=g(612))
    ad, wd = a.to(DEV), w.to(DEV)
    ap = ops.split_planes(ad, kpad=K, fmt=ops.FMT_H8C) if batch == 1 else None
    if batch > 1:   # h8c batches hold an even number of rows: one planes tensor over all batches (M even)
        ap = ops.split_planes(ad, kpad=K, fmt=ops.FMT_H8C)
    wp_all = ops.split_planes(wd, fmt=ops.FMT_H8C)
    wp = ops.Planes(wp_all.p, N, K, wp_all.kpad, ops.FMT_H8C, False)
    af, wf = planes_to_float(ap)[:, :K].double().cpu(), planes_to_float(wp_all)[:, :K].double().cpu()
    acc = torch.cat([af[b * M:(b + 1) * M] @ wf[b * N:(b + 1) * N].t() for b in range(batch)], 0)
    bb = bias.double().view(batch, 1, N).expand(batch, M, N).reshape(batch * M, N)
    res = torch.randn(batch * M, N, generator=g(613)) if "res" in mode else None
    mr = csum = None
    if "rn" in mode:
        mr = torch.stack([torch.randn(M, generator=g(614)) * 0.2, 0.5 + torch.rand(M, generator=g(615))], 1).contiguous()
        csum = torch.randn(N, generator=g(616))
        ref = mr[:, 1:2].double() * (acc - mr[:, 0:1].double() * csum.double().view(1, N)) + bb
    else:
        ref = acc + bb
    if res is not None:
        ref = ref + res.double()
    ref = ref.float()
    got = {}
    try:
        for nw in (8, 4, 4):
            ops.GEMM_FLAVOUR = nw
            kw = dict(bias=bias.to(DEV), batch=batch, m=M, stride_a=ap.batch_stride(M) if batch > 1 else 0,
                      stride_w=wp_all.batch_stride(N) if batch > 1 else 0, stride_bias=N if batch > 1 else 0)
            out = outp = rs = None
            if "C" in mode:
                out = torch.full((batch * M, N), float("nan"), device=DEV)
                kw.update(out=out, stride_c=M * N if batch > 1 else 0)
            if "P" in mode:
                outp = ops.alloc_planes(batch * M, N, DEV, zero=True, fmt=ops.FMT_H8C)
                kw.update(out_planes=outp, stride_cp=outp.batch_stride(M) if batch > 1 else 0)
            if res is not None:
                kw.update(resid=res.to(DEV), stride_r=M * N if batch > 1 else 0)
            if "rs" in mode:
                rs = torch.full((batch * M, 2 * (N // 64)), float("nan"), device=DEV)
                kw.update(rowstats_out=rs)
            if mr is not None:
                kw.update(row_norm=(mr.to(DEV), csum.to(DEV)))
            ops.gemm(ap, wp, **kw)
            cur = [t.clone() if t is not None else None for t in (out, outp.p if outp is not None else None, rs)]
            if out is not None:
                assert_close(out, ref, tol=1e-4, what=f"h8c flavour {nw} {mode}: fp32 output vs float64 on the kernel's operands")
            if outp is not None:
                assert_close(planes_to_float(outp)[:, :N], ref, tol=1.5e-4, what=f"h8c flavour {nw} {mode}: planes output")
            if nw in got:
                assert all(x is None or torch.equal(x, y) for x, y in zip(cur, got[nw])), f"flavour {nw} is not reproducible"
            got[nw] = cur
    finally:
        ops.GEMM_FLAVOUR = 0
    for x, y, what in zip(got[4], got[8], ("fp32 rows", "planes", "strip sums")):
        assert x is None or torch.equal(x, y), f"the 4-wave and the 8-wave h8c flavour differ in their {what}"


@pytest.mark.parametrize("M,N,K,fmt_name,mode", [
    (131072, 96, 192, "b3", "Pres_alpha"),     # MobileNetV2 2c -> c at the finest neck level: planes-only output into a column slice, residual, scalar scale
    (131072, 192, 96, "b3", "Crelu6_slice"),   # MobileNetV2 c -> 2c: A is a column slice of wider planes, ReLU6, fp32 output
    (65536, 96, 192, "f3", "CPres_bias"),      # fp16 pairs, both outputs, bias + per-column scale
    (40000, 192, 96, "b3", "Crelu_bias"),      # 2500 blocks over 2048 waves: waves with one and with two blocks; ReLU
    (16391, 96, 192, "f3", "Cres"),            # M % 16 != 0: clamped loads, masked stores
    (131072, 192, 192, "b3", "Cres"),          # a shape the streaming kernel does NOT take (it ties there): both runs are the tiled kernel
])
def test_gemm_stream_kernel(ops, M, N, K, fmt_name, mode):
    """gemm_stream_kernel (round 6; the skinny launches of the neck: the whole weight matrix resident in LDS, every wave streams 16-row blocks of A through
    registers) against the tiled LDS-DMA kernel on the same call (`GEMM_FLAVOUR = 8` keeps a launch on the tiled kernel): the same MFMAs in the same order and the
    same epilogue operations, so fp32 rows and operand planes are bit-identical; both against float64 on the kernel's operands."""
    fmt = {"b3": ops.FMT_B3, "f3": ops.FMT_F3}[fmt_name]
    slice_a = "slice" in mode
    a_full = torch.randn(M, 2 * K if slice_a else K, generator=g(710)) * 0.8
    w = torch.randn(N, K, generator=g(711)) / K ** 0.5
    ap_full = ops.split_planes(a_full.to(DEV), fmt=fmt)
    ap = ap_full.cols(K, 2 * K) if slice_a else ap_full
    wp = ops.split_planes(w.to(DEV), fmt=fmt)
    af = planes_to_float(ap_full).double().cpu()
    af = af[:, K:2 * K] if slice_a else af[:, :K]
    acc = af @ planes_to_float(wp)[:, :K].double().cpu().t()
    bias = torch.randn(N, generator=g(712)).to(DEV) if "bias" in mode else None
    cs = (0.5 + torch.rand(N, generator=g(713))).to(DEV) if "bias" in mode else None
    alpha = 0.37 if "alpha" in mode else 1.0
    res = torch.randn(M, N, generator=g(714)).to(DEV) if "res" in mode else None
    act = "relu6" if "relu6" in mode else "relu" if "relu" in mode else "none"
    ref = acc + (bias.double().cpu() if bias is not None else 0.0)
    if act == "relu6":
        ref = ref.clamp(0.0, 6.0)
    elif act == "relu":
        ref = ref.clamp(min=0.0)
    ref = ref * ((cs.double().cpu() if cs is not None else 1.0) * alpha)
    if res is not None:
        ref = ref + res.double().cpu()
    ref = ref.float()
    got = {}
    try:
        for nw in (0, 8, 0):          # 0: by shape (the streaming kernel takes these launches), 8: forced onto the tiled kernel
            ops.GEMM_FLAVOUR = nw
            out = torch.full((M, N), float("nan"), device=DEV) if "C" in mode else None
            wide = ops.alloc_planes(M, 2 * N, DEV, zero=True, fmt=fmt) if "P" in mode else None      # planes output into a column slice of a wider matrix
            outp = wide.cols(N, 2 * N) if wide is not None else None
            ops.gemm(ap, wp, out, bias=bias, act=act, alpha=alpha, colscale=cs, resid=res, out_planes=outp)
            cur = [out.clone() if out is not None else None, wide.p.clone() if wide is not None else None]
            if out is not None:
                assert_close(out, ref, tol=3e-5 if fmt_name == "b3" else 3e-6, what=f"stream gemm flavour {nw} {mode}: fp32 output")
            if wide is not None:
                assert_close(planes_to_float(wide)[:, N:2 * N], ref, tol=4e-5 if fmt_name == "b3" else 4e-6, what=f"stream gemm flavour {nw} {mode}: planes output")
                assert bool((wide.p[:, :2 * N] == 0).all())                                          # the neighbouring columns of the wider planes are untouched
            if nw in got:
                assert all(x is None or torch.equal(x, y) for x, y in zip(cur, got[nw])), f"flavour {nw} is not reproducible"
            got[nw] = cur
    finally:
        ops.GEMM_FLAVOUR = 0
    for x, y, what in zip(got[0], got[8], ("fp32 rows", "planes")):
        assert x is None or torch.equal(x, y), f"the streaming kernel and the tiled kernel differ in their {what}"


# ---------------------------------------------------------------------------------------------------------------------------
# "h8" operand planes (fp16 hi + e5m2 cross-term bytes; csrc/common.h): format, the GEMM on them, and the producers that emit them
def _h8_emulated_product(a, w):
    """What the kernel computes, restated with torch casts: hi.hi exactly + the two cross terms with e5m2-rounded operands."""
    ah = a.clamp(-57344, 57344).half().float()
    wh = w.clamp(-57344, 57344).half().float()
    al, wl = (a - ah) * 2048, (w - wh) * 2048
    q = lambda t: t.to(torch.float8_e5m2).float()
    return ah.double() @ wh.double().t() + (q(ah).double() @ q(wl).double().t() + q(al).double() @ q(wh).double().t()) / 2048


def test_split_planes_h8_roundtrip_and_layout(ops):
    x = torch.randn(37, 50, generator=g(41)) * 3
    for weight in (False, True):
        p = ops.split_planes(x.to(DEV), fmt=ops.FMT_H8, weight=weight)
        assert p.p.shape == (37, 128) and p.kpad == 64 and p.k == 50 and p.fmt == ops.FMT_H8 and p.weight == weight
        back = ops.planes_to_float(p, cols=64).cpu()
        assert torch.all(back[:, 50:] == 0)
        assert ((back[:, :50] - x).abs() <= x.abs() * 2 ** -13).all()      # 11 bits of hi + 3 of lo
        raw = p.p.cpu().view(torch.uint8).view(37, 2, 128)
        hi = raw[:, :, :64].contiguous().view(torch.float16)                # [37, 2, 32]
        assert torch.equal(hi[:, 0, :].float(), x[:, :32].half().float()) and torch.equal(hi[:, 1, :18].float(), x[:, 32:].half().float())
        # chunk g of a block: 8 lo bytes + 8 q(hi) bytes (activation) / swapped (weight), k = 8g .. 8g+7
        ch = raw[:, 0, 64:].reshape(37, 4, 2, 8)
        lo = ch[:, :, 1 if weight else 0].contiguous().view(torch.float8_e5m2).float().reshape(37, 32)
        qh = ch[:, :, 0 if weight else 1].contiguous().view(torch.float8_e5m2).float().reshape(37, 32)
        xh = x[:, :32].half().float()
        assert torch.equal(qh, xh.to(torch.float8_e5m2).float())
        assert torch.equal(lo, ((x[:, :32] - xh) * 2048).to(torch.float8_e5m2).float())
    # saturation instead of infinities
    big = torch.tensor([[1e6, -1e6, 60000.0, 3.0] * 8], device=DEV)
    assert torch.equal(ops.planes_to_float(ops.split_planes(big, fmt=ops.FMT_H8)).cpu()[0, :4], torch.tensor([57344.0, -57344.0, 57344.0, 3.0]))


@pytest.mark.parametrize("M,N,K,act", [(300, 96, 64, "none"), (1000, 576, 1024, "gelu"), (4096, 1024, 768, "none"), (77, 28, 128, "relu"),
                                       (8192, 1024, 4096, "none")])
def test_gemm_h8_operands(ops, M, N, K, act):
    a = torch.randn(M, K, generator=g(42)) * 1.7
    w = torch.randn(N, K, generator=g(43)) / K ** 0.5
    b = torch.randn(N, generator=g(44))
    res = torch.randn(M, N, generator=g(45))
    fact = {"none": lambda t: t, "gelu": F.gelu, "relu": F.relu}[act]
    ref = fact(F.linear(a.double(), w.double(), b.double())).float() + res
    ap = ops.split_planes(a.to(DEV), kpad=K, fmt=ops.FMT_H8)
    wp = ops.split_planes(w.to(DEV), fmt=ops.FMT_H8, weight=True)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(ap, wp, out, bias=b.to(DEV), act=act, resid=res.to(DEV))
    assert_close(out, ref, tol=5e-5, what="h8 gemm vs fp64")
    # the kernel's own arithmetic (pins the pairing of the fp8 operand bytes: a wrong k-slot pairing is off by ~2e-4)
    emu = fact(_h8_emulated_product(a, w) + b.double()).float() + res
    assert_close(out, emu, tol=3e-6, what="h8 gemm vs its emulation")
    # planes output in both formats from h8 operands, and h8 planes from bf16 hi/lo operands
    for ofmt in (ops.FMT_B3, ops.FMT_H8):
        outp = ops.alloc_planes(M, N, DEV, fmt=ofmt)
        ops.gemm(ap, wp, bias=b.to(DEV), act=act, resid=res.to(DEV), out_planes=outp)
        assert_close(planes_to_float(outp), ref, tol=1.5e-4 if ofmt == ops.FMT_H8 else 5e-5, what=f"h8 gemm planes out fmt {ofmt}")
    outp = ops.alloc_planes(M, N, DEV, fmt=ops.FMT_H8)
    ops.gemm(ops.split_planes(a.to(DEV), kpad=K), ops.split_planes(w.to(DEV)), bias=b.to(DEV), act=act, resid=res.to(DEV), out_planes=outp)
    assert_close(planes_to_float(outp), ref, tol=1.5e-4, what="b3 gemm, h8 planes out")


def test_gemm_h8_exact_on_integers_and_cross_terms(ops):
    """Small integers are exact in fp16 (lo = 0): the result must be the exact integer product.  Then operands whose lo parts
    carry all the signal: A = 1 + 2^-12 u (u in {-1, 0, 1}) needs the cross terms to come out right."""
    M, N, K = 256, 128, 128
    a = torch.randint(-8, 9, (M, K), generator=g(46)).float()
    w = torch.randint(-8, 9, (N, K), generator=g(47)).float()
    w[:, 0] += 100 * torch.arange(N)              # asymmetric: catches row/column swaps
    out = torch.empty(M, N, device=DEV)
    ops.gemm(ops.split_planes(a.to(DEV), fmt=ops.FMT_H8), ops.split_planes(w.to(DEV), fmt=ops.FMT_H8, weight=True), out)
    assert torch.equal(out.cpu(), a @ w.t())
    u = torch.randint(-1, 2, (M, K), generator=g(48)).float()
    v = torch.randint(-1, 2, (N, K), generator=g(49)).float()
    a2, w2 = 1 + u * 2 ** -12, 1 + v * 2 ** -12      # hi = 1, lo = +-2^-12: exactly representable in e5m2 after the 2^11 scaling
    ops.gemm(ops.split_planes(a2.to(DEV), fmt=ops.FMT_H8), ops.split_planes(w2.to(DEV), fmt=ops.FMT_H8, weight=True), out)
    exact = (K + (u.sum(1)[:, None] + v.sum(1)[None, :]) * 2 ** -12).double()      # hi.hi + both cross terms (lo.lo is dropped)
    assert (out.cpu().double() - exact).abs().max() <= 1e-6


# ---------------------------------------------------------------------------------------------------------------------------
# "h8c" operand planes (round 4; csrc/common.h, csrc/gemm_h8c.hip): the h8 arithmetic on 3 bytes per element, rows stored in pairs, q(hi) taken
# in registers as the fp16 value's top byte (truncation)
H8C_LO_COMP = 1.09375


def _h8c_emulated_product(a, w):
    """hi.hi exactly + the two cross terms with q(hi) = the fp16 hi value truncated to its top byte (an e5m2) and lo rounded to e5m2 after the
    scaling by 2^11 x 1.09375 that makes up for the truncation's mean (csrc/common.h MMSA_H8C_LO_COMP; the MFMA's block scale undoes the 2^11)."""
    ah = a.clamp(-57344, 57344).half()
    wh = w.clamp(-57344, 57344).half()
    al, wl = (a - ah.float()) * (2048 * H8C_LO_COMP), (w - wh.float()) * (2048 * H8C_LO_COMP)
    q = lambda t: t.to(torch.float8_e5m2).float()
    trunc = lambda h: (h.view(torch.int16) & -256).view(torch.float16).float()
    return ah.double() @ wh.double().t() + (trunc(ah).double() @ q(wl).double().t() + q(al).double() @ trunc(wh).double().t()) / 2048


def test_split_planes_h8c_roundtrip_and_layout(ops):
    x = torch.randn(37, 150, generator=g(241)) * 3
    p = ops.split_planes(x.to(DEV), fmt=ops.FMT_H8C)
    assert p.p.shape == (19, 3 * 192) and p.kpad == 192 and p.k == 150 and p.n == 37 and p.fmt == ops.FMT_H8C
    back = ops.planes_to_float(p, cols=192).cpu()
    assert back.shape == (37, 192) and torch.all(back[:, 150:] == 0)
    assert ((back[:, :150] - x).abs() <= x.abs() * 2 ** -13).all()      # 11 bits of hi + 3 of lo
    raw = p.p.cpu().view(torch.uint8).view(19, 6 * 192)
    hi = raw[:, :4 * 192].contiguous().view(torch.float16).view(38, 192)   # a pair's two rows of fp16 hi values
    assert torch.equal(hi[:37, :150].float(), x.half().float())
    # lo lines: chunk c of pair j = [row 2j: 64 B | row 2j+1: 64 B]; a row's 64 B = 4 groups g of [k = 64c + 8g .. +7 | k = 64c + 32 + 8g .. +7]
    lo = raw[:, 4 * 192:].reshape(19, 3, 2, 4, 2, 8)
    want = ((x - x.half().float()) * (2048 * H8C_LO_COMP)).to(torch.float8_e5m2).float()
    r, k = 36, 64 + 32 + 8 * 2 + 5                                         # row 36 (pair 18, first row), chunk 1, second k-tile, group 2, byte 5
    assert lo[18, 1, 0, 2, 1, 5].view(torch.float8_e5m2).float() == want[r, k]
    r, k = 7, 3                                                             # row 7 (pair 3, second row), chunk 0, first k-tile, group 0, byte 3
    assert lo[3, 0, 1, 0, 0, 3].view(torch.float8_e5m2).float() == want[r, k]
    # row slices start at even rows and are views of the same pairs
    sub = p.rows(4, 10)
    assert torch.equal(ops.planes_to_float(sub).cpu(), back[4:10, :150])
    with pytest.raises(RuntimeError):
        p.rows(3, 9)


@pytest.mark.parametrize("M,N,K,act", [(256, 128, 64, "none"), (300, 192, 128, "none"), (1000, 576, 1024, "gelu"), (4096, 1024, 768, "none"), (135, 130, 192, "relu"),
                                       (8192, 1024, 4096, "none"), (9800, 3072, 1024, "none")])
def test_gemm_h8c_operands(ops, M, N, K, act):
    a = torch.randn(M, K, generator=g(242)) * 1.7
    w = torch.randn(N, K, generator=g(243)) / K ** 0.5
    b = torch.randn(N, generator=g(244))
    res = torch.randn(M, N, generator=g(245))
    fact = {"none": lambda t: t, "gelu": F.gelu, "relu": F.relu}[act]
    ref = fact(F.linear(a.double(), w.double(), b.double())).float() + res
    ap = ops.split_planes(a.to(DEV), fmt=ops.FMT_H8C)
    wp = ops.split_planes(w.to(DEV), fmt=ops.FMT_H8C)
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(ap, wp, out, bias=b.to(DEV), act=act, resid=res.to(DEV))
    assert_close(out, ref, tol=8e-5, what="h8c gemm vs fp64")
    emu = fact(_h8c_emulated_product(a, w) + b.double()).float() + res
    assert_close(out, emu, tol=3e-6, what="h8c gemm vs its emulation")
    again = torch.empty(M, N, device=DEV)
    ops.gemm(ap, wp, again, bias=b.to(DEV), act=act, resid=res.to(DEV))
    assert torch.equal(again, out)
    # planes outputs: every format out of h8c operands, and h8c planes out of bf16 hi/lo and h8 line operands
    for ofmt in (ops.FMT_B3, ops.FMT_H8, ops.FMT_H8C):
        outp = ops.alloc_planes(M, N, DEV, zero=True, fmt=ofmt)
        ops.gemm(ap, wp, bias=b.to(DEV), act=act, resid=res.to(DEV), out_planes=outp)
        assert_close(planes_to_float(outp), ref, tol=5e-5 if ofmt == ops.FMT_B3 else 1.5e-4, what=f"h8c gemm planes out fmt {ofmt}")
        if ofmt == ops.FMT_H8C:    # bit-identical to splitting the fp32 output of the same GEMM
            assert torch.equal(planes_to_float(outp), planes_to_float(ops.split_planes(out, fmt=ops.FMT_H8C)))
            both = torch.empty(M, N, device=DEV)
            outp2 = ops.alloc_planes(M, N, DEV, zero=True, fmt=ofmt)
            ops.gemm(ap, wp, both, bias=b.to(DEV), act=act, resid=res.to(DEV), out_planes=outp2)
            assert torch.equal(both, out) and torch.equal(outp2.p, outp.p)
    if M >= 128:
        for (af, wf) in ((ops.FMT_B3, ops.FMT_B3), (ops.FMT_H8, ops.FMT_H8)):
            outp = ops.alloc_planes(M, N, DEV, zero=True, fmt=ops.FMT_H8C)
            ops.gemm(ops.split_planes(a.to(DEV), kpad=K, fmt=af), ops.split_planes(w.to(DEV), fmt=wf, weight=wf == ops.FMT_H8), bias=b.to(DEV), act=act,
                     resid=res.to(DEV), out_planes=outp)
            assert_close(planes_to_float(outp), ref, tol=1.5e-4, what=f"fmt {af} gemm, h8c planes out")


def test_gemm_h8c_exact_on_integers_and_cross_terms(ops):
    M, N, K = 512, 256, 256
    a = torch.randint(-8, 9, (M, K), generator=g(246)).float()
    w = torch.randint(-8, 9, (N, K), generator=g(247)).float()
    w[:, 0] = 16.0 * torch.arange(N)              # asymmetric: catches row/column swaps (multiples of 16 up to 4080: exact in fp16, lo = 0)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(ops.split_planes(a.to(DEV), fmt=ops.FMT_H8C), ops.split_planes(w.to(DEV), fmt=ops.FMT_H8C), out)
    assert torch.equal(out.cpu(), a @ w.t())
    u = torch.randint(-1, 2, (M, K), generator=g(248)).float()
    v = torch.randint(-1, 2, (N, K), generator=g(249)).float()
    a2, w2 = 1 + u * 2 ** -12, 1 + v * 2 ** -12      # hi = 1 (its top byte too), lo = +-2^-12: +-0.547 after the 2^11 x 1.09375 scaling, which e5m2 rounds to +-0.5
    ops.gemm(ops.split_planes(a2.to(DEV), fmt=ops.FMT_H8C), ops.split_planes(w2.to(DEV), fmt=ops.FMT_H8C), out)
    exact = (K + (u.sum(1)[:, None] + v.sum(1)[None, :]) * 2 ** -12).double()
    assert (out.cpu().double() - exact).abs().max() <= 1e-6


def test_gemm_h8c_batched_layernorm_fold_and_argument_checks(ops):
    """Strided-batched h8c operands (the up-conv's per-image batches), the LayerNorm-fold extras on the h8c kernel, and what it refuses."""
    B, M, N, K = 2, 512, 256, 128
    a = torch.randn(B * M, K, generator=g(250))
    w = torch.randn(N, K, generator=g(251)) / K ** 0.5
    ap, wp = ops.split_planes(a.to(DEV), fmt=ops.FMT_H8C), ops.split_planes(w.to(DEV), fmt=ops.FMT_H8C)
    out = torch.empty(B * M, N, device=DEV)
    ops.gemm(ap, wp, out, batch=B, m=M, stride_a=ap.batch_stride(M), stride_c=M * N)
    assert_close(out, (a.double() @ w.double().t()).float(), tol=8e-5, what="batched h8c gemm")
    # producer: fp32 + h8c planes + per-row strip sums; consumer: row-normalising epilogue on the raw planes (D = N = 256)
    xp, rs = ops.alloc_planes(B * M, N, DEV, fmt=ops.FMT_H8C), torch.empty(B * M, 2 * (N // 64), device=DEV)
    ops.gemm(ap, wp, out, out_planes=xp, rowstats_out=rs)
    x = out.cpu()
    assert_close(rs.view(B * M, N // 64, 2)[:, :, 0].sum(1).cpu(), x.sum(1), tol=1e-5, what="strip sums")
    mr = torch.empty(B * M, 2, device=DEV)
    ops.rowstats_finalize(rs, B * M, N, 1e-6, mr)
    w2 = torch.randn(128, N, generator=g(252)) / N ** 0.5
    lnw, lnb, b2 = torch.randn(N, generator=g(253)) * 0.3 + 1, torch.randn(N, generator=g(254)) * 0.1, torch.randn(128, generator=g(255))
    wf = ops.split_planes((w2 * lnw[None]).to(DEV), fmt=ops.FMT_H8C)
    cs = ops.planes_to_float(wf).double().sum(1).float().contiguous()
    bf = (w2.double() @ lnb.double()).float().add_(b2).to(DEV)
    hp = ops.alloc_planes(B * M, 128, DEV, fmt=ops.FMT_H8C)
    ops.gemm(xp, wf, bias=bf, act="gelu", out_planes=hp, row_norm=(mr, cs))
    ref = F.gelu(F.linear(F.layer_norm(x.double(), (N,), lnw.double(), lnb.double(), 1e-6), w2.double(), b2.double())).float()
    assert_close(planes_to_float(hp), ref, tol=3e-4, what="h8c consumer GEMM with the LayerNorm folded")
    with pytest.raises(RuntimeError):     # K % 64 != 0
        ops.gemm(ops.split_planes(torch.randn(128, 96, device=DEV), kpad=96), ops.split_planes(torch.randn(64, 96, device=DEV), fmt=ops.FMT_H8C), out)
    with pytest.raises(RuntimeError):     # operand formats differ
        ops.gemm(ops.split_planes(torch.randn(128, 128, device=DEV), fmt=ops.FMT_H8), wp, out)
    small = torch.randn(50, K, generator=g(259))     # fewer rows than a tile: rows beyond M are clamped on the way in and masked on the way out
    o50 = torch.empty(50, N, device=DEV)
    ops.gemm(ops.split_planes(small.to(DEV), fmt=ops.FMT_H8C), wp, o50)
    assert_close(o50, (small.double() @ w.double().t()).float(), tol=8e-5, what="h8c gemm, 50 rows")


@pytest.mark.parametrize("rows,C", [(96, 64), (300, 1024), (50, 100), (33, 1280)])
def test_layernorm_h8c_planes(ops, rows, C):
    x = torch.randn(rows, C, generator=g(256)) * 2 + 0.3
    w, b = torch.randn(C, generator=g(257)), torch.randn(C, generator=g(258))
    y = F.layer_norm(x, (C,), w, b, 1e-6)
    p = ops.alloc_planes(rows, C, DEV, zero=True, fmt=ops.FMT_H8C)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, out_planes=p)
    got = planes_to_float(p).cpu()
    assert ((got - y).abs() <= y.abs() * 2 ** -12 + 1e-5).all()
    yk = torch.empty(rows, C, device=DEV)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, out=yk)
    assert torch.equal(planes_to_float(p), planes_to_float(ops.split_planes(yk, fmt=ops.FMT_H8C)))


def test_gemm_h8_argument_checks(ops):
    a = ops.split_planes(torch.randn(128, 96, device=DEV), fmt=ops.FMT_H8)
    w = ops.split_planes(torch.randn(64, 96, device=DEV), fmt=ops.FMT_H8, weight=True)
    out = torch.empty(128, 64, device=DEV)
    with pytest.raises(RuntimeError):     # K % 64 != 0
        ops.gemm(a, w, out)
    with pytest.raises(RuntimeError):     # operand formats differ
        ops.gemm(ops.split_planes(torch.randn(128, 128, device=DEV)), ops.split_planes(torch.randn(64, 128, device=DEV), fmt=ops.FMT_H8, weight=True), out)
    with pytest.raises(RuntimeError):     # activation-ordered planes as W
        ops.gemm(ops.split_planes(torch.randn(128, 128, device=DEV), fmt=ops.FMT_H8), ops.split_planes(torch.randn(64, 128, device=DEV), fmt=ops.FMT_H8), out)


@pytest.mark.parametrize("rows,C", [(96, 64), (300, 1024), (64, 96), (50, 100), (33, 1280)])
def test_layernorm_h8_planes(ops, rows, C):
    x = torch.randn(rows, C, generator=g(50)) * 2 + 0.3
    w, b = torch.randn(C, generator=g(51)), torch.randn(C, generator=g(52))
    y = F.layer_norm(x, (C,), w, b, 1e-6)
    p = ops.alloc_planes(rows, C, DEV, zero=True, fmt=ops.FMT_H8)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, out_planes=p)
    got = planes_to_float(p).cpu()
    assert ((got - y).abs() <= y.abs() * 2 ** -12 + 1e-5).all()
    # bit-identical to splitting the fp32 LayerNorm output of the same kernel
    yk = torch.empty(rows, C, device=DEV)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, out=yk)
    p2 = ops.split_planes(yk, fmt=ops.FMT_H8)
    assert torch.equal(p.p[:, :2 * ops.pad32(C)].cpu().view(torch.uint8).view(rows, -1, 128)[:, :C // 32], p2.p.cpu().view(torch.uint8).view(rows, -1, 128)[:, :C // 32])


def test_convnext_block_layernorm_fold(ops):
    """The ConvNeXt block's front half with its LayerNorm folded (TC:102-111): the 7 x 7 depthwise conv writes the RAW output as planes plus
    per pixel and 64-channel chunk (sum, sum of squares); mmsa_rowstats_finalize -> (mean, rstd); pointwise_conv1 on W o w normalises in
    its epilogue.  Two image groups with their own weights (the two TwinConvNeXt streams), batched GEMM.  Against float64."""
    H, W, C = 16, 24, 128      # one image per group
    N = 4 * C
    x = torch.randn(2, H * W, C, generator=g(301)) * 1.3 + 0.2
    dw = torch.randn(2, C, 7, 7, generator=g(302)) * 0.1
    dwb = torch.randn(2, C, generator=g(303)) * 0.1
    lnw, lnb = 1.0 + 0.1 * torch.randn(2, C, generator=g(304)), 0.05 * torch.randn(2, C, generator=g(305))
    w1 = torch.randn(2, N, C, generator=g(306)) / C ** 0.5
    b1 = torch.randn(2, N, generator=g(307)) * 0.1
    d64, ref = [], []
    for s_ in range(2):
        xi = x[s_].double().t().reshape(1, C, H, W)
        d = F.conv2d(xi, dw[s_].double().unsqueeze(1), dwb[s_].double(), padding=3, groups=C)[0].reshape(C, H * W).t()
        d64.append(d)
        ref.append(F.gelu(F.layer_norm(d, (C,), lnw[s_].double(), lnb[s_].double(), 1e-6) @ w1[s_].double().t() + b1[s_].double()))
    d64, ref = torch.cat(d64), torch.cat(ref)
    P = H * W
    xd = x.reshape(2 * P, C).contiguous().to(DEV)
    wt = dw.reshape(2, C, 49).permute(0, 2, 1).contiguous().to(DEV)       # [group][49][C] tap-major
    n = ops.alloc_planes(2 * P, C, DEV)
    rs = torch.full((2 * P, 2 * (C // 64)), float("nan"), device=DEV)
    ops.dwconv(xd, wt, dwb.to(DEV).contiguous(), None, 2, H, W, 7, imgs_per_group=1, out_planes=n, rowstats_out=rs)
    assert_close(planes_to_float(n), d64.float(), tol=2e-5, what="depthwise conv planes")
    want = torch.stack([d64.view(2 * P, C // 64, 64).sum(-1), (d64 ** 2).view(2 * P, C // 64, 64).sum(-1)], -1).view(2 * P, -1)
    assert_close(rs, want.float(), tol=2e-5, what="strip sums of the conv output")
    mr = torch.empty(2 * P, 2, device=DEV)
    ops.rowstats_finalize(rs, 2 * P, C, 1e-6, mr)
    wps = [ops.split_planes((w1[s_] * lnw[s_][None, :]).to(DEV).contiguous()) for s_ in range(2)]
    buf = torch.cat([wps[0].p, wps[1].p], 0).contiguous()
    wp = ops.Planes(buf[:N], N, C, wps[0].kpad, wps[0].fmt, wps[0].weight)
    wp.full = buf
    cs = torch.cat([planes_to_float(wps[s_])[:N, :C].double().sum(1).float() for s_ in range(2)]).contiguous()
    bf = torch.cat([(w1[s_].double() @ lnb[s_].double()).float().add(b1[s_]) for s_ in range(2)]).to(DEV).contiguous()
    out = ops.alloc_planes(2 * P, N, DEV)
    ops.gemm(n, wp, bias=bf, act="gelu", out_planes=out, row_norm=(mr, cs), batch=2, m=P, stride_a=P * 2 * n.kpad,
             stride_w=N * 2 * wp.kpad, stride_bias=N, stride_cp=P * 2 * out.kpad)
    assert_close(planes_to_float(out), ref.float(), tol=4e-5, what="dwconv -> folded LayerNorm -> pointwise_conv1 -> GELU")
    # without the strip sums the conv is the plain one; and the strip sums need the tiled kernel's shapes
    with pytest.raises(RuntimeError):
        ops.dwconv(xd[:, :96].contiguous(), wt[:, :, :96].contiguous(), dwb[:, :96].to(DEV).contiguous(), None, 2, H, W, 7, imgs_per_group=1,
                   out_planes=ops.alloc_planes(2 * P, 96, DEV), rowstats_out=rs)


@pytest.mark.parametrize("fmt_name,act", [("b3", "none"), ("h8", "gelu")])
def test_gemm_layernorm_fold(ops, fmt_name, act):
    """LayerNorm folded into a producer / consumer pair of GEMMs (IE:396-421; include/mmsa.h mmsa_gemm_split3): the producer writes
    x = resid + a W0^T + b0 as fp32 AND as planes AND its per-row strip sums; mmsa_rowstats_finalize turns them into (mean, rstd); the
    consumer runs on the raw planes against W o w and normalises in its epilogue -- against LayerNorm + Linear in float64, with a stream
    whose mean is not small against its spread."""
    fmt = ops.FMT_H8 if fmt_name == "h8" else ops.FMT_B3
    h8 = fmt == ops.FMT_H8
    M, D, K0, N = 1024, 512, 256, 384
    a0 = torch.randn(M, K0, generator=g(201))
    w0 = torch.randn(D, K0, generator=g(202)) / K0 ** 0.5
    b0 = torch.randn(D, generator=g(203)) * 0.1
    res = torch.randn(M, D, generator=g(204)) + 1.5          # |mean| ~ 1.5 std
    lnw, lnb = 1.0 + 0.1 * torch.randn(D, generator=g(205)), 0.05 * torch.randn(D, generator=g(206))
    w = torch.randn(N, D, generator=g(207)) / D ** 0.5
    b = torch.randn(N, generator=g(208)) * 0.1
    x64 = res.double() + a0.double() @ w0.double().t() + b0.double()
    ref = F.layer_norm(x64, (D,), lnw.double(), lnb.double(), 1e-6) @ w.double().t() + b.double()
    if act == "gelu":
        ref = F.gelu(ref)
    ap = ops.split_planes(a0.to(DEV), kpad=K0, fmt=fmt)
    w0p = ops.split_planes(w0.to(DEV), fmt=fmt, weight=h8)
    x = res.to(DEV).clone()
    xp = ops.alloc_planes(M, D, DEV, fmt=fmt)
    rs = torch.full((M, 2 * (D // 64)), float("nan"), device=DEV)
    ops.gemm(ap, w0p, x, bias=b0.to(DEV), resid=x, out_planes=xp, rowstats_out=rs)
    assert_close(x, x64.float(), tol=3e-5, what="producer fp32 output")
    assert_close(planes_to_float(xp), x64.float(), tol=6e-5, what="producer planes output")
    want = torch.stack([x.double().view(M, D // 64, 64).sum(-1), (x.double() ** 2).view(M, D // 64, 64).sum(-1)], -1).view(M, -1)
    assert_close(rs, want.float(), tol=1e-6, what="row strip sums")
    mr = torch.empty(M, 2, device=DEV)
    ops.rowstats_finalize(rs, M, D, 1e-6, mr)
    assert_close(mr[:, 0], x.double().mean(1).float(), tol=1e-6, what="row mean")
    assert_close(mr[:, 1], (x.double().var(1, unbiased=False) + 1e-6).rsqrt().float(), tol=2e-6, what="row rstd")
    wp = ops.split_planes((w * lnw[None, :]).to(DEV).contiguous(), fmt=fmt, weight=h8)
    cs = planes_to_float(wp)[:N, :D].double().sum(1).float().contiguous()
    bf = (w.double() @ lnb.double()).float().add(b).to(DEV).contiguous()
    out = ops.alloc_planes(M, N, DEV, fmt=fmt)
    ops.gemm(xp, wp, bias=bf, act=act, out_planes=out, row_norm=(mr, cs))
    assert_close(planes_to_float(out), ref.float(), tol=1e-4 if h8 else 4e-5, what=f"folded LayerNorm + Linear ({fmt_name})")
    # the extras are one-shot: the same call without them is a plain GEMM again
    out2 = ops.alloc_planes(M, N, DEV, fmt=fmt)
    ops.gemm(xp, wp, bias=bf, act=act, out_planes=out2)
    assert not torch.equal(out2.p, out.p)
    with pytest.raises(RuntimeError, match="row statistics"):
        ops.gemm(ap, w0p, x, bias=b0.to(DEV), act="gelu", rowstats_out=rs)


# ---- f3 planes (round 4): the bf16 hi/lo layout with fp16 halves -- 22 significant bits, the same three MFMAs per product
@pytest.mark.parametrize("M,N,K,act", [(512, 384, 1536, "none"), (300, 1536, 384, "gelu"), (96, 200, 96, "relu")])
def test_gemm_f3_operands(ops, M, N, K, act):
    """fp16 hi/lo operands: round trip of the planes (2^-21 relative), the GEMM against fp64 (an order of magnitude inside what bf16 hi/lo
    gives on the same data), against its own emulation (fp32 accumulation order only), planes outputs in f3 and bf16 hi/lo, clamp at the fp16 range."""
    a = torch.randn(M, K, generator=g(341)) * 1.7
    w = torch.randn(N, K, generator=g(342)) / K ** 0.5
    b = torch.randn(N, generator=g(343))
    res = torch.randn(M, N, generator=g(344))
    fact = {"none": lambda t: t, "gelu": F.gelu, "relu": F.relu}[act]
    ref = fact(F.linear(a.double(), w.double(), b.double())).float() + res
    ap = ops.split_planes(a.to(DEV), fmt=ops.FMT_F3)
    wp = ops.split_planes(w.to(DEV), fmt=ops.FMT_F3)
    back = planes_to_float(ap)[:, :K].cpu()
    assert ((back - a).abs() <= a.abs() * 2.0 ** -21 + 1e-7).all(), "f3 planes carry 22 significant bits"
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(ap, wp, out, bias=b.to(DEV), act=act, resid=res.to(DEV))
    assert_close(out, ref, tol=2e-6, what="f3 gemm vs fp64")
    outb = torch.empty(M, N, device=DEV)
    ops.gemm(ops.split_planes(a.to(DEV), kpad=ap.kpad), ops.split_planes(w.to(DEV)), outb, bias=b.to(DEV), act=act, resid=res.to(DEV))
    e3 = ((out.cpu().double() - ref.double()).norm() / ref.double().norm()).item()
    eb = ((outb.cpu().double() - ref.double()).norm() / ref.double().norm()).item()
    assert e3 < 0.25 * eb, f"f3 {e3:.2e} should be well inside bf16 hi/lo {eb:.2e}"
    for ofmt in (ops.FMT_F3, ops.FMT_B3):
        outp = ops.alloc_planes(M, N, DEV, zero=True, fmt=ofmt)
        ops.gemm(ap, wp, bias=b.to(DEV), act=act, resid=res.to(DEV), out_planes=outp)
        assert_close(planes_to_float(outp), ref, tol=3e-6 if ofmt == ops.FMT_F3 else 2e-5, what=f"f3 gemm planes out fmt {ofmt}")
        if ofmt == ops.FMT_F3:     # bit-identical to splitting the fp32 output of the same GEMM
            assert torch.equal(planes_to_float(outp), planes_to_float(ops.split_planes(out, fmt=ops.FMT_F3)))
    # f3 planes out of bf16 hi/lo operands, and values beyond the fp16 range are clamped, not turned into inf
    outp = ops.alloc_planes(M, N, DEV, zero=True, fmt=ops.FMT_F3)
    ops.gemm(ops.split_planes(a.to(DEV), kpad=ap.kpad), ops.split_planes(w.to(DEV)), bias=b.to(DEV), act=act, resid=res.to(DEV), out_planes=outp)
    assert_close(planes_to_float(outp), ref, tol=2e-5, what="bf16 hi/lo gemm, f3 planes out")
    big = torch.tensor([[1e6, -1e6, 70000.0, 3.0e-6, -2.0e-7, 1.0] + [0.0] * 26], device=DEV)
    bp = planes_to_float(ops.split_planes(big, fmt=ops.FMT_F3)).cpu()[0]
    assert torch.isfinite(bp).all() and bp[0] == 65504.0 and bp[1] == -65504.0 and bp[2] == 65504.0
    assert abs(bp[3].item() - 3.0e-6) <= 6.1e-8 and abs(bp[4].item() + 2.0e-7) <= 6.1e-8 and bp[5] == 1.0


def test_layernorm_and_batched_gemm_f3(ops):
    """LayerNorm writing f3 planes (whole-line pair stores and the 4-column path) and the batched two-stream GEMM chain of a ConvNeXt block on them."""
    rows, C = 640, 384
    x = torch.randn(2 * rows, C, generator=g(351)) * 2 + 0.3
    lw = torch.randn(2, C, generator=g(352)); lb = torch.randn(2, C, generator=g(353))
    n = ops.alloc_planes(2 * rows, C, DEV, fmt=ops.FMT_F3)
    ops.layernorm(x.to(DEV), lw.to(DEV), lb.to(DEV), 1e-6, out_planes=n, group_rows=rows, w_gstride=C)
    ref = torch.cat([F.layer_norm(x[i * rows:(i + 1) * rows].double(), (C,), lw[i].double(), lb[i].double(), 1e-6) for i in range(2)], 0).float()
    assert_close(planes_to_float(n), ref, tol=2e-6, what="layernorm f3 planes")
    w1 = torch.randn(2, 4 * C, C, generator=g(354)) / C ** 0.5; b1 = torch.randn(2, 4 * C, generator=g(355))
    w1p = [ops.split_planes(w1[i].to(DEV), fmt=ops.FMT_F3) for i in range(2)]
    full = torch.cat([w1p[0].p, w1p[1].p], 0).contiguous()
    wpl = ops.Planes(full[:4 * C], 4 * C, C, w1p[0].kpad, ops.FMT_F3, False)
    h = ops.alloc_planes(2 * rows, 4 * C, DEV, fmt=ops.FMT_F3)
    ops.gemm(n, wpl, bias=b1.reshape(-1).to(DEV), act="gelu", out_planes=h, batch=2, m=rows, stride_a=rows * 2 * n.kpad,
             stride_w=4 * C * 2 * wpl.kpad, stride_bias=4 * C, stride_cp=rows * 2 * h.kpad)
    href = torch.cat([F.gelu(F.linear(ref[i * rows:(i + 1) * rows].double(), w1[i].double(), b1[i].double())) for i in range(2)], 0).float()
    assert_close(planes_to_float(h), href, tol=3e-6, what="batched f3 gemm + gelu, f3 planes out")


# ---- round 5: the register-resident tile epilogue (csrc/gemm_epilogue_regs.inc)
_REGS_CASES = [
    # (operand fmt, output kind, plane fmt out, act, resid, strip sums, row-norm)   -- one per compile-time variant the model runs
    ("h8c", "P", "h8c", "gelu", False, False, True),     # lin1 (IE:162-167) with norm2 folded in
    ("h8c", "P", "h8c", "none", False, False, False),
    ("h8c", "P", "h8", "none", False, False, True),      # qkv (IE:488) with norm1 folded in, h8 line planes for the attention kernels
    ("h8c", "P", "h8", "gelu", False, False, False),
    ("f3", "P", "f3", "gelu", False, False, False),      # ConvNeXt pointwise_conv1 (TC:107-109)
    ("b3", "P", "b3", "none", False, False, False),
    ("h8c", "CP", "h8c", "none", True, True, False),     # proj / lin2 (IE:499,167): the residual stream's producers
    ("f3", "CP", "f3", "none", True, True, False),       # the same for a block that fell back to fp16 pairs
    ("h8c", "C", None, "none", True, False, False),      # extractor output projection, ConvFFN fc2 (AM:447-451)
    ("f3", "C", None, "none", True, False, False),       # ConvNeXt pointwise_conv2 + gamma + residual (TC:110-132)
    ("h8", "C", None, "none", False, False, False),      # value / offset projections, fc1
    # the "rows" form of fp32-only outputs (no per-column scale): whole 256-byte row segments through one LDS transpose per sub-tile
    ("h8c", "C", None, "none", True, False, False, "plain"),     # extractor output projection (AM:494-497), ConvFFN fc2
    ("h8", "C", None, "none", True, False, False, "plain"),
    ("h8c", "C", None, "none", False, False, False, "plain"),    # value / offset projections, fc1
    ("b3", "C", None, "none", False, False, False, "plain"),     # the neck's fc convs (AM:947-950); K <= 256: the 4-wave flavour
    ("h8c", "C", None, "none", False, False, True, "plain"),     # the same with the adapter tokens' LayerNorm folded in (row-normalising form on an fp32 output)
]


@pytest.mark.parametrize("case", _REGS_CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_gemm_register_epilogue_variants(ops, case):
    """Every compile-time variant of the register-resident epilogue on INTERIOR tiles (M % 256 == 0, N % 128 == 0, several tiles per
    workgroup, two batches with per-batch bias / colscale) against float64 -- and bit for bit against the LDS-staged epilogue, which the
    same rows take when they are launched as a ragged matrix (M = 200 < one tile): both epilogues apply the same operations in the same
    order to the same accumulators.  Strip sums: different summation order, so within fp32 rounding."""
    fmt_name, outk, pfmt_name, act, use_res, use_rs, use_rn = case[:7]
    plain = len(case) > 7            # no per-column scale, alpha = 1
    FM = {"b3": ops.FMT_B3, "h8": ops.FMT_H8, "h8c": ops.FMT_H8C, "f3": ops.FMT_F3}
    fmt = FM[fmt_name]
    B, M, N, K = 2, 768, 384, 256
    a = torch.randn(B * M, K, generator=g(501)) * 1.3
    w = torch.randn(B * N, K, generator=g(502)) / K ** 0.5
    bias = torch.randn(B * N, generator=g(503))
    cs = 0.5 + torch.rand(B * N, generator=g(504))
    res = torch.randn(B * M, N, generator=g(505))
    mr = torch.stack([torch.randn(B * M, generator=g(506)) * 0.2, 0.5 + torch.rand(B * M, generator=g(507))], 1).contiguous()
    csum = torch.randn(B * N, generator=g(508))
    fact = {"none": lambda t: t, "gelu": F.gelu}[act]
    ad, wd = a.to(DEV), w.to(DEV)
    ap = ops.split_planes(ad, kpad=K, fmt=fmt)
    wp_all = ops.split_planes(wd, fmt=fmt, weight=fmt == ops.FMT_H8)
    wp = ops.Planes(wp_all.p, N, K, wp_all.kpad, fmt, fmt == ops.FMT_H8)
    af, wf = planes_to_float(ap)[:, :K].double().cpu(), planes_to_float(wp_all)[:, :K].double().cpu()   # the operands as the kernel sees them
    acc = torch.stack([af[b * M:(b + 1) * M] @ wf[b * N:(b + 1) * N].t() for b in range(B)]).view(B * M, N)
    bb, cc = bias.double().view(B, 1, N).expand(B, M, N).reshape(B * M, N), cs.double().view(B, 1, N).expand(B, M, N).reshape(B * M, N)
    if use_rn:
        pre = mr[:, 1:2].double() * (acc - mr[:, 0:1].double() * csum.double().view(B, 1, N).expand(B, M, N).reshape(B * M, N)) + bb
    else:
        pre = acc + bb
    ref = fact(pre) * (1.0 if plain else cc * 0.75)
    if use_res:
        ref = ref + res.double()
    ref = ref.float()

    def run(rows, m_arg, batch):
        kw = dict(bias=bias.to(DEV), act=act, batch=batch, m=m_arg, stride_a=ap.batch_stride(M), stride_w=wp_all.batch_stride(N), stride_bias=N)
        if not plain:
            kw.update(colscale=cs.to(DEV), alpha=0.75)
        out = outp = rs = None
        if "C" in outk:
            out = torch.full((B * M, N), float("nan"), device=DEV)
            kw.update(out=out, stride_c=M * N)
        if "P" in outk:
            outp = ops.alloc_planes(B * M, N, DEV, zero=True, fmt=FM[pfmt_name])
            kw.update(out_planes=outp, stride_cp=outp.batch_stride(M))
        if use_res:
            kw.update(resid=res.to(DEV), stride_r=M * N)
        if use_rs:
            rs = torch.full((B * M, 2 * (N // 64)), float("nan"), device=DEV)
            kw.update(rowstats_out=rs)
        if use_rn:
            kw.update(row_norm=(mr.to(DEV), csum.to(DEV)))
        ops.gemm(ap, wp, **kw)
        return out, outp, rs

    out, outp, rs = run(B * M, M, B)
    ctol = {"b3": 3e-5, "f3": 3e-6, "h8": 1e-4, "h8c": 1e-4}     # the operand format's own product error (cross terms on e5m2, no lo x lo term)
    ptol = {"b3": 3e-5, "f3": 3e-6, "h8": 1.5e-4, "h8c": 1.5e-4}   # + the rounding of the stored planes
    if out is not None:
        assert_close(out, ref, tol=ctol[fmt_name], what=f"{case}: fp32 output vs float64 on the kernel's operands")
    if outp is not None:
        assert_close(planes_to_float(outp), ref, tol=max(ptol[pfmt_name], ctol[fmt_name]), what=f"{case}: planes output")
        if out is not None:   # the planes are the split of the stored fp32 values
            assert torch.equal(planes_to_float(outp), planes_to_float(ops.split_planes(out, fmt=FM[pfmt_name])))
    if rs is not None:
        x = out.double().view(B * M, N // 64, 64)
        want = torch.stack([x.sum(-1), (x * x).sum(-1)], -1).view(B * M, -1).float()
        assert_close(rs, want, tol=2e-6, what=f"{case}: strip sums")
    # the staged epilogue on the same rows: batch 0 only, its first 200 rows as a ragged one-tile-row matrix (200 < 256: no interior tile)
    if not (use_rn and outk == "C"):     # (the row-normalising form on an fp32 output exists for whole tiles only: the launcher refuses ragged M)
        out2, outp2, rs2 = run(200, 200, 1)
        if out is not None:
            assert torch.equal(out2[:200], out[:200]), f"{case}: register and staged epilogues differ (fp32)"
        if outp is not None:
            assert torch.equal(planes_to_float(outp2)[:200], planes_to_float(outp)[:200]), f"{case}: register and staged epilogues differ (planes)"
        if rs is not None:
            assert_close(rs2[:200], rs[:200], tol=2e-6, what=f"{case}: strip sums, staged vs register epilogue")
    if use_rn and outk == "C":
        with pytest.raises(RuntimeError, match="row-normalising"):
            run(200, 200, 1)
    again = run(B * M, M, B)
    for t0, t1 in zip((out, outp.p if outp is not None else None, rs), (again[0], again[1].p if again[1] is not None else None, again[2])):
        if t0 is not None:
            assert torch.equal(t0, t1), f"{case}: not reproducible"
