"""GPU-side BIT-EXACT check of the window bookkeeping (north_star: "bit-exact for window index/permute bookkeeping"):
window_partition + zero-pad (IE:504-528) and window_unpartition + crop (IE:531-551) exist on the device only as address
arithmetic inside the windowed-attention kernels, so an index-valued tensor is pushed THROUGH those kernels and the result is
compared as integers with the reference's own partition tables (tests/golden/bookkeeping.npz, written by the unmodified
reference functions on an index tensor).

How attention is made to reveal the permutation: every token's K is a +-a code of its in-window position p (taken from the
golden table), its Q is the code of the position NEXT to it in the same window (the next non-pad slot, cyclically), its V carries
its own token id.  The diagonal-dominant logits make the softmax exactly one-hot in fp32 (margin > 60 in the exponent), so the
kernel returns for token t the id (to ~1e-7 relative; rounded to the integer) of the token that the reference's window_partition
places in the successor slot of t's window; a token gathered into a wrong window / slot, a pad slot that is not bias-valued, or an output scattered to a wrong
row changes integers.  Ids < 2^16 are exact in the bf16 hi+lo planes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GEOMS = [(14, 14, 14), (16, 16, 14), (20, 20, 14), (64, 64, 14), (32, 32, 14)]   # all five of tests/golden/bookkeeping.npz


def _codes(n, hd, seed, max_corr):
    """n distinct +-1 codes of length hd whose pairwise correlations stay <= max_corr * hd (verified)."""
    g = torch.Generator().manual_seed(seed)
    for _ in range(200):
        c = (torch.randint(0, 2, (n, hd), generator=g) * 2 - 1).float()
        gram = c @ c.t()
        gram.fill_diagonal_(-hd)
        if gram.max().item() <= hd * max_corr:
            return c
    raise AssertionError("no code set with the required margin")


def _expected_and_qkv(win, B, H, W, ws, hd, amp, max_corr):
    """win: golden [B*nW, ws, ws] (token id + 1, 0 = pad).  Returns qkv fp32 [B*H*W, 3*hd] and expected ids [B*H*W]."""
    nslot = ws * ws
    codes = _codes(nslot, hd, 1000 + H, max_corr) * amp
    flat = torch.from_numpy(win).reshape(-1, nslot)            # [windows, slots]
    T = B * H * W
    q = torch.zeros(T, hd)
    k = torch.zeros(T, hd)
    v = torch.zeros(T, hd)
    expect = torch.zeros(T, dtype=torch.int64)
    for wrow in flat:
        slots = [s for s in range(nslot) if wrow[s] > 0]
        for i, s in enumerate(slots):
            t = int(wrow[s]) - 1
            s_next = slots[(i + 1) % len(slots)]
            k[t] = codes[s]
            q[t] = codes[s_next]
            v[t, :] = float(t + 1)            # every channel carries the id: all 64 output channels are checked
            expect[t] = int(wrow[s_next])
    return torch.cat([q, k, v], 1).contiguous(), expect


@pytest.mark.parametrize("vf", [False, True])
@pytest.mark.parametrize("H,W,ws", GEOMS)
def test_window_bookkeeping_bit_exact_fused_kernel(golden_dir, H, W, ws, vf):
    """wattn_persist_kernel (head_dim 64: the ViT-B / ViT-L path): partition, 64->70-style padding, unpartition as integers.
    vf: the production form -- h8 planes, every contraction on the fp16 MFMA; fp16 holds integers up to 2048 exactly, so the id
    travels as two base-1024 digits (channels 0..31 the low digit, 32..63 the high one)."""
    import mmsa
    from mmsa import ops
    g = np.load(os.path.join(golden_dir, "bookkeeping.npz"))
    win = g[f"wp_{H}_{W}_{ws}"]
    B, hd = 2, 64
    qkv, expect = _expected_and_qkv(win, B, H, W, ws, hd, amp=4.0, max_corr=0.6)     # logits: 64*16/8 = 128 on the match, <= 0.6*128 elsewhere
    if vf:
        ids = qkv[:, 2 * hd].clone()
        qkv[:, 2 * hd:2 * hd + 32] = torch.remainder(ids, 1024.0)[:, None]
        qkv[:, 2 * hd + 32:] = torch.floor(ids / 1024.0)[:, None]
        qp = ops.split_planes(qkv.to(DEV), fmt=ops.FMT_H8)
        bias_p = ops.split_planes(torch.zeros(1, 3 * hd, device=DEV), kpad=3 * hd, fmt=ops.FMT_H8)
    else:
        qp = ops.split_planes(qkv.to(DEV), fmt=ops.FMT_F3)       # the kernels' hi/lo form reads fp16 pairs (f3 planes)
        bias_p = ops.split_planes(torch.zeros(1, 3 * hd, device=DEV), kpad=3 * hd, fmt=ops.FMT_F3)    # pad slots: k = v = bias = 0
    relp = ops.window_relpos_planes(torch.zeros(2 * ws - 1, hd, device=DEV), torch.zeros(2 * ws - 1, hd, device=DEV), ws,
                                    fmt=ops.FMT_H8 if vf else ops.FMT_F3)
    out = ops.alloc_planes(B * H * W, hd, DEV)
    ops.window_attention(qp, bias_p, relp, out, B, H, W, 1, hd, ws, hd ** -0.5)
    got = ops.planes_to_float(out).cpu()
    # the softmax is one-hot up to the rounding of exp2 / the normalisation (1e-7 relative): ids are recovered by rounding
    assert (got - got.round()).abs().max() < 1e-2, "the softmax was not one-hot"
    ids = got.round().to(torch.int64)
    if vf:
        ids = (ids[:, :32] + 1024 * ids[:, 32:]).repeat(1, 2)
    assert torch.equal(ids, expect[:, None].expand(-1, hd)), f"window bookkeeping differs for {H}x{W} ws={ws}"
    # and the inverse table of the reference: window_unpartition(window_partition(idx)) == idx, i.e. every token was written
    assert torch.equal(torch.from_numpy(g[f"wu_{H}_{W}_{ws}"]).reshape(-1), torch.arange(1, B * H * W + 1))


@pytest.mark.parametrize("H,W,ws", GEOMS[:3] + GEOMS[4:])
def test_window_bookkeeping_bit_exact_generic_kernel(golden_dir, H, W, ws):
    """attn_kernel's windowed mode (head_dim 32: the tiny models) through mmsa_attention_planes."""
    import mmsa
    from mmsa import ops
    g = np.load(os.path.join(golden_dir, "bookkeeping.npz"))
    win = g[f"wp_{H}_{W}_{ws}"]
    B, hd = 2, 32
    qkv, expect = _expected_and_qkv(win, B, H, W, ws, hd, amp=8.0, max_corr=0.75)     # logits: 32*64/sqrt(32) = 362 on the match, <= 272 elsewhere
    qp = ops.split_planes(qkv.to(DEV), fmt=ops.FMT_F3)
    bias_p = ops.split_planes(torch.zeros(1, 3 * hd, device=DEV), kpad=3 * hd, fmt=ops.FMT_F3)
    rp = torch.zeros(B * H * W, 2 * ws, device=DEV)
    out = ops.alloc_planes(B * H * W, hd, DEV)
    ops.attention(qp, bias_p, rp, out, B, H, W, 1, hd, ws, hd ** -0.5)
    got = ops.planes_to_float(out).cpu()
    assert (got - got.round()).abs().max() < 1e-2
    assert torch.equal(got.round().to(torch.int64), expect[:, None].expand(-1, hd)), f"window bookkeeping differs for {H}x{W} ws={ws}"


def test_rel_pos_gather_index_bit_exact_on_device(golden_dir):
    """get_rel_pos's float -> long gather table (IE:579-584) as used by the device path: the packed tables re-indexed by the
    kernel must select the reference's rows.  With one-hot q the rel-pos term IS the selected table entry."""
    import mmsa.backbone as bb
    g = np.load(os.path.join(golden_dir, "bookkeeping.npz"))
    for (q, L) in ((14, 27), (64, 127), (14, 31), (20, 31), (16, 31), (32, 127)):
        tab = torch.from_numpy(g[f"rp_in_{q}_{L}"]).to(DEV)
        got = bb._rel_table(q, tab).cpu()
        want = torch.from_numpy(g[f"rp_out_{q}_{L}"])
        if L == 2 * q - 1:      # pure integer gather: bit-exact
            assert torch.equal(got, want), f"rel-pos table ({q}, {L})"
        else:                   # table first resized by linear interpolation (IE:568-575; fp32 on the device): same rows to rounding
            assert (got - want).abs().max() <= 2e-6 * want.abs().max(), f"rel-pos table ({q}, {L})"
