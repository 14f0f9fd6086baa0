"""Tensor-level wrappers over the C ABI (include/mmsa.h).  PyTorch is only plumbing here: device memory,
the current HIP stream and shape bookkeeping.  No arithmetic is done by torch on this path, and nothing
falls back to torch when the library fails -- errors propagate as RuntimeError."""
import ctypes

import torch

from . import lib

ACT = {"none": 0, "gelu": 1, "relu": 2, "relu6": 3, "hswish": 4, "sigmoid": 5}

# When set to a list, every gemm() launch is bracketed by HIP events recorded on the launch stream and
# (flops, start, stop) is appended; bench.py uses this for the roofline of the dominant kernel.
GEMM_PROFILE = None


def _event():
    ev = ctypes.c_void_p()
    lib.call("mmsa_event_create", ctypes.byref(ev))
    return ev


def collect_gemm_profile(prof):
    """-> (total algorithmic FLOPs, total kernel milliseconds) of the recorded launches; frees the events."""
    flops, ms = 0.0, 0.0
    for f, e0, e1 in prof:
        t = ctypes.c_float()
        lib.call("mmsa_event_elapsed_ms", e0, e1, ctypes.byref(t))
        flops += f
        ms += t.value
        lib.call("mmsa_event_destroy", e0)
        lib.call("mmsa_event_destroy", e1)
    return flops, ms


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"mmsa: {name} must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"mmsa: {name} must be {dtype}, got {t.dtype}")
    return t.data_ptr()


def _mat(t, name, dtype=torch.float32):
    """2-D matrix view with unit column stride -> (ptr, rows, cols, ld)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError(f"mmsa: {name} must be a 2-D view with unit column stride, got shape {tuple(t.shape)} stride {t.stride()}")
    return _chk(t, dtype, name), t.shape[0], t.shape[1], t.stride(0)


class Planes:
    """bf16 hi/lo planes (int16 storage) of a matrix: weights [N, Kpad] or activations [rows, cols]."""

    def __init__(self, hi, lo, n=None, k=None, kpad=None):
        self.hi, self.lo = hi, lo
        self.n = hi.shape[0] if n is None else n
        self.k = hi.shape[1] if k is None else k
        self.kpad = hi.shape[1] if kpad is None else kpad

    def __getitem__(self, idx):
        return Planes(self.hi[idx], self.lo[idx])

    def mats(self, name):
        ph, rows, cols, ld = _mat(self.hi, name + ".hi", torch.int16)
        pl, _, _, ld2 = _mat(self.lo, name + ".lo", torch.int16)
        if ld != ld2 or self.hi.shape != self.lo.shape:
            raise RuntimeError(f"mmsa: {name} hi/lo planes must share shape and stride")
        return ph, pl, rows, cols, ld


def pad32(k):
    return (k + 31) // 32 * 32


def alloc_planes(rows, cols, device, zero=False):
    f = torch.zeros if zero else torch.empty
    return Planes(f(rows, cols, dtype=torch.int16, device=device), f(rows, cols, dtype=torch.int16, device=device))


def split_planes(w2d, kpad=None, out=None):
    """fp32 [N, K] (device) -> Planes with K zero-padded to a multiple of 32."""
    p, n, k, ld = _mat(w2d, "weight")
    kpad = kpad or pad32(k)
    if out is None:
        out = Planes(torch.empty(n, kpad, dtype=torch.int16, device=w2d.device),
                     torch.empty(n, kpad, dtype=torch.int16, device=w2d.device), n, k, kpad)
    lib.call("mmsa_split_planes", p, ld, n, k, kpad, out.hi.data_ptr(), out.lo.data_ptr(), _stream())
    return out


def gemm(a, w, out=None, bias=None, act="none", alpha=1.0, colscale=None, resid=None, beta=1.0, resid_mod=0,
         batch=1, stride_a=0, stride_w=0, stride_bias=0, stride_r=0, stride_c=0, m=None, pixel_shuffle=None,
         out_planes=None, stride_cp=0):
    """out / out_planes = beta*resid + colscale*alpha*act(a @ w^T + bias).
    a: fp32 2-D view or activation Planes; w: weight Planes; out: fp32 view and/or out_planes: Planes."""
    if isinstance(a, Planes):
        pah, pal, ma, ka, lda = a.mats("A")
        pa = None
    else:
        pa, ma, ka, lda = _mat(a, "A")
        pah = pal = None
    m = ma if m is None else m
    if ka < w.kpad and lda < w.kpad:
        raise RuntimeError(f"mmsa.gemm: A has {ka} columns (ld {lda}) but the packed weight expects K={w.kpad}")
    pc, ldc = None, 0
    if out is not None:
        pc, _, _, ldc = _mat(out, "C")
    pch = pcl = None
    ldcp = 0
    if out_planes is not None:
        pch, pcl, _, _, ldcp = out_planes.mats("Cplanes")
    pr, ldr = None, 0
    if resid is not None:
        pr, _, _, ldr = _mat(resid, "resid")
    ps = pixel_shuffle or (0, 0, 0)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = _event(), _event()
        lib.call("mmsa_event_record", e0, _stream())
    lib.call("mmsa_gemm_split3", pa, pah, pal, lda, stride_a, w.hi.data_ptr(), w.lo.data_ptr(), stride_w,
             _chk(bias, name="bias"), stride_bias, _chk(colscale, name="colscale"), pr, ldr, stride_r, resid_mod, beta,
             pc, ldc, stride_c, pch, pcl, ldcp, stride_cp, m, w.n, w.kpad, batch, ACT[act], alpha,
             1 if pixel_shuffle else 0, ps[0], ps[1], ps[2], _stream())
    if prof is not None:
        lib.call("mmsa_event_record", e1, _stream())
        prof.append((2.0 * m * w.n * w.k * batch, e0, e1))
    return out if out is not None else out_planes


def layernorm(x, w, b, eps, out=None, out2=None, patchify=None, out_planes=None):
    px, rows, c, ldx = _mat(x, "x")
    py, ldy = None, 0
    if out is not None:
        py, _, _, ldy = _mat(out, "y")
    p2, ld2 = None, 0
    if out2 is not None:
        p2, _, _, ld2 = _mat(out2, "y2")
    ph = pl = None
    ldp = 0
    if out_planes is not None:
        ph, pl, _, _, ldp = out_planes.mats("y planes")
    mh, mw = patchify or (0, 0)
    lib.call("mmsa_layernorm_rows", px, ldx, _chk(w), _chk(b), eps, py, ldy, p2, ld2, ph, pl, ldp, rows, c,
             1 if patchify else 0, mh, mw, _stream())
    return out if out is not None else out_planes


def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    """Drop-in for MultiScaleDeformableAttention.ms_deform_attn_forward (ops/src/vision.cpp:14)."""
    for t, n in ((value, "value"), (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight")):
        if not t.is_contiguous():
            raise RuntimeError(f"{n} tensor has to be contiguous")  # ms_deform_attn_cuda.cu:28-32
    if not (spatial_shapes.is_contiguous() and level_start_index.is_contiguous()):
        raise RuntimeError("spatial_shapes / level_start_index tensor has to be contiguous")
    n, s, m, d = value.shape
    _, lq, _, l, p, _ = sampling_loc.shape
    out = torch.empty(n, lq, m * d, dtype=torch.float32, device=value.device)
    lib.call("mmsa_ms_deform_attn_forward", _chk(value), _chk(spatial_shapes, torch.int64), _chk(level_start_index, torch.int64),
             _chk(sampling_loc), _chk(attn_weight), out.data_ptr(), n, s, m, d, l, lq, p, im2col_step, _stream())
    return out


def msda_fused(value2d, spatial_shapes, level_start_index, raw, ref_points, out, batch, spatial, heads, d, levels, lq, points,
               out_planes=None):
    pv, _, _, _ = _mat(value2d, "value")
    pr, _, _, ldraw = _mat(raw, "raw")
    po, ldo = None, 0
    if out is not None:
        po, _, _, ldo = _mat(out, "out")
    ph = pl = None
    ldp = 0
    if out_planes is not None:
        ph, pl, _, _, ldp = out_planes.mats("out planes")
    lib.call("mmsa_msda_fused", pv, _chk(spatial_shapes, torch.int64), _chk(level_start_index, torch.int64), pr, ldraw,
             _chk(ref_points), po, ldo, ph, pl, ldp, batch, spatial, heads, d, levels, lq, points, _stream())
    return out if out is not None else out_planes


def relpos_bias(qkv, rh, rw, rp, b, h, w, heads, hd, ws):
    if isinstance(qkv, Planes):
        ph, pl, _, _, ldq = qkv.mats("qkv")
        lib.call("mmsa_relpos_bias_planes", ph, pl, ldq, _chk(rh), _chk(rw), _chk(rp), b, h, w, heads, hd, ws, _stream())
    else:
        pq, _, _, ldq = _mat(qkv, "qkv")
        lib.call("mmsa_relpos_bias", pq, ldq, _chk(rh), _chk(rw), _chk(rp), b, h, w, heads, hd, ws, _stream())
    return rp


def attention(qkv, qkv_bias, rp, out, b, h, w, heads, hd, ws, scale):
    if isinstance(qkv, Planes):
        ph, pl, _, _, ldq = qkv.mats("qkv")
        oh, ol, _, _, ldo = out.mats("out")
        lib.call("mmsa_attention_planes", ph, pl, ldq, _chk(qkv_bias.hi, torch.int16), _chk(qkv_bias.lo, torch.int16), _chk(rp),
                 oh, ol, ldo, b, h, w, heads, hd, ws, scale, _stream())
    else:
        pq, _, _, ldq = _mat(qkv, "qkv")
        po, _, _, ldo = _mat(out, "out")
        lib.call("mmsa_attention", pq, ldq, _chk(qkv_bias), _chk(rp), po, ldo, b, h, w, heads, hd, ws, scale, _stream())
    return out


def colstats(x, stride_b, b, hw, out, wrow=None):
    px, _, c, ldx = _mat(x, "x")
    lib.call("mmsa_colstats", px, ldx, stride_b, _chk(wrow), b, hw, c, _chk(out, torch.float64), _stream())
    return out


def ffrm_finalize(stats, b, hw, c, mean_w, mean_b, wc, gn_w, gn_b, mean_o, rstd_o, mult_o, scratch):
    lib.call("mmsa_ffrm_finalize", _chk(stats, torch.float64), b, hw, c, mean_w, mean_b, _chk(wc), _chk(gn_w), _chk(gn_b),
             _chk(mean_o), _chk(rstd_o), _chk(mult_o), _chk(scratch), _stream())


def lnhw_apply(x, mean, rstd, mult, w, bias, out, b, hw):
    px, _, c, ldx = _mat(x, "x")
    po, _, _, ldo = _mat(out, "y")
    lib.call("mmsa_lnhw_apply", px, ldx, _chk(mean), _chk(rstd), _chk(mult), _chk(w), _chk(bias), po, ldo, b, hw, c, _stream())
    return out


def dwconv(x, w, bias, out, b, h, wd, k, act="none", xstride_b=None, ystride_b=None, out_planes=None):
    """Depthwise conv; result to fp32 `out` and/or `out_planes` (which then share the row stride)."""
    px, _, c, ldx = _mat(x, "x")
    po, ldo = None, 0
    if out is not None:
        po, _, _, ldo = _mat(out, "y")
    ph = pl = None
    if out_planes is not None:
        ph, pl, _, _, ldp = out_planes.mats("y planes")
        if out is not None and ldp != ldo:
            raise RuntimeError("mmsa.dwconv: fp32 output and planes must share the row stride")
        ldo = ldp
    xs = h * wd * ldx if xstride_b is None else xstride_b
    ys = h * wd * ldo if ystride_b is None else ystride_b
    lib.call("mmsa_dwconv_nhwc", px, ldx, xs, _chk(w), _chk(bias), po, ph, pl, ldo, ys, b, h, wd, c, k, ACT[act], _stream())
    return out if out is not None else out_planes


def gconv(x, w, bias, out, b, h, wd, groups, cin_g, cout_g, k, act="none"):
    px, _, _, ldx = _mat(x, "x")
    po, _, _, ldo = _mat(out, "y")
    lib.call("mmsa_gconv_nhwc", px, ldx, _chk(w), _chk(bias), po, ldo, b, h, wd, groups, cin_g, cout_g, k, ACT[act], _stream())
    return out


def im2col_nchw(x, c0, cin, p, out):
    b, ctot, h, w = x.shape
    if not x.is_contiguous():
        raise RuntimeError("mmsa.im2col_nchw: input image must be contiguous NCHW")
    po, _, kpad, ld = _mat(out, "out")
    if ld != kpad:
        raise RuntimeError("mmsa.im2col_nchw: output must be dense")
    lib.call("mmsa_im2col_nchw", _chk(x), b, ctot, c0, cin, h, w, p, po, kpad, _stream())
    return out


def gram_tn(x, y, stride_b, g, b, p, nblk=1):
    px, _, c, ldx = _mat(x, "X")
    py, _, _, ldy = _mat(y, "Y")
    lib.call("mmsa_gram_tn", px, ldx, py, ldy, stride_b, _chk(g), b, p, c, nblk, _stream())
    return g


def chanattn_build(g, sq, sq_stride, sk, sk_stride, temp, wp, planes, b, c, heads):
    lib.call("mmsa_chanattn_build", _chk(g), sq, sq_stride, sk, sk_stride, _chk(temp), _chk(wp),
             planes.hi.data_ptr(), planes.lo.data_ptr(), b, c, planes.kpad, heads, _stream())


def gffm_build(e, px_, py_, b, c):
    lib.call("mmsa_gffm_build", _chk(e), px_.hi.data_ptr(), px_.lo.data_ptr(), py_.hi.data_ptr(), py_.lo.data_ptr(),
             b, c, px_.kpad, _stream())


def gelu_gate(x, out, c):
    px, rows, _, ldx = _mat(x, "x")
    po, _, _, ldo = _mat(out, "y")
    lib.call("mmsa_gelu_gate", px, ldx, po, ldo, rows, c, _stream())
    return out


def pool_hw(z, out, b, h, w):
    pz, _, c, ldz = _mat(z, "z")
    po, _, _, ldo = _mat(out, "out")
    lib.call("mmsa_pool_hw", pz, ldz, po, ldo, b, h, w, c, _stream())
    return out


def ca_apply(z, att, out, b, h, w):
    pz, _, c, ldz = _mat(z, "z")
    pa, _, _, lda = _mat(att, "att")
    po, _, _, ldo = _mat(out, "out")
    lib.call("mmsa_ca_apply", pz, ldz, pa, lda, po, ldo, b, h, w, c, _stream())
    return out


def tail_fuse(cmap, cstride_b, xtok, bn_scale, bn_shift, out, b, hc, wc, hx, wx):
    pc, _, c, ldc = _mat(cmap, "cmap")
    px, _, _, ldx = _mat(xtok, "xtok")
    lib.call("mmsa_tail_fuse", pc, ldc, cstride_b, px, ldx, _chk(bn_scale), _chk(bn_shift), _chk(out), b, hc, wc, hx, wx, c, _stream())
    return out
