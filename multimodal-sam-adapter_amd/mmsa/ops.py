"""Tensor-level wrappers over the C ABI (include/mmsa.h).  PyTorch is only plumbing here: device memory,
the current HIP stream and shape bookkeeping.  No arithmetic is done by torch on this path, and nothing
falls back to torch when the library fails -- errors propagate as RuntimeError."""
import ctypes

import os

import torch

from . import lib

ACT = {"none": 0, "gelu": 1, "relu": 2, "relu6": 3, "hswish": 4, "sigmoid": 5}

# When set to a list, every gemm() launch is bracketed by HIP events recorded on the launch stream and
# (flops, start, stop) is appended; bench.py uses this for the roofline of the dominant kernel.
GEMM_PROFILE = None
CLAMP_WATCH = None   # a 1-element fp32 device tensor: the clamp watch word handed to every plane-producing launch (include/mmsa.h "Clamp watch"); the backbone
                     # sets it around its forward (clamp_watch below) -- None = no watch
GEMM_MAX_GRID = 0    # > 0: cap on the persistent workgroups of every gemm() launch (mmsa.chains gives each concurrent chain its share of the CUs)
GEMM_FLAVOUR = 0     # 0: workgroup shape chosen by the library; 4 / 8 force the 128- / 256-row form of the LDS-DMA GEMM (tests, A/B runs; bit-identical results)
GEMM_SHAPES = None   # optional parallel list of (m, n, k, batch, act, has_resid, outputs) per profiled launch (tools/gemm_shapes.py)


def _event():
    ev = ctypes.c_void_p()
    lib.call("mmsa_event_create", ctypes.byref(ev))
    return ev


def collect_gemm_profile(prof):
    """-> (total algorithmic FLOPs, total kernel milliseconds) of the recorded launches; frees the events.
    collect_gemm_profile.bytes = compulsory bytes of the same launches (operands read once, outputs written once)."""
    flops, ms = 0.0, 0.0
    collect_gemm_profile.bytes = sum(p[3] for p in prof)
    collect_gemm_profile.launches = []     # (FLOPs, compulsory bytes, milliseconds) per launch, in launch order
    for f, e0, e1, by in prof:
        t = ctypes.c_float()
        lib.call("mmsa_event_elapsed_ms", e0, e1, ctypes.byref(t))
        flops += f
        ms += t.value
        collect_gemm_profile.launches.append((f, by, t.value))
        lib.call("mmsa_event_destroy", e0)
        lib.call("mmsa_event_destroy", e1)
    return flops, ms


def _stream():
    """The current stream of the current device.  Every entry point runs under `torch.cuda.device(tensor.device)` (the module
    forwards set it) and `_chk` refuses tensors of another device, so a launch never goes to device A's stream with device B's memory."""
    return torch.cuda.current_stream().cuda_stream


def _clamp_ptr():
    """Device address of the clamp watch word of the forward in progress (None outside one): include/mmsa.h "Clamp watch"."""
    return CLAMP_WATCH.data_ptr() if CLAMP_WATCH is not None else None


class clamp_watch:
    """with ops.clamp_watch(word): every plane-producing launch inside reports values it had to clamp into `word` (a 1-element fp32 device tensor)."""

    def __init__(self, word):
        self.word = word

    def __enter__(self):
        global CLAMP_WATCH
        self.keep, CLAMP_WATCH = CLAMP_WATCH, self.word
        return self

    def __exit__(self, *exc):
        global CLAMP_WATCH
        CLAMP_WATCH = self.keep
        return False


def _chk(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"mmsa: {name} must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"mmsa: {name} must be {dtype}, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"mmsa: {name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call under torch.cuda.device(tensor.device)")
    return t.data_ptr()


def _mat(t, name, dtype=torch.float32):
    """2-D matrix view with unit column stride -> (ptr, rows, cols, ld)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError(f"mmsa: {name} must be a 2-D view with unit column stride, got shape {tuple(t.shape)} stride {t.stride()}")
    return _chk(t, dtype, name), t.shape[0], t.shape[1], t.stride(0)


FMT_B3, FMT_H8, FMT_H8C, FMT_F3 = 0, 1, 2, 3   # MMSA_FMT_* (include/mmsa.h); F3 = the B3 layout with fp16 halves (22 significant bits)


class Planes:
    """Operand planes (include/mmsa.h) of a matrix: ONE int16 tensor [rows, 2*kpad]; 128 bytes per row and 32-wide k-block.
    fmt FMT_B3: bf16 hi/lo ("split3": the 32 hi values then the 32 lo values).  fmt FMT_H8: fp16 hi + e5m2 cross-term bytes
    (32 fp16, then four 16-byte chunks of lo / q(hi) bytes; `weight` = the chunk order of a W operand).  Used for weights
    [N, K] and for activations; a producer writes the format of the Planes it is handed.
    fmt FMT_H8C (round 4): the h8 arithmetic on 3 bytes per element, stored by ROW PAIRS -- the tensor is [ceil(rows / 2), 3*kpad] int16
    (kpad % 64 == 0): a pair's two rows of kpad fp16 hi values, then kpad / 64 lines of 128 lo bytes (include/mmsa.h); `ld` handed to the
    library is the pair stride.  Row slices start at even rows; the same layout serves activations and weights."""

    def __init__(self, p, n=None, k=None, kpad=None, fmt=FMT_B3, weight=False, split=0):
        self.p = p
        self.split = split   # > 0 (multiple of 32, fmt FMT_B3): columns >= split are h8-encoded (the v third of qkv planes: fp16 hi for the attention kernels' P V)
        if fmt == FMT_H8C and (n is None or kpad is None):
            raise RuntimeError("mmsa.Planes: h8c planes need their row count and padded width (the tensor holds row pairs)")
        self.n = p.shape[0] if n is None else n
        self.kpad = p.shape[1] // 2 if kpad is None else kpad
        self.k = self.kpad if k is None else k
        self.fmt = fmt
        self.weight = weight
        self.gen = None   # (cell, value): set by a producer whose buffer is reused; live() tells whether it still holds this data

    def stamp(self, cell):
        """Mark these planes as valid while `cell[0]` keeps its current value (the producer bumps it when it reuses the buffer)."""
        self.gen = (cell, cell[0])
        return self

    def live(self):
        return self.gen is None or self.gen[0][0] == self.gen[1]

    def rows(self, lo, hi=None):
        """Row slice (same columns)."""
        if self.fmt == FMT_H8C:
            hi = self.n if hi is None else hi
            if lo % 2 or (hi % 2 and hi != self.n):
                raise RuntimeError("mmsa.Planes.rows: h8c planes are stored by row pairs -- slices start and end at even rows")
            return Planes(self.p[lo // 2:(hi + 1) // 2], hi - lo, self.k, self.kpad, self.fmt, self.weight)
        return Planes(self.p[lo:hi], None, self.k, self.kpad, self.fmt, self.weight, self.split)

    def cols(self, lo, hi):
        """Column slice [lo, hi) of the matrix (both multiples of 32): the k-blocks are self-contained in the layout."""
        if lo % 32 or hi % 32 or self.fmt == FMT_H8C:
            raise RuntimeError("mmsa.Planes.cols: column bounds must be multiples of 32 (and the planes bf16 hi/lo or h8 line planes)")
        return Planes(self.p[:, 2 * lo:2 * hi], self.n, hi - lo, hi - lo, self.fmt, self.weight)

    def batch_stride(self, rows):
        """uint16 elements between two batches of `rows` rows each inside these planes (the strided-batched GEMM's stride arguments)."""
        if self.fmt == FMT_H8C:
            if rows % 2:
                raise RuntimeError("mmsa.Planes.batch_stride: h8c batches hold an even number of rows")
            return (rows // 2) * self.p.stride(0)
        return rows * self.p.stride(0)

    def mat(self, name):
        ptr, rows, cols, ld = _mat(self.p, name, torch.int16)
        if ptr % 128 or ld % 64:
            raise RuntimeError(f"mmsa: {name} planes must be 128-byte aligned with a row stride that is a multiple of 64")
        if self.fmt == FMT_H8C:      # (pointer, logical rows, padded width, row-PAIR stride)
            if ld < 3 * self.kpad or self.kpad % 64:
                raise RuntimeError(f"mmsa: {name}: h8c planes need kpad % 64 == 0 and a pair stride >= 3 * kpad")
            return ptr, self.n, self.kpad, ld
        return ptr, rows, cols // 2, ld


def pad32(k):
    return (k + 31) // 32 * 32


def pad64(k):
    return (k + 63) // 64 * 64


def planes_shape(rows, cols, fmt):
    """(tensor rows, tensor columns, padded width) of the int16 tensor behind Planes of a [rows, cols] matrix."""
    if fmt == FMT_H8C:
        kp = pad64(cols)
        return (rows + 1) // 2, 3 * kp, kp
    kp = pad32(cols)
    return rows, 2 * kp, kp


def alloc_planes(rows, cols, device, zero=False, fmt=FMT_B3, split=0):
    f = torch.zeros if zero else torch.empty
    if split and (split % 32 or fmt not in (FMT_B3, FMT_F3) or not 0 < split < cols):
        raise RuntimeError(f"mmsa.alloc_planes: split={split} must be a multiple of 32 inside a bf16 hi/lo or f3 matrix of {cols} columns")
    tr, tc, kp = planes_shape(rows, cols, fmt)
    return Planes(f(tr, tc, dtype=torch.int16, device=device), rows, cols, kp, fmt, split=split)


def cp_format(pl):
    """The GEMM's output-format argument for these planes (include/mmsa.h: bits 0..7 format, bits 8.. = split / 32)."""
    return FMT_B3 if pl is None else pl.fmt | ((pl.split // 32) << 8)


def planes_to_float(pl, cols=None):
    """Debug/test helper: reconstruct hi + lo as fp32 [rows, cols] (torch ops; not used on the product path)."""
    r, w = pl.p.shape
    if pl.split:   # bf16 hi/lo columns below the split, h8 (activation chunk order) from it on
        lo_part = planes_to_float(Planes(pl.p[:, :2 * pl.split], pl.n, pl.split, pl.split, pl.fmt))
        hi_part = planes_to_float(Planes(pl.p[:, 2 * pl.split:], pl.n, pl.kpad - pl.split, pl.kpad - pl.split, FMT_H8))
        return torch.cat([lo_part, hi_part], 1)[:, :(pl.k if cols is None else cols)]
    if pl.fmt == FMT_H8C:
        kp = pl.kpad
        by = pl.p[:, :3 * kp].contiguous().view(torch.uint8).view(r, 6 * kp)
        hi = by[:, :4 * kp].contiguous().view(torch.float16).float().reshape(2 * r, kp)
        lo = by[:, 4 * kp:].reshape(r, kp // 64, 2, 4, 2, 8).permute(0, 2, 1, 4, 3, 5)        # [pair, row, chunk, k-tile, g, e]
        lo = lo.contiguous().view(torch.float8_e5m2).float().reshape(2 * r, kp)
        return (hi + lo / (2048.0 * 1.09375))[:pl.n, :(pl.k if cols is None else cols)]   # csrc/common.h MMSA_H8C_LO_COMP
    if pl.fmt == FMT_H8:
        blk = pl.p.contiguous().view(torch.uint8).view(r, w // 64, 128)
        hi = blk[:, :, :64].contiguous().view(torch.float16).float()                       # [r, nb, 32]
        ch = blk[:, :, 64:].reshape(r, w // 64, 4, 2, 8)                                    # chunk g: (lo | q(hi)) or (q(hi) | lo)
        lo = ch[:, :, :, 1 if pl.weight else 0].contiguous().view(torch.float8_e5m2).float().reshape(r, w // 64, 32)
        out = (hi + lo / 2048.0).reshape(r, w // 2)
    elif pl.fmt == FMT_F3:
        v = pl.p.contiguous().view(torch.float16).float().view(r, w // 64, 2, 32)
        out = (v[:, :, 0] + v[:, :, 1]).reshape(r, w // 2)
    else:
        v = (pl.p.to(torch.int32) << 16).view(torch.float32).view(r, w // 64, 2, 32)
        out = (v[:, :, 0] + v[:, :, 1]).reshape(r, w // 2)
    return out[:, :(pl.k if cols is None else cols)]


def split_planes(w2d, kpad=None, out=None, fmt=FMT_B3, weight=False):
    """fp32 [N, K] (device) -> Planes with K zero-padded to a multiple of 32.  fmt FMT_H8: `weight` selects the chunk order of a
    GEMM W operand (q(hi) | lo) instead of an activation's (lo | q(hi))."""
    p, n, k, ld = _mat(w2d, "weight")
    if out is None:
        kpad = kpad or (pad64(k) if fmt == FMT_H8C else pad32(k))
        if fmt == FMT_H8C:
            if kpad % 64:
                raise RuntimeError("mmsa.split_planes: h8c planes need kpad % 64 == 0")
            out = Planes(torch.zeros((n + 1) // 2, 3 * kpad, dtype=torch.int16, device=w2d.device), n, k, kpad, fmt, False)
        else:
            out = Planes(torch.empty(n, 2 * kpad, dtype=torch.int16, device=w2d.device), n, k, kpad, fmt, weight)
    kpad = out.kpad
    if out.fmt == FMT_H8C:
        if out.n != n or not out.p.is_contiguous() or out.p.shape[1] != 3 * kpad:
            raise RuntimeError("mmsa.split_planes: dense h8c planes of the source's row count expected")
        kind = 3
    else:
        kind = 0 if out.fmt == FMT_B3 else 4 if out.fmt == FMT_F3 else (2 if out.weight else 1)
    lib.call("mmsa_split_planes", p, ld, n, k, kpad, out.p.data_ptr(), kind, _clamp_ptr(), _stream())
    return out


def gemm(a, w, out=None, bias=None, act="none", alpha=1.0, colscale=None, resid=None, beta=1.0, resid_mod=0,
         batch=1, stride_a=0, stride_w=0, stride_bias=0, stride_r=0, stride_c=0, m=None, pixel_shuffle=None,
         out_planes=None, stride_cp=0, rowstats_out=None, row_norm=None):
    """out / out_planes = beta*resid + colscale*alpha*act(a @ w^T + bias).
    a: fp32 2-D view or activation Planes; w: weight Planes; out: fp32 view and/or out_planes: Planes.
    LayerNorm fold (include/mmsa.h, mmsa_gemm_split3): `rowstats_out` [M, N/64, 2] fp32 -- also write per-row strip sums of the stored
    values; `row_norm` = (mean_rstd [M, 2], colsum [N]) -- out_planes = colscale*alpha*act(rstd_r * (a @ w^T - mean_r * colsum) + bias)."""
    fmt = w.fmt
    if isinstance(a, Planes):
        pap, ma, ka, lda = a.mat("A")
        pa = None
        if ka < w.kpad:
            raise RuntimeError(f"mmsa.gemm: A planes have {ka} columns but the packed weight expects K={w.kpad}")
        if a.fmt != w.fmt or a.weight or (w.fmt == FMT_H8 and not w.weight) or (w.fmt == FMT_H8C and w.weight):
            raise RuntimeError(f"mmsa.gemm: operand formats differ (A fmt {a.fmt}, W fmt {w.fmt} weight={w.weight})")
    elif fmt not in (FMT_B3, FMT_F3):
        raise RuntimeError("mmsa.gemm: h8 weights need A as h8 planes")
    else:
        pa, ma, ka, lda = _mat(a, "A")
        pap = None
        if ka < w.kpad and lda < w.kpad:
            raise RuntimeError(f"mmsa.gemm: A has {ka} columns (ld {lda}) but the packed weight expects K={w.kpad}")
    m = ma if m is None else m
    pc, ldc = None, 0
    if out is not None:
        pc, _, _, ldc = _mat(out, "C")
    pcp, ldcp = None, 0
    if out_planes is not None:
        pcp, _, _, ldcp = out_planes.mat("Cplanes")
    pr, ldr = None, 0
    if resid is not None:
        pr, _, _, ldr = _mat(resid, "resid")
    ps = pixel_shuffle or (0, 0, 0)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = _event(), _event()
        lib.call("mmsa_event_record", e0, _stream())
    lib.call("mmsa_gemm_split3", pa, pap, lda, stride_a, w.p.data_ptr(), stride_w,
             _chk(bias, name="bias"), stride_bias, _chk(colscale, name="colscale"), pr, ldr, stride_r, resid_mod, beta,
             pc, ldc, stride_c, pcp, ldcp, stride_cp, m, w.n, w.kpad, batch, ACT[act], alpha,
             1 if pixel_shuffle else 0, ps[0], ps[1], ps[2], fmt, cp_format(out_planes), GEMM_MAX_GRID,
             _chk(rowstats_out), _chk(row_norm[0]) if row_norm else None, _chk(row_norm[1]) if row_norm else None, GEMM_FLAVOUR, _clamp_ptr(), _stream())
    if prof is not None:
        lib.call("mmsa_event_record", e1, _stream())
        ob = 3.0 if fmt == FMT_H8C else 4.0          # bytes per operand element (h8c planes: 3)
        pb = 0.0 if out_planes is None else (3.0 if out_planes.fmt == FMT_H8C else 4.0)
        prof.append((2.0 * m * w.n * w.k * batch, e0, e1,
                     batch * (ob * (m * w.kpad + w.n * w.kpad) + m * w.n * (pb + 4.0 * ((1 if out is not None else 0) + (1 if resid is not None else 0))))))
        if GEMM_SHAPES is not None:
            GEMM_SHAPES.append((m, w.n, w.k, batch, act, resid is not None, ("C" if out is not None else "") + ("P" if out_planes is not None else ""),
                                "planes" if pap is not None else "fp32"))
    return out if out is not None else out_planes


def convnext_mlp_fused(a, w1, w2, b1, b2, gamma, x, m, batch=1, stride_a=0, stride_w1=0, stride_w2=0, stride_x=0):
    """x[b] <- x[b] + gamma[b] * (gelu(a[b] @ w1[b].T + b1[b]) @ w2[b].T + b2[b]) in place (one kernel; C = 96).
    a: bf16 hi/lo or f3 activation Planes; w1 [4C, C], w2 [C, 4C]: weight Planes of the same format (first batch; batch strides in uint16 units)."""
    pa, _, _, lda = a.mat("A")
    px, _, c, ldx = _mat(x, "x")
    if a.fmt not in (FMT_B3, FMT_F3) or w1.fmt != a.fmt or w2.fmt != a.fmt:
        raise RuntimeError("mmsa.convnext_mlp_fused: bf16 hi/lo or f3 planes expected, the same format for A, W1 and W2")
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = _event(), _event()
        lib.call("mmsa_event_record", e0, _stream())
    lib.call("mmsa_convnext_mlp_fused", pa, lda, stride_a, w1.p.data_ptr(), stride_w1, w2.p.data_ptr(), stride_w2, _chk(b1), _chk(b2),
             _chk(gamma), px, ldx, stride_x, m, c, batch, GEMM_MAX_GRID, a.fmt, _clamp_ptr(), _stream())
    if prof is not None:      # both contractions of the pair count towards the GEMM family (bench.py roofline)
        lib.call("mmsa_event_record", e1, _stream())
        prof.append((2.0 * 2.0 * m * c * 4 * c * batch, e0, e1, 4.0 * batch * (m * c * 3 + 2 * 4 * c * c)))
        if GEMM_SHAPES is not None:
            GEMM_SHAPES.append((m, c, 4 * c, batch, "gelu", True, "C", "fused-pair"))
    return x


def convnext_mlp_fused_supported(c):
    return c == 96


def zero_(t):
    """Zero-fill a contiguous GPU tensor on the current stream (mmsa_zero_bytes: no framework kernel inside a captured step)."""
    if not t.is_contiguous():
        raise RuntimeError("mmsa.ops.zero_: contiguous tensor expected")
    if t.numel():
        lib.call("mmsa_zero_bytes", _chk(t, t.dtype), t.numel() * t.element_size(), _stream())
    return t


def rowstats_finalize(rowstats, rows, d, eps, out):
    """Strip sums of a GEMM's `rowstats_out` -> (mean, rstd) per row, [rows, 2] fp32 (LayerNorm statistics over d = 64 * strips columns)."""
    lib.call("mmsa_rowstats_finalize", _chk(rowstats), rows, d // 64, d, eps, _chk(out), _stream())
    return out


def layernorm(x, w, b, eps, out=None, out2=None, patchify=None, out_planes=None, group_rows=0, w_gstride=0, y_gcol=0, y_wrap=False):
    """Row LayerNorm.  group_rows > 0: stacked row groups with their own weights (w, b of shape [groups, C]); group g writes
    at column offset g * y_gcol, and at row (row % group_rows) when y_wrap."""
    px, rows, c, ldx = _mat(x, "x")
    py, ldy = None, 0
    if out is not None:
        py, _, _, ldy = _mat(out, "y")
    p2, ld2 = None, 0
    if out2 is not None:
        p2, _, _, ld2 = _mat(out2, "y2")
    pp, ldp = None, 0
    if out_planes is not None:
        pp, _, _, ldp = out_planes.mat("y planes")
    mh, mw = patchify or (0, 0)
    lib.call("mmsa_layernorm_rows", px, ldx, _chk(w), _chk(b), eps, py, ldy, p2, ld2, pp, ldp, rows, c,
             1 if patchify else 0, mh, mw, group_rows, w_gstride, y_gcol, 1 if y_wrap else 0,
             out_planes.fmt if out_planes is not None else FMT_B3, _clamp_ptr(), _stream())
    return out if out is not None else out_planes


MSDA_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}   # MMSA_DT_* (include/mmsa.h)


def _msda_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, extra=()):
    dt = value.dtype
    if dt not in MSDA_DTYPES:
        raise RuntimeError(f"ms_deform_attn: dtype {dt} not supported (float32, float16, float64)")   # the reference's dispatch set
    for t, n in ((value, "value"), (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight")) + tuple(extra):
        if not t.is_contiguous():
            raise RuntimeError(f"{n} tensor has to be contiguous")  # ms_deform_attn_cuda.cu:28-32
        if t.dtype != dt:
            raise RuntimeError(f"{n} must have the dtype of value ({dt}), got {t.dtype}")
    if not (spatial_shapes.is_contiguous() and level_start_index.is_contiguous()):
        raise RuntimeError("spatial_shapes / level_start_index tensor has to be contiguous")
    return dt


def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    """Drop-in for MultiScaleDeformableAttention.ms_deform_attn_forward (ops/src/vision.cpp:14); float32 / float16 / float64."""
    dt = _msda_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    n, s, m, d = value.shape
    _, lq, _, l, p, _ = sampling_loc.shape
    out = torch.empty(n, lq, m * d, dtype=dt, device=value.device)
    lib.call("mmsa_ms_deform_attn_forward", _chk(value, dt), _chk(spatial_shapes, torch.int64), _chk(level_start_index, torch.int64),
             _chk(sampling_loc, dt), _chk(attn_weight, dt), out.data_ptr(), n, s, m, d, l, lq, p, im2col_step, MSDA_DTYPES[dt], _stream())
    return out


def msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step=64):
    """Drop-in for MultiScaleDeformableAttention.ms_deform_attn_backward (ops/src/vision.cpp:15) ->
    [grad_value, grad_sampling_loc, grad_attn_weight]."""
    dt = _msda_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, extra=((grad_output, "grad_output"),))
    n, s, m, d = value.shape
    _, lq, _, l, p, _ = sampling_loc.shape
    if tuple(grad_output.shape) != (n, lq, m * d):
        raise RuntimeError(f"grad_output must be [{n}, {lq}, {m * d}], got {tuple(grad_output.shape)}")
    gv, gl, ga = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
    lib.call("mmsa_ms_deform_attn_backward", _chk(value, dt), _chk(spatial_shapes, torch.int64), _chk(level_start_index, torch.int64),
             _chk(sampling_loc, dt), _chk(attn_weight, dt), _chk(grad_output, dt), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
             n, s, m, d, l, lq, p, im2col_step, MSDA_DTYPES[dt], _stream())
    return [gv, gl, ga]


def msda_fused(value2d, spatial_shapes, level_start_index, raw, ref_points, out, batch, spatial, heads, d, levels, lq, points,
               out_planes=None, lo_bytes=False):
    """Fused MSDeformAttn middle part (include/mmsa.h).  value2d: fp32 [batch * spatial, heads * d] -- or H8 activation Planes of that matrix (the gather on fp16
    values, mmsa_msda_fused_planes; `lo_bytes`: add the planes' e5m2 lo bytes)."""
    pr, _, _, ldraw = _mat(raw, "raw")
    if isinstance(value2d, Planes):
        if value2d.fmt != FMT_H8 or value2d.weight or value2d.split:
            raise RuntimeError("mmsa.msda_fused: value planes must be H8 activation planes")
        pvp, _, _, ldvp = value2d.mat("value planes")
        po, ldo = None, 0
        if out is not None:
            po, _, _, ldo = _mat(out, "out")
        pp, ldp = None, 0
        if out_planes is not None:
            pp, _, _, ldp = out_planes.mat("out planes")
        lib.call("mmsa_msda_fused_planes", pvp, ldvp, 1 if lo_bytes else 0, _chk(spatial_shapes, torch.int64), _chk(level_start_index, torch.int64), pr, ldraw,
                 _chk(ref_points), po, ldo, pp, ldp, out_planes.fmt if out_planes is not None else FMT_B3,
                 batch, spatial, heads, d, levels, lq, points, _clamp_ptr(), _stream())
        return out if out is not None else out_planes
    pv, _, _, _ = _mat(value2d, "value")
    po, ldo = None, 0
    if out is not None:
        po, _, _, ldo = _mat(out, "out")
    pp, ldp = None, 0
    if out_planes is not None:
        pp, _, _, ldp = out_planes.mat("out planes")
    lib.call("mmsa_msda_fused", pv, _chk(spatial_shapes, torch.int64), _chk(level_start_index, torch.int64), pr, ldraw,
             _chk(ref_points), po, ldo, pp, ldp, out_planes.fmt if out_planes is not None else FMT_B3,
             batch, spatial, heads, d, levels, lq, points, _clamp_ptr(), _stream())
    return out if out is not None else out_planes


def relpos_bias(qkv, rh, rw, rp, b, h, w, heads, hd, ws):
    if isinstance(qkv, Planes):
        pq, _, _, ldq = qkv.mat("qkv")
        lib.call("mmsa_relpos_bias_planes", pq, ldq, _chk(rh), _chk(rw), _chk(rp), b, h, w, heads, hd, ws, _stream())
    else:
        pq, _, _, ldq = _mat(qkv, "qkv")
        lib.call("mmsa_relpos_bias", pq, ldq, _chk(rh), _chk(rw), _chk(rp), b, h, w, heads, hd, ws, _stream())
    return rp


def _v_fmt(qkv, qkv_bias, d, fused=False, rel=None):
    """v_fmt argument of the attention kernels from the planes' own description (include/mmsa.h): 0 = plain f3 (fp16 hi/lo) planes;
    1 = the v columns (from 2*D on) of BOTH the qkv planes and the bias planes are h8-encoded (Planes.split; the entry with a rel-pos
    prepass); 2 = qkv, bias and rel-pos planes are h8 planes throughout (the fused rel-pos entries: every contraction on fp16)."""
    if qkv.fmt == FMT_H8 or qkv_bias.fmt == FMT_H8 or (rel is not None and rel.fmt == FMT_H8):
        if not (fused and qkv.fmt == FMT_H8 and qkv_bias.fmt == FMT_H8 and rel is not None and rel.fmt == FMT_H8
                and not (qkv.weight or qkv_bias.weight or rel.weight)):
            raise RuntimeError("mmsa attention: h8 planes need the fused rel-pos entry with qkv, bias AND rel-pos planes in the h8 (activation) format")
        return 2
    if qkv.split != qkv_bias.split or qkv.split not in (0, 2 * d) or (fused and qkv.split):
        raise RuntimeError(f"mmsa attention: qkv planes (split {qkv.split}) and bias planes (split {qkv_bias.split}) must both be f3 (fp16 hi/lo) "
                           f"planes, either plain or -- for the entry with a rel-pos prepass -- with the v columns from {2 * d} on as h8 planes")
    if qkv.fmt != FMT_F3 or qkv_bias.fmt != FMT_F3 or (rel is not None and rel.fmt != FMT_F3):
        raise RuntimeError("mmsa attention: the hi/lo operand form of the attention kernels reads f3 planes (fp16 hi/lo pairs: ops.FMT_F3) since round 4 -- "
                           "qkv, bias and rel-pos planes; bf16 hi/lo planes have the same layout and would be misread")
    return 1 if qkv.split else 0


def split_planes_qkv(x2d, d):
    """fp32 [N, 3*d] (q | k | v) -> qkv Planes with q, k as fp16 hi/lo (f3) planes and v as h8 planes (Planes.split = 2*d): the layout the
    qkv GEMM writes for the attention kernels' fp16 P V.  Also for the [1, 3*d] bias row (pad tokens: k = v = bias)."""
    if (2 * d) % 32 or d % 32:
        raise RuntimeError("mmsa.split_planes_qkv: embed dim must be a multiple of 32")
    qk = split_planes(x2d[:, :2 * d].contiguous(), fmt=FMT_F3)
    v = split_planes(x2d[:, 2 * d:].contiguous(), fmt=FMT_H8)
    return Planes(torch.cat([qk.p, v.p], 1).contiguous(), x2d.shape[0], 3 * d, 3 * d, FMT_F3, split=2 * d)


def _guard(g):
    """Pointer of an attention logit-guard word (a 1-element fp32 device tensor, include/mmsa.h "Attention logit guard") or None."""
    if g is None:
        return None
    if g.dtype != torch.float32 or g.numel() != 1 or not g.is_cuda:
        raise RuntimeError("mmsa attention: max_logit must be a 1-element fp32 device tensor")
    return g.data_ptr()


def attention(qkv, qkv_bias, rp, out, b, h, w, heads, hd, ws, scale, max_logit=None):
    if isinstance(qkv, Planes):
        pq, _, _, ldq = qkv.mat("qkv")
        po, _, _, ldo = out.mat("out")
        lib.call("mmsa_attention_planes", pq, ldq, _chk(qkv_bias.p, torch.int16), _chk(rp),
                 po, ldo, b, h, w, heads, hd, ws, scale, out.fmt, _v_fmt(qkv, qkv_bias, heads * hd), _guard(max_logit), _stream())
    else:
        pq, _, _, ldq = _mat(qkv, "qkv")
        po, _, _, ldo = _mat(out, "out")
        lib.call("mmsa_attention", pq, ldq, _chk(qkv_bias), _chk(rp), po, ldo, b, h, w, heads, hd, ws, scale, _stream())
    return out


def window_relpos_planes(rel_pos_h, rel_pos_w, ws, fmt=FMT_B3):
    """Pack a windowed block's rel-pos tables for window_attention: [64, hd] fp32 -> Planes; rows 0..2ws-2 = rel_pos_h,
    rows 32.. = rel_pos_w (tables must already have 2*ws-1 rows: get_rel_pos's resize, IE:568-575, is the caller's)."""
    L = 2 * ws - 1
    if rel_pos_h.shape[0] != L or rel_pos_w.shape[0] != L or L > 27:
        raise RuntimeError(f"mmsa.window_relpos_planes: tables must have {L} <= 27 rows")
    m = torch.zeros(64, rel_pos_h.shape[1], dtype=torch.float32, device=rel_pos_h.device)
    m[:L] = rel_pos_h
    m[32:32 + L] = rel_pos_w
    return split_planes(m, fmt=fmt)


def global_relpos_planes(rel_pos_h, rel_pos_w, fmt=FMT_B3):
    """Pack a global block's rel-pos tables (already 2H-1 / 2W-1 rows) for global_attention: [256, hd] -> Planes (fmt FMT_H8 with h8
    qkv planes: the all-fp16 form of the kernel)."""
    if rel_pos_h.shape[0] > 127 or rel_pos_w.shape[0] > 127:
        raise RuntimeError("mmsa.global_relpos_planes: tables must have at most 127 rows")
    m = torch.zeros(256, rel_pos_h.shape[1], dtype=torch.float32, device=rel_pos_h.device)
    m[:rel_pos_h.shape[0]] = rel_pos_h
    m[128:128 + rel_pos_w.shape[0]] = rel_pos_w
    return split_planes(m, fmt=fmt)


def global_attention(qkv, qkv_bias, relg, out, b, h, w, heads, hd, scale, max_logit=None):
    """Global attention with the rel-pos terms fused (planes in, planes out); W = 64, H <= 64, head_dim 64."""
    pq, _, _, ldq = qkv.mat("qkv")
    po, _, _, ldo = out.mat("out")
    lib.call("mmsa_global_attention_planes", pq, ldq, _chk(qkv_bias.p, torch.int16), _chk(relg.p, torch.int16), po, ldo,
             b, h, w, heads, hd, scale, out.fmt, _v_fmt(qkv, qkv_bias, heads * hd, fused=True, rel=relg), _guard(max_logit), _stream())
    return out


_SELECTORS = {}


def window_selector(ws, device, f16=False):
    """[208, 32] 0/1 matrix of mmsa_window_attention_planes: row j selects key row j // ws and key column 14 + j % ws; 1.0 in bf16, or
    in fp16 for the all-fp16 form of the kernel (v_fmt = 2)."""
    key = (ws, str(device), f16)
    if key not in _SELECTORS:
        sel = torch.zeros(208, 32, dtype=torch.int16)
        j = torch.arange(min(ws * ws, 208))
        one = 0x3C00     # fp16 1.0: every form of the kernel runs fp16 MFMAs (hi/lo pairs or hi only) since round 4
        sel[j, j // ws] = one
        sel[j, 14 + j % ws] = one
        _SELECTORS[key] = sel.to(device)
    return _SELECTORS[key]


def window_attention(qkv, qkv_bias, relp, out, b, h, w, heads, hd, ws, scale, max_logit=None):
    """Windowed attention with the rel-pos bias fused (planes in, planes out); head_dim 64, ws <= 14."""
    if not 1 <= ws <= 14:
        raise RuntimeError(f"mmsa.window_attention: window_size {ws} not supported (1..14)")
    pq, _, _, ldq = qkv.mat("qkv")
    po, _, _, ldo = out.mat("out")
    vf = _v_fmt(qkv, qkv_bias, heads * hd, fused=True, rel=relp)
    lib.call("mmsa_window_attention_planes", pq, ldq, _chk(qkv_bias.p, torch.int16), _chk(relp.p, torch.int16),
             _chk(window_selector(ws, qkv.p.device, f16=vf == 2), torch.int16), po, ldo, b, h, w, heads, hd, ws, scale, out.fmt,
             vf, _guard(max_logit), _stream())
    return out


def colstats(x, stride_b, b, hw, out, wrow=None, out_is_zero=False):
    """out_is_zero: the caller has zeroed `out` on the current stream (one memset for several accumulator buffers)."""
    px, _, c, ldx = _mat(x, "x")
    lib.call("mmsa_colstats", px, ldx, stride_b, _chk(wrow), b, hw, c, _chk(out, torch.float64), int(out_is_zero), _stream())
    return out


def ffrm_finalize(stats, b, hw, c, mean_w, mean_b, wc, gn_w, gn_b, mean_o, rstd_o, mult_o, scratch):
    lib.call("mmsa_ffrm_finalize", _chk(stats, torch.float64), b, hw, c, mean_w, mean_b, _chk(wc), _chk(gn_w), _chk(gn_b),
             _chk(mean_o), _chk(rstd_o), _chk(mult_o), _chk(scratch), _stream())


def lnhw_apply(x, mean, rstd, mult, w, bias, out, b, hw):
    px, _, c, ldx = _mat(x, "x")
    po, _, _, ldo = _mat(out, "y")
    lib.call("mmsa_lnhw_apply", px, ldx, _chk(mean), _chk(rstd), _chk(mult), _chk(w), _chk(bias), po, ldo, b, hw, c, _stream())
    return out


def dwconv(x, w, bias, out, b, h, wd, k, act="none", xstride_b=None, ystride_b=None, out_planes=None, pstride_b=None,
           imgs_per_group=0, rowstats_out=None):
    """Depthwise conv; result to fp32 `out` and/or interleaved `out_planes`.  rowstats_out (7 x 7 only): fp32 [b*h*wd, 2*(C/64)] strip
    sums of the output for a row-normalising consumer GEMM (the ConvNeXt LayerNorm fold)."""
    px, _, c, ldx = _mat(x, "x")
    po, ldo = None, 0
    if out is not None:
        po, _, _, ldo = _mat(out, "y")
    pp, ldp = None, 0
    if out_planes is not None:
        pp, _, _, ldp = out_planes.mat("y planes")
    xs = h * wd * ldx if xstride_b is None else xstride_b
    ys = h * wd * ldo if ystride_b is None else ystride_b
    ps = h * wd * ldp if pstride_b is None else pstride_b
    lib.call("mmsa_dwconv_nhwc", px, ldx, xs, _chk(w), _chk(bias), po, ldo, ys, pp, ldp, ps, out_planes.fmt if out_planes is not None else FMT_B3,
             b, h, wd, c, k, ACT[act], imgs_per_group, _chk(rowstats_out), _clamp_ptr(), _stream())
    return out if out is not None else out_planes


def gconv(x, w, bias, out, b, h, wd, groups, cin_g, cout_g, k, act="none"):
    px, _, _, ldx = _mat(x, "x")
    po, _, _, ldo = _mat(out, "y")
    lib.call("mmsa_gconv_nhwc", px, ldx, _chk(w), _chk(bias), po, ldo, b, h, wd, groups, cin_g, cout_g, k, ACT[act], _stream())
    return out


def dwpair_gate(x, w_tap, out, b, h, wd, c, out_planes=None):
    """Neck Mlp middle (AM:127-132): gelu(dw3x3(x)[:, :C]) * dw3x3(x)[:, C:]; weights tap-major [9, C, 2, 2]."""
    px, _, _, ldx = _mat(x, "x")
    po, ldo = (None, 0)
    if out is not None:
        po, _, _, ldo = _mat(out, "y")
    pp, ldp = (None, 0)
    if out_planes is not None:
        pp, _, _, ldp = out_planes.mat("y planes")
    lib.call("mmsa_dwpair_gate", px, ldx, _chk(w_tap), po, ldo, pp, ldp, b, h, wd, c, _stream())
    return out if out is not None else out_planes


def im2col_nchw(x, c0, cin, p, out):
    b, ctot, h, w = x.shape
    if not x.is_contiguous():
        raise RuntimeError("mmsa.im2col_nchw: input image must be contiguous NCHW")
    po, _, kpad, ld = _mat(out, "out")
    if ld != kpad:
        raise RuntimeError("mmsa.im2col_nchw: output must be dense")
    lib.call("mmsa_im2col_nchw", _chk(x), b, ctot, c0, cin, h, w, p, po, kpad, _stream())
    return out


def gram_tn_scratch_bytes(b, p, c):
    return int(lib.raw.mmsa_gram_tn_scratch_bytes(b, p, c))


def gram_tn(x, y, stride_b, g, b, p, nblk=1, scratch=None):
    """G[b] = X[b]^T Y[b] -> float64 [B*c, c]: fp32 MFMA per 256-row slice, the slices summed in double in slice order (run-to-run
    identical, no zeroing of G needed).  scratch: a float32 / uint8 buffer of >= gram_tn_scratch_bytes(b, p, c) bytes (allocated here
    when None -- the backbone passes a workspace buffer)."""
    px, _, c, ldx = _mat(x, "X")
    py, _, _, ldy = _mat(y, "Y")
    need = gram_tn_scratch_bytes(b, p, c)
    if scratch is None:
        scratch = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    if not scratch.is_cuda or not scratch.is_contiguous():
        raise RuntimeError("gram_tn: scratch must be a contiguous device tensor")
    lib.call("mmsa_gram_tn", px, ldx, py, ldy, stride_b, _chk(g, torch.float64, "G"), b, p, c, nblk, scratch.data_ptr(),
             scratch.numel() * scratch.element_size(), _stream())
    return g


def chanattn_build(g, sq, sq_stride, sk, sk_stride, temp, wp, planes, b, c, heads):
    lib.call("mmsa_chanattn_build", _chk(g, torch.float64, "G"), sq, sq_stride, sk, sk_stride, _chk(temp), _chk(wp),
             planes.p.data_ptr(), b, c, planes.kpad, heads, _stream())


def gffm_build(e, px_, py_, b, c):
    lib.call("mmsa_gffm_build", _chk(e, torch.float64, "E"), px_.p.data_ptr(), py_.p.data_ptr(), b, c, px_.kpad, _stream())


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


def ca_apply(z, att, out, b, h, w, out_planes=None):
    pz, _, c, ldz = _mat(z, "z")
    pa, _, _, lda = _mat(att, "att")
    po, ldo = (None, 0)
    if out is not None:
        po, _, _, ldo = _mat(out, "out")
    pp, ldp = (None, 0)
    if out_planes is not None:
        pp, _, _, ldp = out_planes.mat("out planes")
    lib.call("mmsa_ca_apply", pz, ldz, pa, lda, po, ldo, pp, ldp, b, h, w, c, _stream())
    return out if out is not None else out_planes


def tail_fuse(cmap, cstride_b, xtok, bn_scale, bn_shift, out, b, hc, wc, hx, wx, out_planes=None):
    """f = BatchNorm_eval(cmap + bilinear_resize(xtok)) as NCHW; xtok None = no ViT feature added (add_vit_feature=False, BK:326)."""
    pc, _, c, ldc = _mat(cmap, "cmap")
    px, ldx = None, 0
    if xtok is not None:
        px, _, _, ldx = _mat(xtok, "xtok")
    pp, ldp = (out_planes.p.data_ptr(), 2 * out_planes.kpad) if out_planes is not None else (None, 0)
    lib.call("mmsa_tail_fuse", pc, ldc, cstride_b, px, ldx, _chk(bn_scale), _chk(bn_shift), _chk(out), pp, ldp, b, hc, wc, hx, wx, c, _stream())
    return out
