"""GPU: the C-ABI drop-ins of the reference's native extension (segmentation/ops/src/vision.cpp:13-16) beyond the fp32 forward of
test_ops_gpu.py: the reference's dtype dispatch (float64 / float16, ms_deform_attn_cuda.cu:64,134) and ms_deform_attn_backward,
against goldens computed by the reference's own ms_deform_attn_core_pytorch (forward) and autograd through it (backward)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ext():
    import mmsa.msda_ext as e
    return e


def _fwd_args(g, tag, dt):
    return [torch.from_numpy(g[f"{tag}_{k}"]).to(DEV) if k in ("shapes", "lsi") else torch.from_numpy(g[f"{tag}_{k}"]).to(DEV, dt)
            for k in ("value", "shapes", "lsi", "loc", "aw")]


def test_forward_float64_reference_known_answer(ext, golden_dir):
    """The reference's double-precision forward check (ops/test.py:26-50: torch.allclose at default tolerances) through the C ABI."""
    g = np.load(os.path.join(golden_dir, "msda.npz"))
    args = _fwd_args(g, "t", torch.float64)
    out = ext.ms_deform_attn_forward(*args, 2)
    assert out.dtype == torch.float64
    assert torch.allclose(out.cpu(), torch.from_numpy(g["t_out64"]))
    assert (out.cpu() - torch.from_numpy(g["t_out64"])).abs().max() < 1e-15


@pytest.mark.parametrize("tag", ["inj", "ext"])
def test_forward_float16_and_float64_border_cases(ext, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "msda.npz"))
    ref = torch.from_numpy(g[f"{tag}_out"])
    o64 = ext.ms_deform_attn_forward(*_fwd_args(g, tag, torch.float64), 64)
    assert (o64.cpu().float() - ref).abs().max() <= 2e-6 * ref.abs().max()
    o16 = ext.ms_deform_attn_forward(*_fwd_args(g, tag, torch.float16), 64)
    assert o16.dtype == torch.float16
    # half inputs: the operands are rounded to 11 bits; the reference's own fp32 bar (ops/test.py:53-75) is rtol 1e-2 / atol 1e-3
    assert torch.allclose(o16.cpu().float(), ref, rtol=1e-2, atol=1e-2 * ref.abs().max().item())
    with pytest.raises(RuntimeError):
        ext.ms_deform_attn_forward(*_fwd_args(g, tag, torch.bfloat16), 64)


def _bwd_case(g, tag, dt):
    a = {k: torch.from_numpy(g[f"{tag}_{k}"]) for k in ("value", "shapes", "lsi", "loc", "aw", "gout", "gvalue", "gloc", "gaw")}
    args = [a["value"].to(DEV, dt), a["shapes"].to(DEV), a["lsi"].to(DEV), a["loc"].to(DEV, dt), a["aw"].to(DEV, dt), a["gout"].to(DEV, dt)]
    return args, a


def test_backward_float64_gradcheck_geometry(ext, golden_dir):
    """ms_deform_attn_backward on the geometry of the reference's gradient check (ops/test.py:77-90), float64, against autograd
    through the reference's pytorch core."""
    g = np.load(os.path.join(golden_dir, "msda_bwd.npz"))
    args, a = _bwd_case(g, "t64", torch.float64)
    for got, key in zip(ext.ms_deform_attn_backward(*args, 2), ("gvalue", "gloc", "gaw")):
        assert got.dtype == torch.float64 and tuple(got.shape) == tuple(a[key].shape)
        assert torch.allclose(got.cpu(), a[key], rtol=1e-10, atol=1e-12), key


@pytest.mark.parametrize("tag", ["inj", "ext", "d40"])
def test_backward_float32_border_samples(ext, golden_dir, tag):
    """Border-crossing samples, three levels / one level, and a D = 40 head (not a power of two: the atomics path)."""
    g = np.load(os.path.join(golden_dir, "msda_bwd.npz"))
    args, a = _bwd_case(g, tag, torch.float32)
    outs = ext.ms_deform_attn_backward(*args, 64)
    again = ext.ms_deform_attn_backward(*args, 64)          # outputs are (re)zeroed by the call itself
    for got, got2, key in zip(outs, again, ("gvalue", "gloc", "gaw")):
        ref = a[key]
        assert (got.cpu() - ref).abs().max() <= 2e-5 * ref.abs().max(), key
        assert (got2.cpu() - ref).abs().max() <= 2e-5 * ref.abs().max(), key


def test_backward_float16(ext, golden_dir):
    """Half tensors: the kernel reads the fp16-rounded inputs, computes in fp32 and rounds each gradient once (grad_value: fp16
    atomics).  Expected values = autograd through the oracle's core on the SAME rounded inputs (rounding sampling_loc to 11 bits
    moves samples across pixel / border boundaries, where the location gradient is discontinuous: the fp32 golden of the
    unrounded inputs is not the right comparison for gloc)."""
    from oracle import ref_encoder as R  # checker only
    g = np.load(os.path.join(golden_dir, "msda_bwd.npz"))
    args, a = _bwd_case(g, "inj", torch.float16)
    value, shapes, lsi, loc, aw, gout = [t.cpu() for t in args]
    v32, l32, w32 = value.double().requires_grad_(True), loc.double().requires_grad_(True), aw.double().requires_grad_(True)
    R.msda_core(v32, shapes, l32, w32).backward(gout.double())
    for got, ref, key in zip(ext.ms_deform_attn_backward(*args, 64), (v32.grad, l32.grad, w32.grad), ("gvalue", "gloc", "gaw")):
        assert got.dtype == torch.float16
        ref = ref.float()
        assert (got.cpu().float() - ref).norm() <= 1e-2 * ref.norm(), key


def test_autograd_function_matches_reference_gradients(ext, golden_dir):
    """MSDeformAttnFunction (ms_deform_attn_func.py:19-50) end to end: forward value + the three gradients through autograd."""
    g = np.load(os.path.join(golden_dir, "msda_bwd.npz"))
    args, a = _bwd_case(g, "ext", torch.float32)
    value, shapes, lsi, loc, aw, gout = args
    value, loc, aw = value.requires_grad_(True), loc.requires_grad_(True), aw.requires_grad_(True)
    y = ext.MSDeformAttnFunction.apply(value, shapes, lsi, loc, aw, 64)
    y.backward(gout)
    for got, key in ((value.grad, "gvalue"), (loc.grad, "gloc"), (aw.grad, "gaw")):
        assert (got.cpu() - a[key]).abs().max() <= 2e-5 * a[key].abs().max(), key
    assert shapes.grad is None


def test_fused_msda_head_width_40(golden_dir):
    """The hot-path fused gather with D = 40 channels per head (ViT-H: 1280 * 0.5 / 16), 10 lanes per (query, head)."""
    from mmsa import ops
    g = np.load(os.path.join(golden_dir, "msda_bwd.npz"))
    value, shapes, lsi, loc, aw = _fwd_args(g, "d40", torch.float32)
    out = ops.msda_forward(value, shapes, lsi, loc, aw)
    from oracle import ref_encoder as R
    ref = R.msda_core(value.cpu(), shapes.cpu(), loc.cpu(), aw.cpu())
    assert (out.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()
