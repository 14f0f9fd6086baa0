"""CPU: the plain-C restatement of the reference's MSDA kernel (oracle/msda_ref.c) against the golden vectors."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "libmsda_ref.so"))


def _run(clib, fn, dt, value, shapes, lsi, loc, aw):
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.empty((N, Lq, M * D), dtype=dt)
    arrs = [np.ascontiguousarray(value, dt), np.ascontiguousarray(shapes, np.int64), np.ascontiguousarray(lsi, np.int64),
            np.ascontiguousarray(loc, dt), np.ascontiguousarray(aw, dt), out]
    getattr(clib, fn)(*[a.ctypes.data_as(ctypes.c_void_p) for a in arrs], N, S, M, D, L, Lq, P)
    return out


def test_reference_fixture(clib, golden_dir):
    g = np.load(os.path.join(golden_dir, "msda.npz"))
    a = [g[k] for k in ("t_value", "t_shapes", "t_lsi", "t_loc", "t_aw")]
    assert np.allclose(_run(clib, "msda_ref_f64", np.float64, *a), g["t_out64"])  # ops/test.py:26-50
    assert np.allclose(_run(clib, "msda_ref_f32", np.float32, *a), g["t_out"], rtol=1e-2, atol=1e-3)  # ops/test.py:53-75
    assert np.allclose(_run(clib, "msda_ref_f32", np.float32, *a), g["t_out"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("tag", ["inj", "ext"])
def test_border_samples(clib, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "msda.npz"))
    a = [g[f"{tag}_{k}"] for k in ("value", "shapes", "lsi", "loc", "aw")]
    assert np.allclose(_run(clib, "msda_ref_f32", np.float32, *a), g[f"{tag}_out"], rtol=1e-4, atol=1e-5)


def _run_bwd(clib, fn, dt, g, tag):
    value, loc, aw, gout = (np.ascontiguousarray(g[f"{tag}_{k}"], dt) for k in ("value", "loc", "aw", "gout"))
    shapes, lsi = np.ascontiguousarray(g[f"{tag}_shapes"], np.int64), np.ascontiguousarray(g[f"{tag}_lsi"], np.int64)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    gv, gl, ga = np.empty_like(value), np.empty_like(loc), np.empty_like(aw)
    getattr(clib, fn)(*[a.ctypes.data_as(ctypes.c_void_p) for a in (value, shapes, lsi, loc, aw, gout, gv, gl, ga)], N, S, M, D, L, Lq, P)
    return gv, gl, ga


def test_backward_reference_gradcheck_geometry_f64(clib, golden_dir):
    """The C restatement of ms_deform_attn_backward against autograd through the reference's own pytorch core, float64, on the
    geometry of the reference's gradient check (ops/test.py:16-20,77-90)."""
    g = np.load(os.path.join(golden_dir, "msda_bwd.npz"))
    for got, key in zip(_run_bwd(clib, "msda_ref_bwd_f64", np.float64, g, "t64"), ("gvalue", "gloc", "gaw")):
        assert np.allclose(got, g[f"t64_{key}"], rtol=1e-10, atol=1e-12), key


@pytest.mark.parametrize("tag", ["inj", "ext", "d40"])
def test_backward_border_samples_f32(clib, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "msda_bwd.npz"))
    for got, key in zip(_run_bwd(clib, "msda_ref_bwd_f32", np.float32, g, tag), ("gvalue", "gloc", "gaw")):
        ref = g[f"{tag}_{key}"]
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), key
