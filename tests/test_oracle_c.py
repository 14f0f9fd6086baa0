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
