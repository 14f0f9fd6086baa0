"""Shared helpers for the parity tests."""
import torch

# Parity gate of BASELINE.json's north_star: within 1e-3 relative of the reference CPU encoder.
# Fixed here as: per-tensor relative L2 error <= 1e-3 AND max-abs error / max-abs(reference) <= 1e-3.
REL_TOL = 1e-3


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def max_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def assert_close(a, b, tol=REL_TOL, what=""):
    assert tuple(a.shape) == tuple(b.shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    r, m = rel_l2(a, b), max_rel(a, b)
    assert r <= tol and m <= tol, f"{what}: rel_l2={r:.3e} max_rel={m:.3e} > {tol:g}"
    return r, m
