"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerance of the fp32 parity claim: max-abs error relative to max|reference|.
# Measured fp32-vs-fp64 error of one reference forward is 1-2e-6; the bar is 1e-4 per forward.
FWD_RTOL = 1e-4


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_err(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(a.reshape(ref.shape) - ref).max() / max(np.abs(ref).max(), 1e-30))


def assert_close(a, ref, tol, what=""):
    e = rel_err(a, ref)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e
