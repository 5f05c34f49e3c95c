"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def bits_equal(a, b):
    """Bit-for-bit equality (NaN == NaN, +0 != -0)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    return a.tobytes() == b.tobytes()


def assert_bits(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert a.dtype == b.dtype, f"{what}: dtype {a.dtype} vs {b.dtype}"
    if a.tobytes() != b.tobytes():
        ai = a.view(f"u{a.dtype.itemsize}") if a.dtype.kind == "f" else a
        bi = b.view(f"u{b.dtype.itemsize}") if b.dtype.kind == "f" else b
        bad = np.argwhere(ai != bi)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {bad.shape[0]} mismatching elements, first at {i}: {a[i]!r} vs {b[i]!r}")


def econ_kwargs(g):
    return dict(
        max_shares=int(g["max_shares"]),
        starting_balance=float(g["starting_balance"]),
        per_share_commission=float(g["commission"]),
        initial_margin_requirement=float(g["imr"]),
        maintenance_margin_requirement=float(g["mmr"]),
    )
