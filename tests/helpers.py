"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def _canon(a):
    """Bytes of an array with every NaN mapped to one canonical NaN: a NaN's sign and payload are
    not part of the contract (x86 and gfx950 generate different default NaNs); everything else,
    including the sign of zero, is."""
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = a.copy()
        a[np.isnan(a)] = np.nan
    return a


def bits_equal(a, b):
    """Bit-for-bit equality (any NaN == any NaN, +0 != -0)."""
    a, b = _canon(a), _canon(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    return a.tobytes() == b.tobytes()


def assert_bits(a, b, what=""):
    a, b = _canon(a), _canon(b)
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
