"""Minimal Box space.  The reference stores gym.spaces.Box objects (TSE:224-234) that
nothing on the hot path reads; gym is used when importable, else this stand-in."""
import numpy as np

try:  # pragma: no cover - gym is not installed in the build image
    from gym.spaces import Box  # type: ignore
except Exception:  # noqa: BLE001

    class Box:
        def __init__(self, low, high, dtype=np.float64):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
