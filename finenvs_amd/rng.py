"""Host mirror of the device redraw generator (Philox4x32-10, csrc/fe_env.hip:philox_u32).

Used only to pick the eval env's first day in redraw="device" mode so that the
whole day sequence is a pure function of (seed, draw counter)."""

_M0, _M1 = 0xD2511F53, 0xCD9E8D57
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF


def philox_u32(seed: int, counter: int) -> int:
    c = [counter & _MASK, (counter >> 32) & _MASK, 0x46454E56, 0]
    k = [seed & _MASK, (seed >> 32) & _MASK]
    for _ in range(10):
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & _MASK, p1 & _MASK, ((p0 >> 32) ^ c[3] ^ k[1]) & _MASK, p0 & _MASK]
        k = [(k[0] + _W0) & _MASK, (k[1] + _W1) & _MASK]
    return c[0]


def redraw_day(seed: int, counter: int, num_days: int) -> int:
    return (philox_u32(seed, counter) * num_days) >> 32
