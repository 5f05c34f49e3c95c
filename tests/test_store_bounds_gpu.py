"""GPU: the observation stores never leave the caller's buffer.

The step / reset / render kernels write the observation through raw buffer stores whose descriptor covers exactly the
packs of one wavefront iteration; tail lanes are dropped by the hardware range check (no per-lane predicate,
finenvs_amd/csrc/fe_step_kernel.h:stream_tile).  Here the observation is a window INSIDE a larger allocation with
sentinel bands on both sides -- an out-of-range store would land in the bands, not in unmapped memory -- for every
store width (16 / 8 / 4 bytes), both dtypes, single and multi asset, env counts that leave ragged last tiles and
ragged last iterations.  Run this file first after touching the store path.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 1 << 16  # elements on either side


def _cases():
    #  N,   A,  W, dtype            (W * 5 * A decides the pack width: % 4 -> 16 B for f32, % 2 -> 16 B for f64)
    yield 1000, 1, 64, torch.float64
    yield 1003, 1, 64, torch.float64     # ragged last tile
    yield 517, 1, 7, torch.float64       # odd env size: 8-byte packs
    yield 1000, 1, 64, torch.float32
    yield 999, 1, 30, torch.float32      # 150 elements: 8-byte packs
    yield 333, 1, 13, torch.float32      # 65 elements: 4-byte packs
    yield 301, 3, 8, torch.float64
    yield 301, 3, 9, torch.float64       # 135 elements: 8-byte packs
    yield 77, 30, 16, torch.float64
    yield 77, 30, 16, torch.float32
    yield 65, 7, 5, torch.float32        # 175 elements: 4-byte packs
    yield 1, 1, 3, torch.float64
    yield 9, 1, 390, torch.float64       # the reference's default window (TSE:19)
    yield 9, 1, 390, torch.float32       # 1950 elements: 8-byte packs


@pytest.mark.parametrize("N,A,W,dt", list(_cases()))
def test_observation_stores_stay_inside_the_buffer(N, A, W, dt):
    import finenvs_amd
    from finenvs_amd import _lib
    from finenvs_amd.data import synthetic

    prices, day_id, _ = synthetic.synthetic_series(5, A, 40 + W, 11)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=3,
                                    obs_dtype=dt)
    ref = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=3,
                                    obs_dtype=dt)
    n = N * W * 5 * A
    sentinel = -12345.5
    big = torch.full((GUARD + n + GUARD,), sentinel, dtype=dt, device="cuda")
    window = big[GUARD:GUARD + n]
    esz = big.element_size()
    assert (window.data_ptr() - big.data_ptr()) == GUARD * esz
    rew = torch.empty((N,), dtype=torch.float64, device="cuda")
    done = torch.empty((N,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(5)

    def guards_intact(what):
        assert bool((big[:GUARD] == sentinel).all()) and bool((big[GUARD + n:] == sentinel).all()), f"{what}: store outside the observation"

    _lib.check(env._lib.fe_env_reset_obs(env._handle, window.data_ptr(), st))
    torch.cuda.synchronize()
    guards_intact("reset")
    assert torch.equal(window.view(N, W, 5 * A), ref.reset())
    for k in range(3):
        a = (torch.rand((N, A), generator=g, device="cuda") * 2 - 1).float()
        window.fill_(sentinel)
        _lib.check(env._lib.fe_env_step(env._handle, a.data_ptr(), window.data_ptr(), rew.data_ptr(), done.data_ptr(), st))
        torch.cuda.synchronize()
        guards_intact(f"step {k}")
        o, r, d, _ = ref.step(a)
        assert torch.equal(window.view(N, W, 5 * A), o) and torch.equal(rew, r) and torch.equal(done, d)
    # render: any number of descriptors (here fewer than the env count, so the last tile is ragged in another place)
    src, pos = ref.describe()
    m = max(1, N - 5)
    window.fill_(sentinel)
    _lib.check(env._lib.fe_env_render_n(env._handle, src.data_ptr(), pos.data_ptr(), m, window.data_ptr(), st))
    torch.cuda.synchronize()
    guards_intact("render")
    assert bool((window[m * W * 5 * A:] == sentinel).all()), "render wrote past its last descriptor"
    assert torch.equal(window[: m * W * 5 * A].view(m, W, 5 * A), ref.reset()[:m])
