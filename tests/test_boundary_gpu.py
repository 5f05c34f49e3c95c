"""GPU tests of the drop-in boundary's ownership and device rules (round-2 additions):

* the default env returns FRESH observation tensors, so the reference's PPO buffer -- which keeps a
  VIEW of the first stored `states` until its next torch.cat (finenvs/agents/PPO/buffer.py:45, 53-56)
  -- stores the right states;
* an env whose device_id is not the current device launches on its own device (TSE:28, 45);
* fe_env_create(logret = NULL) computes the log-return table on the device.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import assert_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    import finenvs_amd

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return finenvs_amd


@pytest.fixture(scope="module")
def fo():
    from oracle import fe_oracle

    fe_oracle.build()
    return fe_oracle


def t2n(t):
    return t.detach().cpu().numpy()


def _tables(fo, num_days, A, bars, W, seed=1234, drop=0.0):
    from finenvs_amd.data import synthetic

    prices, day_id, _ = synthetic.synthetic_series(num_days, A, bars, seed, drop)
    P, LR, *_ = fo.tables_from_series(prices, day_id, W)
    return P, LR


class _RefBufferStates:
    """The `states` half of the reference's PPO Buffer, restated (buffer.py:33-56): the first store keeps
    `states.unsqueeze(1)` -- a VIEW of the caller's tensor -- and only the next store copies it (torch.cat)."""

    def __init__(self):
        self.container = None

    def store(self, states: torch.Tensor) -> None:
        t = states.unsqueeze(1)
        self.container = t if self.container is None else torch.cat([self.container, t], dim=1)

    def clear(self) -> None:
        self.container = None


@pytest.mark.parametrize("obs_buffers", [0, 3])
def test_reference_buffer_retention_pattern(fe, fo, obs_buffers):
    """reset -> [agent.step; env.step; agent.store(states, ...); states = next_states] x T, then clear, twice
    (examples/time_series/PPO_LSTM_training_SPY.py:22-30).  Every stored state must equal the oracle's
    observation of that step.  obs_buffers=0 (the default) is the reference's fresh-tensor semantics; a ring
    is safe for this loop only from 3 buffers up (with 2 the first state of every batch is overwritten before
    the buffer's first torch.cat copies it -- the reason the ring is opt-in)."""
    P, LR = _tables(fo, 6, 1, 40, 8)
    N, W, T = 33, 8, 5
    ref = fo.OracleEnv(P, LR, W, num_envs=N, evaluate=True)
    env = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=W, num_envs=N, evaluate=True, obs_buffers=obs_buffers)
    assert fe.TimeSeriesEnv.__init__.__kwdefaults__["obs_buffers"] == 0
    buf = _RefBufferStates()
    g = torch.Generator().manual_seed(4)
    states = env.reset()
    want = [ref.reset().copy()]
    for batch in range(2):
        for t in range(T):
            a = (torch.rand((N, 1), generator=g) * 2 - 1).float()
            next_states, _, _, _ = env.step(a.to(env.device))
            ref.step(a.numpy())
            want.append(ref.obs.copy())
            buf.store(states)
            states = next_states
        stored = t2n(buf.container)  # (N, T, W, 5)
        for t in range(T):
            assert_bits(stored[:, t], want[batch * T + t], f"batch {batch}: stored state {t}")
        buf.clear()


def test_two_buffer_ring_would_corrupt_the_first_stored_state(fe, fo):
    """Documents WHY the ring is opt-in: with obs_buffers=2 the buffer's view of state 0 is overwritten by step 2."""
    P, LR = _tables(fo, 6, 1, 40, 8)
    env = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=8, num_envs=9, evaluate=True, obs_buffers=2)
    buf = _RefBufferStates()
    a = torch.full((9, 1), 0.7, device=env.device)
    s0 = env.reset()
    keep0 = s0.clone()
    s1, *_ = env.step(a)
    buf.store(s0)          # a view of ring buffer 0
    s2, *_ = env.step(a)   # writes ring buffer 0 again
    buf.store(s1)
    assert s2.data_ptr() == s0.data_ptr()
    assert not torch.equal(buf.container[:, 0], keep0)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs")
def test_env_on_a_device_that_is_not_current(fe, fo):
    """device_id=1 while cuda:0 is current: construction, reset, step, rollouts and the trajectory kernels all run
    on cuda:1 and equal the same env built on cuda:0; the caller's current device is left alone."""
    from finenvs_amd.rollout import FusedLinearRollout
    from finenvs_amd.trajectory import TrajectoryBuffer

    P, LR = _tables(fo, 6, 2, 40, 8)
    N, A, W, T = 300, 2, 8, 12
    torch.cuda.set_device(0)
    envs = [fe.TimeSeriesEnv(tables=(P, LR), num_intervals=W, num_envs=N, redraw="device", seed=3, device_id=d)
            for d in (0, 1)]
    assert envs[1].device == "cuda:1" and envs[1]._lib.fe_env_device(envs[1]._handle) == 1
    assert torch.cuda.current_device() == 0
    trajs = [TrajectoryBuffer(T, N, A, device=e.device) for e in envs]
    g = torch.Generator().manual_seed(8)
    o = [e.reset() for e in envs]
    assert torch.equal(o[0].cpu(), o[1].cpu())
    for t in range(T):
        a = (torch.rand((N, A), generator=g) * 2 - 1).float()
        outs = []
        for e, tr in zip(envs, trajs):
            sa, sr, sd = tr.next_slot()
            sa.copy_(a.to(e.device))
            outs.append(e.step(sa, rewards_out=sr, dones_out=sd))
        assert outs[1][0].device == torch.device("cuda:1")
        for x, y in zip(outs[0][:3], outs[1][:3]):
            assert torch.equal(x.cpu(), y.cpu()), f"step {t}"
        assert torch.cuda.current_device() == 0
    vals = torch.zeros((T, N))
    r0 = trajs[0].returns_and_advantages(vals.to("cuda:0"), torch.zeros(N, device="cuda:0"))
    r1 = trajs[1].returns_and_advantages(vals.to("cuda:1"), torch.zeros(N, device="cuda:1"))
    assert torch.equal(r0[0].cpu(), r1[0].cpu())
    w = torch.randn((W, 5), dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    rolls = [FusedLinearRollout(e, w, 0.1) for e in envs]
    res = [r.run(6) for r in rolls]
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x.cpu(), y.cpu())
    assert torch.cuda.current_device() == 0


def test_fe_env_create_computes_the_log_return_table_when_passed_null(fe, fo):
    """`logret = NULL -> computed`.  Equal to the table built from the whole series (a18) except the
    open-over-previous-close entry of each day's row 0, whose previous close lies outside the slice (documented
    in include/finenvs_amd.h): there the library applies the series-row-0 rule, 100*ln(O/O) = 0."""
    from finenvs_amd import _lib

    lib = _lib.load()
    A, W, N = 2, 8, 50
    P, LR = _tables(fo, 6, A, 40, W, drop=0.1)
    D, L, _ = P.shape
    dP = torch.from_numpy(P).cuda()
    cfg = _lib.FeConfig(N, D, L, W, A, 5, 0, 1e4, 0.01, 1.5, 0.25, 0, 0, 0, -1)  # training mode, no eval env
    h = C.c_void_p()
    _lib.check(lib.fe_env_create(C.byref(cfg), dP.data_ptr(), None, C.byref(h)))
    assert lib.fe_env_logret(h) not in (None, 0, dP.data_ptr())
    got = torch.empty((D, L, 4 * A), dtype=torch.float64, device="cuda")
    _lib.check(lib.fe_build_logret_tables(dP.data_ptr(), got.data_ptr(), D, L, A, None))  # the same kernel, caller-owned output
    torch.cuda.synchronize()
    got = t2n(got)
    want = LR.copy()
    first_open = np.zeros_like(want, dtype=bool)
    first_open[:, 0, 0::4] = True
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got[~first_open], want[~first_open], rtol=1e-13, atol=1e-17)
    assert np.all(got[first_open] == 0.0)
    # and the env built on it steps like the oracle fed the same table
    idx = torch.arange(N, dtype=torch.int64, device="cuda") % D
    z = lambda dt, n=N * A: torch.zeros((n,), dtype=dt, device="cuda")  # noqa: E731
    spot, cash, lng, sht, mar = z(torch.int64, N), torch.full((N * A,), 1e4, device="cuda"), z(torch.float32), z(torch.float32), z(torch.float64)
    ctr = z(torch.int64, 2)
    _lib.check(lib.fe_env_bind_state(h, idx.data_ptr(), spot.data_ptr(), cash.data_ptr(), lng.data_ptr(), sht.data_ptr(),
                                     mar.data_ptr(), None, None, ctr.data_ptr()))
    ref = fo.OracleEnv(P, got, W, num_envs=N, evaluate=False, eval_env=-1)
    obs = torch.empty((N, W, 5 * A), dtype=torch.float64, device="cuda")
    rew, done = z(torch.float64, N), z(torch.int32, N)
    g = torch.Generator().manual_seed(6)
    for t in range(60):
        a = (torch.rand((N, A), generator=g) * 2 - 1).float()
        da = a.cuda()
        _lib.check(lib.fe_env_step(h, da.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), None))
        torch.cuda.synchronize()
        o_r, r_r, d_r, _ = ref.step(a.numpy())
        assert_bits(t2n(obs), o_r, f"step {t} obs")
        assert_bits(t2n(rew), r_r, f"step {t} rewards")
        assert_bits(t2n(done), d_r, f"step {t} dones")
    assert lib.fe_env_destroy(h) == 0


def test_ring_audition_changes_placement_only(fe, fo):
    """obs_audition (ring mode): extra candidate buffers are tried at construction and the fastest kept.  The ring
    keeps its size, the record says what was measured, and every value equals the oracle / an un-auditioned env."""
    P, LR = _tables(fo, 6, 2, 40, 8)
    N, A, W = 700, 2, 8
    env = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=W, num_envs=N, evaluate=True, obs_buffers=2, obs_audition=4)
    rec = env.obs_audition
    assert len(env._obs_ring) == 2 and rec["candidates"] == 6 and len(rec["us"]) == 6 and len(rec["kept"]) == 2
    assert all(t > 0 for t in rec["us"]) and rec["kept"] == sorted(rec["kept"])
    ref = fo.OracleEnv(P, LR, W, num_envs=N, evaluate=True)
    assert_bits(t2n(env.reset()), ref.reset(), "reset obs")
    g = torch.Generator().manual_seed(3)
    for t in range(50):
        a = (torch.rand((N, A), generator=g) * 2 - 1).float()
        o, r, d, _ = env.step(a.to(env.device))
        o2, r2, d2, _ = ref.step(a.numpy())
        assert_bits(t2n(o), o2, f"step {t} obs"); assert_bits(t2n(r), r2, f"step {t} rewards"); assert_bits(t2n(d), d2, f"step {t} dones")
    # the audition is bounded in bytes as well as in candidates, and a candidate must beat a ring member by min_gain
    nbytes = N * W * 5 * A * 8
    before = [t.data_ptr() for t in env._obs_ring]
    env.audition_ring(extra=8, budget_bytes=3 * nbytes + 100, min_gain=1.0)  # nothing is 100 % faster: the ring stays
    assert env.obs_audition["candidates"] == 2 + 3 and env.obs_audition["kept"] == [0, 1]
    assert [t.data_ptr() for t in env._obs_ring] == before
    env.audition_ring(extra=8, budget_bytes=0)
    assert env.obs_audition["candidates"] == 2 and env.obs_audition["us"] == []
    plain = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=W, num_envs=N, evaluate=True, obs_audition=4)  # fresh-tensor mode: ignored
    assert not hasattr(plain, "obs_audition") and plain._obs_ring == []


def test_single_asset_launch_geometry_rule(fe, fo):
    """A single-asset tile is a whole number of phase-2 workgroup iterations (512 f64 / 1024 f32 tuples) and the grid is
    capped at 4 (f64) / 6 (f32) workgroups per CU -- DESIGN.md section 5; set_launch overrides and restores it."""
    from finenvs_amd.data import synthetic

    prices, day_id, _ = synthetic.synthetic_series(8, 1, 200, 3)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for W, dt, unit, per_cu in ((64, torch.float64, 8, 4), (64, torch.float32, 16, 6), (32, torch.float64, 16, 4), (100, torch.float64, 128, 4)):
        env = fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=65536, redraw="device", obs_dtype=dt)
        info = env.launch_info()
        assert info["tile_envs"] % unit == 0, (W, dt, info)
        assert info["grid"] <= cus * per_cu and info["grid"] % 8 == 0, (W, dt, info)
        over = env.set_launch(16, 512)
        assert over["tile_envs"] == 16 and over["grid"] == 512
        assert env.set_launch() == info  # 0 = automatic again
    small = fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=64, num_envs=100, redraw="device")
    assert small.launch_info()["grid"] == (100 + small.launch_info()["tile_envs"] - 1) // small.launch_info()["tile_envs"]


def test_two_envs_on_two_streams_do_not_interfere(fe, fo):
    """The C ABI takes the stream per call and copies its parameter block per launch: two env objects stepped from two
    HIP streams, interleaved without synchronisation in between, give exactly what each gives alone."""
    from finenvs_amd.data import synthetic

    prices, day_id, _ = synthetic.synthetic_series(6, 2, 60, 3, 0.05)
    mk = lambda seed: fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=8, num_envs=5000, redraw="device", seed=seed)
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = [(torch.rand((5000, 2), generator=g, device="cuda") * 2 - 1).float() for _ in range(12)]
    alone = []
    for seed in (1, 2):
        env = mk(seed)
        env.reset()
        out = [env.step(a) for a in acts]
        alone.append(([o[1].clone() for o in out], out[-1][0].clone(), env.cash.clone()))
    envs, streams = [mk(1), mk(2)], [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    rews = [[], []]
    last = [None, None]
    for e, s in zip(envs, streams):
        with torch.cuda.stream(s):
            e.reset()
    for a in acts:  # interleaved, no synchronisation between the two streams
        for i, (e, s) in enumerate(zip(envs, streams)):
            with torch.cuda.stream(s):
                obs, r, d, _ = e.step(a)
                rews[i].append(r)
                last[i] = obs
    torch.cuda.synchronize()
    for i in range(2):
        for k in range(len(acts)):
            assert torch.equal(rews[i][k], alone[i][0][k]), f"env {i} step {k} rewards"
        assert torch.equal(last[i], alone[i][1]) and torch.equal(envs[i].cash, alone[i][2])


def test_action_dtypes_cast_promote_or_refuse(fe, fo):
    """float64 actions take the reference's dtype promotion (TSE:298-302, 353-374; pinned by rollout_f64_actions.npz in
    tests/test_hip_parity.py) and here, for seeded multi-asset inputs, equal the oracle's promoted arithmetic bit for bit;
    cast_actions=True opts into a cast to f32 instead (then equal to the f32 call); other dtypes are refused."""
    P, LR = _tables(fo, 5, 1, 40, 8)
    a64 = (torch.rand((12, 1), generator=torch.Generator().manual_seed(3), dtype=torch.float64) * 2 - 1).cuda()
    env = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=8, num_envs=12, evaluate=True)
    with pytest.raises(ValueError, match="float32"):
        env.step(a64.half())
    o32, r32, d32, _ = env.step(a64.float())  # the refused call changed nothing
    env2 = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=8, num_envs=12, evaluate=True, cast_actions=True)
    o, r, d, _ = env2.step(a64)
    assert not env2.shares_promoted
    assert_bits(t2n(o), t2n(o32), "obs")
    assert_bits(t2n(r), t2n(r32), "rewards")
    assert_bits(t2n(d), t2n(d32), "dones")
    # promoted arithmetic vs the oracle, f64 / f32 steps mixed, small balance: 3 assets (tile loop: sleeve sum, shared done);
    # one asset with f64 observations (the software pipeline), with f32 observations (the tile loop at A = 1) and with an odd
    # observation size (8-byte stores)
    for A, N, W, f32 in ((3, 257, 8, False), (1, 3000, 8, False), (1, 3000, 8, True), (1, 700, 7, False)):
        P, LR = _tables(fo, 6, A, 40, W)
        env = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=W, num_envs=N, evaluate=True, starting_balance=900,
                               obs_dtype=torch.float32 if f32 else torch.float64)
        ref = fo.OracleEnv(P, LR, W, num_envs=N, evaluate=True, starting_balance=900, obs_f32=f32)
        g = torch.Generator().manual_seed(5 + A)
        for t in range(90):
            a = torch.rand((N, A), generator=g, dtype=torch.float64) * 2 - 1
            a = a if (t < 30 or t >= 60) else a.float()
            o, r, d, _ = env.step(a.cuda())
            o2, r2, d2, _ = ref.step(a.numpy())
            what = f"A={A} W={W} f32={f32} step {t}"
            assert_bits(t2n(o), o2, what + " obs"); assert_bits(t2n(r), r2, what + " rewards"); assert_bits(t2n(d), d2, what + " dones")
            assert_bits(t2n(env.cash), ref.cash, what + " cash")
        assert env.shares_promoted and ref.shares_f64
    P, LR = _tables(fo, 6, 3, 40, 8)
    N, A = 257, 3
    # the fused rollouts run the f32 arithmetic only: they refuse a promoted env
    from finenvs_amd.rollout import FusedLinearRollout

    env3 = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=8, num_envs=N, redraw="device")
    roll = FusedLinearRollout(env3, torch.ones((8, 5), dtype=torch.float64) * 0.1, 0.0)
    roll.run(1)
    env3.step(torch.zeros((N, A), dtype=torch.float64, device=env3.device))
    roll.sync_from_env()
    with pytest.raises(RuntimeError, match="float64 actions"):
        roll.run(1)


@pytest.mark.parametrize("N,A,W,dt", [(1003, 1, 16, torch.float64), (4099, 1, 64, torch.float32), (77, 3, 8, torch.float64)])
def test_step_notify_equals_step_and_reports_the_eval_env_early(fe, fo, N, A, W, dt):
    """fe_env_step_notify (the default mode's per-step host read, TSE:510): same results as fe_env_step bit for bit --
    only the tile order differs --, and the host flag carries (seq << 1) | done of the evaluation env."""
    import ctypes as C

    from finenvs_amd import _lib
    from finenvs_amd.data import synthetic

    prices, day_id, _ = synthetic.synthetic_series(8, A, 30, 21, 0.05)
    mk = lambda: fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=5, obs_dtype=dt)
    a_env, b_env = mk(), mk()
    lib = a_env._lib
    flag = C.c_void_p()
    _lib.check(lib.fe_host_flag_create(C.byref(flag)))
    word = C.c_uint64.from_address(flag.value)
    assert word.value == 0
    g = torch.Generator(device="cuda").manual_seed(2)
    obs = torch.empty((N, W, 5 * A), dtype=dt, device="cuda")
    rew = torch.empty((N,), dtype=torch.float64, device="cuda")
    done = torch.empty((N,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    seen_done = 0
    for k in range(1, 45):
        a = (torch.rand((N, A), generator=g, device="cuda") * 2 - 1).float()
        _lib.check(lib.fe_env_step_notify(a_env._handle, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), flag, 1000 + k, st))
        o2, r2, d2, _ = b_env.step(a)
        torch.cuda.synchronize()
        assert word.value >> 1 == 1000 + k
        assert (word.value & 1) == int(done[-1]), "the flag's low bit is the evaluation env's done flag"
        seen_done += word.value & 1
        assert torch.equal(obs, o2) and torch.equal(rew, r2) and torch.equal(done, d2)
        for x, y in ((a_env.cash, b_env.cash), (a_env.margin, b_env.margin), (a_env._spot0, b_env._spot0), (a_env.env_indices, b_env.env_indices)):
            assert torch.equal(x, y)
    assert seen_done >= 1  # 30-bar days: the evaluation env finished at least once (and redrew its day identically)
    # a training-mode shard without the evaluation env -> refused
    sh = fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=2 * N, rank=0, world_size=2, redraw="device", obs_dtype=dt)
    rc = lib.fe_env_step_notify(sh._handle, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), flag, 1, st)
    assert rc == _lib.FE_ERR_ARG and b"evaluation env" in lib.fe_last_error()
    # evaluate mode: the LAST workgroup reports (seq << 32) | how many envs have terminated so far (TSE:531)
    ev = fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, evaluate=True, obs_dtype=dt)
    ev2 = fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, evaluate=True, obs_dtype=dt)
    for k in range(1, 36):
        a = (torch.rand((N, A), generator=g, device="cuda") * 2 - 1).float()
        _lib.check(lib.fe_env_step_notify(ev._handle, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), flag, 70 + k, st))
        torch.cuda.synchronize()
        assert word.value >> 32 == 70 + k and (word.value & 0xFFFFFFFF) == int(ev._counters[0])
        o2, r2, d2, info2 = ev2.step(a)  # the class's own evaluate-mode step (polls its own flag)
        assert torch.equal(obs, o2) and torch.equal(rew, r2) and torch.equal(done, d2)
        if "returns" in info2:  # all envs had terminated: the raw env's counter says the same, then clear it as step() did
            assert int(ev._counters[0]) == N and torch.equal(ev.episode_returns, info2["returns"])
            ev.reset_evaluation_metrics()
            break
    else:
        raise AssertionError("no evaluation episode set finished within 35 steps of 30-bar days")
    # with trajectory outputs and bound episode statistics: the full form + flag, again equal to the plain calls
    from finenvs_amd.stats import EpisodeStats

    sa, sb = EpisodeStats(a_env), EpisodeStats(b_env)
    src = torch.empty((N,), dtype=torch.int64, device="cuda")
    pos = torch.empty((N, A), dtype=torch.float64, device="cuda")
    aout = torch.empty((N, A), dtype=torch.float32, device="cuda")
    src2, pos2, aout2 = torch.empty_like(src), torch.empty_like(pos), torch.empty_like(aout)
    for k in range(100, 140):
        a = (torch.rand((N, A), generator=g, device="cuda") * 2 - 1).float()
        _lib.check(lib.fe_env_step_traj_notify(a_env._handle, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(),
                                               aout.data_ptr(), src.data_ptr(), pos.data_ptr(), flag, 5000 + k, st))
        o2, r2, d2, _ = b_env.step(a, descriptors_out=(src2, pos2), actions_out=aout2)
        torch.cuda.synchronize()
        assert word.value == ((5000 + k) << 1 | int(done[-1]))
        assert torch.equal(obs, o2) and torch.equal(rew, r2) and torch.equal(done, d2)
        assert torch.equal(src, src2) and torch.equal(pos, pos2) and torch.equal(aout, aout2) and torch.equal(aout, a)
        assert torch.equal(sa.running_returns, sb.running_returns)
    ra, rb = sa.read(), sb.read()
    assert ra["num_training_episodes"] == rb["num_training_episodes"] > 0
    assert ra["evaluation_return"] == rb["evaluation_return"] and ra["num_evaluation_episodes"] == rb["num_evaluation_episodes"]
    # the finished-episode sums are per-env partials added up in a fixed order (fe_env_stats_reduce): the notify form's
    # reversed tile walk gives the same bits
    assert ra["mean_training_return"] == rb["mean_training_return"]
    assert ra["std_dev_training_return"] == rb["std_dev_training_return"]
    sa.close()
    sb.close()
    torch.cuda.synchronize()
    _lib.check(lib.fe_host_flag_destroy(flag))


def test_default_mode_polls_the_host_flag_and_keeps_the_reference_rng_stream(fe, fo):
    """redraw="torch" (the class default): step() decides the evaluation env's redraw from the host flag; the draws it
    takes from torch's global generator are the ones the plain dones[-1].item() path takes (same seed -> same day
    sequence, same generator state afterwards)."""
    from finenvs_amd.data import synthetic
    from finenvs_amd.stats import EpisodeStats

    prices, day_id, _ = synthetic.synthetic_series(7, 1, 24, 8)
    runs = []
    for plain in (False, True):
        torch.manual_seed(99)
        env = fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=4, num_envs=9)
        assert env.redraw == "torch" and env._flag is not None
        stats = None
        if plain:  # the plain path: fe_env_step + the reference's dones[-1].item()
            env._lib.fe_host_flag_destroy(env._flag)
            env._flag = None
        g = torch.Generator(device="cuda").manual_seed(1)
        days = []
        for _ in range(120):
            env.step((torch.rand((9, 1), generator=g, device="cuda") * 2 - 1).float())
            days.append(int(env.env_indices[-1]))
        runs.append((days, torch.rand(3).tolist(), env.cash.clone()))
        if stats is not None:
            stats.close()
    assert runs[0][0] == runs[1][0] and len(set(runs[0][0])) > 1
    assert runs[0][1] == runs[1][1], "the global generator must be in the same state afterwards"
    assert torch.equal(runs[0][2], runs[1][2])


def test_default_mode_step_refuses_stream_capture_at_once(fe, fo):
    """The class-default step() reads a host flag every step (TSE:510; evaluate mode TSE:531).  Under a caller's own
    stream capture the launch does not run, so the flag would be polled for a kernel that is not executing: step()
    must raise immediately (the reference's `.item()` fails under capture just the same), not spin for its timeout and
    then synchronise mid-capture.  redraw="device" captures fine."""
    import time

    from finenvs_amd.data import synthetic

    prices, day_id, _ = synthetic.synthetic_series(7, 1, 24, 8)
    for kw in ({}, {"evaluate": True}):
        env = fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=4, num_envs=9, obs_buffers=1, **kw)
        assert env._flag is not None
        a = torch.zeros((9, 1), device=env.device)
        env.step(a)  # allocator / lazy init outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        t0 = time.monotonic()
        with pytest.raises(RuntimeError, match="cannot be captured"):
            with torch.cuda.graph(g):
                env.step(a)
        assert time.monotonic() - t0 < 5.0
        env.step(a)  # the env is still usable afterwards (nothing was launched, the sequence number did not move)
        torch.cuda.synchronize()
    env = fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=4, num_envs=9, obs_buffers=1, redraw="device")
    a = torch.zeros((9, 1), device=env.device)
    rew, done = torch.empty((9,), dtype=torch.float64, device=env.device), torch.empty((9,), dtype=torch.int32, device=env.device)
    env.step(a, rewards_out=rew, dones_out=done)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.step(a, rewards_out=rew, dones_out=done)
    before = env._spot0.clone()
    g.replay()
    torch.cuda.synchronize()
    assert not torch.equal(before, env._spot0)


def test_host_flag_wait_synchronises_once_and_rereads_before_it_gives_up(fe, fo):
    """A word that never arrives (here: the flag is polled for a sequence number no launch carries) ends in
    HostFlagTimeout after `flag_timeout_s`, having synchronised the stream; the env keeps working afterwards."""
    from finenvs_amd.data import synthetic
    from finenvs_amd.environments.time_series_env import HostFlagTimeout

    prices, day_id, _ = synthetic.synthetic_series(7, 1, 24, 8)
    env = fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=4, num_envs=9)
    a = torch.zeros((9, 1), device=env.device)
    env.step(a)
    env.flag_timeout_s = 0.2
    with pytest.raises(HostFlagTimeout, match="host flag"):
        env._eval_env_done(env._flag_seq + 12345)
    env.step(a)
    torch.cuda.synchronize()


def test_env_objects_can_be_driven_from_different_host_threads(fe, fo):
    """one env object is driven from one host thread at a time, but the C ABI is re-entrant ACROSS
    env objects (per-call parameter block, thread-local error text, a mutex around the launch-preparation cache).  Four
    threads, each with its own env, stream and fused MLP rollout (which goes through that cache), stepping
    concurrently: every thread gets what it gets alone."""
    import threading

    from finenvs_amd.data import synthetic
    from finenvs_amd.rollout import FusedMLPRollout

    prices, day_id, _ = synthetic.synthetic_series(6, 1, 40, 3)
    g = torch.Generator().manual_seed(0)
    W1, b1, W2 = torch.randn((40, 32), generator=g) * 0.5, torch.randn(32, generator=g) * 0.1, torch.randn(32, generator=g) * 0.3

    def work(seed, out, use_stream):
        env = fe.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=8, num_envs=700 + seed, redraw="device", seed=seed)
        ga = torch.Generator(device="cuda").manual_seed(seed)
        s = torch.cuda.Stream() if use_stream else torch.cuda.current_stream()
        with torch.cuda.stream(s):
            tot = torch.zeros((), dtype=torch.float64, device="cuda")
            for _ in range(150):
                a = (torch.rand((env.num_envs, 1), generator=ga, device="cuda") * 2 - 1).float()
                _, r, _, _ = env.step(a)
                tot += r.sum()
            roll = FusedMLPRollout(env, W1, b1, W2, 0.0, activation="relu")
            _, r2, _ = roll.run(20, record_actions=False)
            tot += r2.sum()
            s.synchronize()
        out[seed] = (float(tot), env.cash.clone())

    alone, together = {}, {}
    for seed in range(4):
        work(seed, alone, False)
    threads = [threading.Thread(target=work, args=(seed, together, True)) for seed in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
        assert not t.is_alive()
    for seed in range(4):
        assert together[seed][0] == alone[seed][0], seed
        assert torch.equal(together[seed][1], alone[seed][1]), seed


def test_store_policy_follows_how_the_buffers_are_used_not_what_is_stored(fe, fo):
    """A single-asset observation of 128 - 256 MiB is stored sc1 | nt when the caller alternates over buffers (a ring of two
    overflows the 256 MiB Infinity Cache) and plain sc1 when it rewrites the SAME buffer launch after launch (the cache
    absorbs it) -- decided per launch in launch_env from the previous launch's pointer (advisor, round 4).  The policy is a
    cache hint: an env that rewrites one buffer and an env that alternates over two produce identical bytes."""
    P, LR = _tables(fo, 6, 1, 100, 64)
    N, W = 65536, 64  # 168 MB per f64 observation: the size class the policy applies to
    one = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=W, num_envs=N, evaluate=True, obs_buffers=1)
    two = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=W, num_envs=N, evaluate=True, obs_buffers=2)
    g = torch.Generator(device="cuda:0").manual_seed(5)
    assert torch.equal(one.reset(), two.reset())
    for t in range(5):
        a = (torch.rand((N, 1), generator=g, device="cuda:0") * 2 - 1).float()
        o1, r1, d1, _ = one.step(a)
        o2, r2, d2, _ = two.step(a)
        assert o1.data_ptr() == one._obs_ring[0].data_ptr()
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), f"step {t}"
