"""Pins the C oracle (oracle/fe_oracle.c) bit-for-bit against fixtures produced by
running the reference itself (oracle/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import fe_oracle as fo
from tests.helpers import assert_bits, bits_equal, econ_kwargs, load_golden

TABLE_CASES = ["tables_full.npz", "tables_ragged.npz", "tables_skip2.npz", "tables_oih.npz",
               # the reference's other two fixtures, at its default window (TSE:19) and the example's (W = 4)
               "tables_ibm.npz", "tables_spy.npz", "tables_ibm_w4.npz", "tables_spy_w4.npz"]


@pytest.mark.parametrize("name", TABLE_CASES)
def test_tables_match_reference(name):
    g = load_golden(name)
    W = int(g["W"])
    starts, stops, L = fo.bounds(g["series_day_id"], W)
    assert_bits(starts, g["ref_start_indices"], "start_indices")
    assert_bits(stops, g["ref_stop_indices"], "stop_indices")
    assert L == int(g["ref_max_length"])
    P = fo.build_tables(g["series_prices"], starts, stops, L)
    assert_bits(P, g["ref_price_environments"], "price_environments")
    # the log transform goes through libm's log vs torch's vectorised log: allow 2 ulp,
    # and require the NaN padding to be identical
    lr = fo.build_logret(g["series_prices"])
    ref = g["ref_log_return_dataset"]
    assert np.array_equal(np.isnan(lr), np.isnan(ref))
    np.testing.assert_allclose(lr, ref, rtol=5e-16, atol=1e-17)
    LR = fo.build_tables(ref, starts, stops, L)  # slicing/padding of the reference series: exact
    assert_bits(LR, g["ref_log_return_environments"], "log_return_environments")


def _replay(g, name, host_redraw=False, check_obs_full=True):
    W, N = int(g["W"]), int(g["N"])
    evaluate = bool(int(g["evaluate"]))
    if host_redraw:
        torch.manual_seed(int(g["torch_seed"]))
    env = fo.OracleEnv(g["prices"], g["logret"], W, evaluate=evaluate, env_indices=g["init_env_idx"],
                       **econ_kwargs(g))
    D = env.D
    if host_redraw:
        # the reference burns global-generator draws while building its NaN padding (TSE:207-210)
        for d in range(D):
            rem = int(np.isnan(g["prices"][d, :, 0]).sum())
            if rem > 0:
                torch.rand((rem, 4))
        first = int(torch.randint(0, D, (1,)))  # the constructor's draw, TSE:253-255
        if N == D + 1:  # the replicated fixtures overwrite env_indices after construction
            assert first == int(g["init_env_idx"][-1])
    if int(g.get("obs_reset_last_row_only", 0)):
        assert_bits(env.reset()[:, -1, :], g["obs_reset"], f"{name} reset obs (last row)")
    else:
        assert_bits(env.reset(), g["obs_reset"].reshape(N, W, -1), f"{name} reset obs")
    T = g["actions"].shape[0]
    returns = None
    for t in range(T):
        obs, rew, done, info = env.step(g["actions"][t])
        what = f"{name} step {t}"
        assert_bits(rew, g["rewards"][t], what + " rewards")
        assert_bits(done, g["dones"][t], what + " dones")
        if host_redraw and done[-1]:
            env.set_day(N - 1, int(torch.randint(0, D, (1,))))  # TSE:510-513
        assert_bits(env.cash.reshape(-1), g["cash"][t].reshape(-1), what + " cash")
        assert_bits(env.margin.reshape(-1), g["margin"][t].reshape(-1), what + " margin")
        assert_bits(env.long.reshape(-1), g["long"][t].reshape(-1), what + " long")
        assert_bits(env.short.reshape(-1), g["short"][t].reshape(-1), what + " short")
        assert_bits(env.spot0, g["spot0"][t] if g["spot0"][t].ndim == 1 else g["spot0"][t][:, 0], what + " spot0")
        if "env_idx" in g:
            assert_bits(env.env_idx, g["env_idx"][t], what + " env_idx")
        if check_obs_full:
            assert_bits(obs, g["obs"][t].reshape(obs.shape), what + " obs")
        else:
            assert_bits(obs[:, -1, :], g["obs"][t], what + " obs last row")
        if "returns" in info:
            returns = info["returns"]
    return returns


@pytest.mark.parametrize("name", ["rollout_train_native.npz", "rollout_train_n64.npz"])
def test_training_rollouts_with_torch_redraw(name):
    _replay(load_golden(name), name, host_redraw=True)


@pytest.mark.parametrize("name", ["rollout_eval.npz", "rollout_eval_ragged.npz"])
def test_evaluate_rollouts_emit_reference_returns(name):
    g = load_golden(name)
    returns = _replay(g, name)
    assert returns is not None
    assert_bits(returns, g["returns"], "episode returns")


@pytest.mark.parametrize("name", ["rollout_stress_60.npz", "rollout_stress_150.npz", "rollout_stress_400.npz",
                                  "rollout_stress_1500.npz", "rollout_econ.npz", "rollout_edge_actions.npz"])
def test_stress_rollouts(name):
    g = load_golden(name)
    _replay(g, name)


def test_stress_fixtures_reach_every_branch():
    """The fixtures must exercise illegal trades, margin calls and bankruptcy."""
    calls = bankrupt = illegal = 0
    for bal in (60, 150, 400, 1500):
        g = load_golden(f"rollout_stress_{bal}.npz")
        T = g["actions"].shape[0]
        # bankruptcy-only dones: done while the calendar says the episode goes on
        spot_prev = np.concatenate([g["init_spot0"][None], g["spot0"][:-1]])
        cal_done = (spot_prev + 1 + int(g["W"])) >= g["prices"].shape[1]
        bankrupt += int(((g["dones"] == 1) & ~cal_done).sum())
        pos_prev = np.concatenate([(g["init_long"] - g["init_short"])[None], (g["long"] - g["short"])[:-1]])
        want = np.clip(np.rint(g["actions"] * 5.5), -5, 5)
        got = (g["long"] - g["short"]) - pos_prev
        illegal += int(((want != 0) & (got == 0) & (g["dones"] == 0)).sum())
        calls += int((g["margin"][1:] > 1.5 * g["short"][1:] * 1e-9).sum())
        assert T == 120
    assert bankrupt > 10 and illegal > 100 and calls > 100


def test_real_data_oih_rollout():
    g = load_golden("rollout_oih.npz")
    _replay(g, "rollout_oih", host_redraw=True, check_obs_full=False)


@pytest.mark.parametrize("name", ["rollout_ibm_w390.npz", "rollout_spy_w390.npz", "rollout_ibm_w4.npz", "rollout_spy_w4.npz"])
def test_real_data_ibm_spy_rollouts(name):
    """The reference's own unit test builds IBM, OIH and SPY (tests/unit/test_time_series_env.py:10-14) and steps
    them 1000 times; these are 900-step training rollouts of the reference on IBM and SPY, at its default window
    (W = 390, TSE:19) and at the example's (W = 4), eval-env redraws included."""
    g = load_golden(name)
    assert g["dones"].sum() >= 4
    _replay(g, name, host_redraw=True, check_obs_full=False)


@pytest.mark.parametrize("name", ["rollout_sleeves3.npz", "rollout_sleeves30.npz", "rollout_sleeves3_f64.npz"])
def test_multi_asset_sleeves_equal_side_by_side_references(name):
    """The oracle's sleeve loop against A reference envs stepped side by side; rollout_sleeves30.npz = the 30-asset
    shape of BASELINE configs 3-5 (full observations at `obs_steps`, the newest window row at every step)."""
    g = load_golden(name)
    W, N, A = int(g["W"]), int(g["N"]), int(g["A"])
    env = fo.OracleEnv(g["prices"], g["logret"], W, evaluate=True, env_indices=g["init_env_idx"], **econ_kwargs(g))
    assert env.A == A
    assert_bits(env.reset(), g["obs_reset"], "reset obs")
    full_at = {int(t): i for i, t in enumerate(g["obs_steps"])} if "obs_steps" in g else None
    f64_steps = g.get("act_f64")  # rollout_sleeves3_f64.npz: float64 actions at these steps (the references promote, TSE:353-374)
    for t in range(g["actions"].shape[0]):
        a = g["actions"][t] if (f64_steps is None or f64_steps[t]) else g["actions"][t].astype(np.float32)
        obs, rew, done, _ = env.step(a)
        env.terminated[:] = 0  # the fixture cleared the metrics each step
        env.n_terminated[0] = 0
        what = f"sleeves step {t}"
        if full_at is None:
            assert_bits(obs, g["obs"][t], what + " obs")
        else:
            assert_bits(np.ascontiguousarray(obs[:, -1, :]), g["obs_last_row"][t], what + " newest window row")
            if t in full_at:
                assert_bits(obs, g["obs"][full_at[t]], what + " obs")
        assert_bits(rew, g["rewards"][t], what + " rewards")
        assert_bits(done, g["dones"][t], what + " dones")
        assert_bits(env.cash, g["cash"][t], what + " cash")
        assert_bits(env.margin, g["margin"][t], what + " margin")
        assert_bits(env.long.astype(g["long"].dtype), g["long"][t], what + " long")
        assert_bits(env.short.astype(g["short"].dtype), g["short"][t], what + " short")
    assert g["dones"].sum() > 0
    assert (f64_steps is None) or (env.shares_f64 and 0 < int(f64_steps.sum()) < len(f64_steps))


def test_share_change_rounding_probes():
    g = load_golden("rounding.npz")
    for ms, key in ((5, "share_changes_ms5"), (9, "share_changes_ms9")):
        # drive the oracle's a3 through a one-step env whose only visible effect is the position
        acts = g["actions"]
        n = acts.shape[0]
        P = np.full((1, 4, 4), 1.0)
        LR = np.zeros((1, 4, 4))
        env = fo.OracleEnv(P, LR, 2, num_envs=n, max_shares=ms, starting_balance=1e6, evaluate=True)
        env.step(acts)
        got = env.long.reshape(-1) - env.short.reshape(-1)
        want = g[key]
        assert np.array_equal(got, want + 0.0)


def test_f32_inplace_add_of_f64_rounds_once():
    """SURVEY 8(c) `rounding_*`: the `f32 += f64` single-rounding probe, as a known-answer vector from torch."""
    g = load_golden("rounding.npz")
    assert_bits(fo.f32_iadd_f64(g["f32_base"], g["f64_delta"]), g["f32_iadd_f64"], "f32 += f64")
    assert_bits(fo.f32_iadd_f64(g["f32_base"], g["f64_delta"], subtract=True), g["f32_isub_f64"], "f32 -= f64")
    assert g["f32_iadd_f64"][0] == np.float32(1.00000012) and g["f32_iadd_f64"][1] == np.float32(1.0)
    # double rounding (f64 sum -> f32 delta first) would get some of these wrong: the probe must be able to tell
    naive = (g["f32_base"] + g["f64_delta"].astype(np.float32)).astype(np.float32)
    assert (naive != g["f32_iadd_f64"]).any()


def test_agent_return_bookkeeping_matches_reference_agent():
    """agent_stats.npz was recorded from the reference's own PPOAgent.store / log_progress (PPO_agent.py:110-168)
    during a reference training rollout: running returns bit-exact after every step; at every log point the same
    evaluation return and episode count, mean / std of the finished-episode returns to f32 rounding (the
    reference reduces an f32 list, the restatement keeps f64 sums)."""
    g = load_golden("agent_stats.npz")
    N = int(g["N"])
    st = fo.EpisodeStatsOracle(N, N - 1)
    logs = {int(r[0]): r for r in g["logs"]}
    for t in range(g["rewards"].shape[0]):
        st.step(g["rewards"][t], g["dones"][t])
        assert_bits(st.running, g["running"][t], f"running returns after step {t}")
        if t in logs:
            _, ev, has_ev, n, mean, std, logged = logs[t]
            got = st.read(reset=False)
            assert (got["evaluation_return"] is not None) == bool(has_ev) == bool(logged)
            assert got["num_training_episodes"] == int(n)
            if has_ev:  # log_progress only reports (and clears) once an evaluation episode has finished
                assert np.float32(got["evaluation_return"]) == np.float32(ev)
                assert got["mean_training_return"] == pytest.approx(mean, rel=2e-6)
                assert got["std_dev_training_return"] == pytest.approx(std, rel=2e-6)
                st.read(reset=True)
    assert sum(int(r[6]) for r in g["logs"]) >= 3


def test_ppo_discounted_returns_match_reference_buffer():
    g = load_golden("ppo_returns.npz")
    out = fo.discounted_returns(g["rewards"], g["dones"], g["last_values"], float(g["gamma"]))
    assert_bits(out, g["returns"], "returns")
    adv = out - g["values"]
    assert_bits(adv.astype(np.float32), g["advantages"], "advantages")


def test_oracle_openmp_threads_do_not_change_results():
    g = load_golden("rollout_train_n64.npz")
    W = int(g["W"])
    a = fo.OracleEnv(g["prices"], g["logret"], W, evaluate=True, env_indices=g["init_env_idx"], nthreads=1)
    b = fo.OracleEnv(g["prices"], g["logret"], W, evaluate=True, env_indices=g["init_env_idx"], nthreads=4)
    for t in range(60):
        oa, ra, da, _ = a.step(g["actions"][t])
        ob, rb, db, _ = b.step(g["actions"][t])
        assert_bits(oa, ob); assert_bits(ra, rb); assert_bits(da, db)


def test_philox_known_answer():
    """The device redraw generator is Philox4x32-10 (Salmon et al., SC'11).  Pinned to the PUBLISHED known-answer
    vectors of Random123's kat_vectors file through a generic four-word implementation, to which the three
    restatements in this repo -- the oracle's C (fo_philox_u32), the host mirror (finenvs_amd/rng.py) and, through
    the oracle-vs-HIP rollouts, the kernel's (fe_device_common.h:philox_u32) -- are then tied: they are that function
    with counter words (ctr_lo, ctr_hi, 0x46454e56, 0) and key (seed_lo, seed_hi), first output word."""
    import ctypes as C

    from finenvs_amd.rng import philox_u32, redraw_day

    M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF

    def philox4x32_10(c, k):
        c, k = list(c), list(k)
        for _ in range(10):
            p0, p1 = M0 * c[0], M1 * c[2]
            c = [((p1 >> 32) ^ c[1] ^ k[0]) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k[1]) & MASK, p0 & MASK]
            k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
        return c

    # Random123 kat_vectors: "philox4x32 10 <ctr x4> <key x2> <expected x4>"
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([MASK] * 4, [MASK] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in kat:
        assert philox4x32_10(ctr, key) == want
    lib = fo.lib()
    for seed, ctr in [(0, 0), (1, 0), (0x123456789ABCDEF, 7), (42, 2**33 + 5), (2**64 - 1, 2**64 - 1)]:
        want = philox4x32_10([ctr & MASK, ctr >> 32, 0x46454E56, 0], [seed & MASK, seed >> 32])[0]
        assert lib.fo_philox_u32(C.c_uint64(seed), C.c_uint64(ctr)) == want
        assert philox_u32(seed, ctr) == want
    for i in range(50):
        assert lib.fo_redraw_day(C.c_uint64(9), C.c_uint64(i), C.c_int64(64)) == redraw_day(9, i, 64)
    days = [lib.fo_redraw_day(C.c_uint64(9), C.c_uint64(i), C.c_int64(64)) for i in range(2000)]
    assert min(days) == 0 and max(days) == 63


def test_oracle_policy_forms_are_consistent():
    """The two restated policy forms (window sum vs indicator table + position term) agree to rounding,
    and the table is NaN exactly where no window can start."""
    from finenvs_amd.data import synthetic

    prices, day_id, _ = synthetic.synthetic_series(5, 2, 40, 3, 0.1)
    W = 70 if False else 9
    P, LR, *_ = fo.tables_from_series(prices, day_id, W)
    D, L, _ = P.shape
    rng = np.random.default_rng(1)
    weights = rng.normal(0, 2.0, (W, 5))
    table, wsum = fo.policy_table(LR, weights, W)
    assert table.shape == (D, L, 2)
    assert np.isnan(table[:, L - W + 1:, :]).all() and np.isfinite(table[:, 0, :]).all()
    assert wsum == pytest.approx(weights[:, 4].sum(), rel=1e-14)
    env = fo.OracleEnv(P, LR, W, num_envs=11, evaluate=True)
    obs = env.reset().copy()
    for t in range(25):
        a_win = fo.policy_linear(obs, weights, 0.1)
        row = env.env_idx * L + env.spot0 if t == 0 else row_next
        a_tab = fo.policy_table_actions(table, wsum, 0.1, row, obs[:, 0, 4::5].copy())
        assert np.abs(a_win.astype(np.float64) - a_tab).max() < 1e-6
        idx_pre, spot_pre = env.env_idx.copy(), env.spot0.copy()
        obs, _, _, _ = env.step(a_win)
        obs = obs.copy()
        row_next = idx_pre * L + spot_pre + 1


def _replay_f64_actions(g, promoted: bool):
    """Replays rollout_f64_actions.npz through the oracle; promoted=False forces the all-f32 arithmetic (actions cast to
    f32), the thing every other fixture pins.  Returns the first (step, field) that differs from the reference, or None."""
    W, N = int(g["W"]), int(g["N"])
    env = fo.OracleEnv(g["prices"], g["logret"], W, num_envs=N, evaluate=True, **econ_kwargs(g))
    if not np.array_equal(env.reset(), g["obs_reset"]):
        return (-1, "obs_reset")
    for t in range(g["actions"].shape[0]):
        a = g["actions"][t] if (g["act_f64"][t] and promoted) else g["actions"][t].astype(np.float32)
        obs, rew, done, _ = env.step(a)
        for name, got, want in (("rewards", rew, g["rewards"][t]), ("dones", done, g["dones"][t]), ("cash", env.cash.reshape(-1), g["cash"][t]),
                                ("margin", env.margin.reshape(-1), g["margin"][t]),
                                ("long", env.long.reshape(-1).astype(np.float64), g["long"][t]),
                                ("short", env.short.reshape(-1).astype(np.float64), g["short"][t]),
                                ("spot0", env.spot0, g["spot0"][t]), ("obs_last_row", np.ascontiguousarray(obs[:, -1, :]), g["obs_last_row"][t])):
            if not bits_equal(np.asarray(got), np.asarray(want).astype(np.asarray(got).dtype)):  # (any NaN == any NaN)
                return (t, name)
    return None


@pytest.mark.parametrize("fname", ["rollout_f64_actions.npz", "rollout_f64_actions_econ.npz"])
def test_float64_actions_follow_the_references_dtype_promotion(fname):
    """rollout_f64_actions.npz (default economics) / rollout_f64_actions_econ.npz (imr 1.4, commission 0.035: neither exact in
    f32, so `imr * short_shares` -- an f64 product once short_shares is f64, TSE:376-379 -- differs from the f32 product): the reference stepped with float64 actions (steps 0-59), then float32 actions on the env
    whose share tensors that promoted to f64 (60-99), then float64 again.  The oracle's promoted arithmetic
    (fo_step_ex: commission products, short-entry commission and liquidation fee in f64; share change in the actions'
    dtype) reproduces every reward, done flag, state value and observation row bit for bit -- and the plain f32
    arithmetic does NOT (the fixture can tell the two apart), which is why the build no longer casts f64 actions."""
    g = load_golden(fname)
    assert int(g["act_f64"].sum()) == 90 and g["dones"].sum() > 100
    assert _replay_f64_actions(g, promoted=True) is None
    diff = _replay_f64_actions(g, promoted=False)
    assert diff is not None and diff[1] in ("rewards", "cash", "margin"), diff
    if fname.endswith("_econ.npz"):
        # the fixture separates the f64 `imr * short_shares` product from the f32 one: some margin the reference holds is
        # not what (double)((float)imr * short) * open gives for the same short count and bar
        imr, W = float(g["imr"]), int(g["W"])
        idx, last = g["env_idx"], g["spot0"] + W - 1
        found = False
        for t in range(1, 60):
            keep = (g["dones"][t] == 0) & (g["short"][t] > 0) & (g["short"][t] == g["short"][t - 1])
            if not keep.any():
                continue
            O = g["prices"][idx[t - 1][keep], last[t][keep], 0]
            sh = g["short"][t][keep]
            m32 = (np.float32(imr) * sh.astype(np.float32)).astype(np.float64) * O
            m64 = (imr * sh) * O
            found |= bool((m32 != m64).any())
        assert found
