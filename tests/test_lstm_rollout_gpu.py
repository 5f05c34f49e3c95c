"""GPU parity of the fused rollout's LSTM head (fe_env_rollout_lstm: the actor of the reference's own time-series
scripts, gate contractions on the matrix cores, recurrent weights in registers).

* the whole K-step rollout -- actions, rewards, dones, state -- equals the oracle's loop
  ``actions = fo.policy_lstm(obs); obs, r, d = step(actions)`` BIT FOR BIT: the v_mfma_f32_32x32x2_f32 accumulation
  is an fmaf chain whose order oracle/fe_oracle.c:fo_policy_lstm restates, and sigmoid / tanh are built from
  exactly-rounded operations only (rintf, fmaf, ldexpf, IEEE division), the same sequence on both sides;
* against torch's own ``nn.LSTM`` + ``nn.Linear`` + ``tanh`` in fp32 on the rendered observation (what
  finenvs/agents/networks/lstm.py:49-57 computes): 1e-5 absolute -- a different summation order and libm's
  exp/tanh; that is the tolerance north_star states for floating point.
"""
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


def _modules(H, seed, gain=6.0):
    """nn.LSTM(5, H) + nn.Linear(H, 1) with torch's default init, the input weights scaled up so that log-returns of
    ~1e-3..5e-2 move the gates and the actions spread over (-1, 1)."""
    torch.manual_seed(seed)
    lstm = torch.nn.LSTM(5, H, num_layers=1, batch_first=True)
    lin = torch.nn.Linear(H, 1)
    with torch.no_grad():
        lstm.weight_ih_l0[:, :4].mul_(gain * np.sqrt(H))
        lstm.weight_ih_l0[:, 4].mul_(4.0)
        lin.weight.mul_(6.0)
    return lstm, lin


def _packed(fo, lstm, lin):
    whh, wx = fo.lstm_pack(t2n(lstm.weight_ih_l0), t2n(lstm.weight_hh_l0), t2n(lstm.bias_ih_l0), t2n(lstm.bias_hh_l0))
    return whh, wx, t2n(lin.weight).reshape(-1).copy(), float(lin.bias.detach())


def _make(fe, fo, N, A, W, days, bars, drop, evaluate, seed):
    P, LR = _tables(fo, days, A, bars, W, seed=seed, drop=drop)
    D = P.shape[0]
    idx = (np.arange(N) * 7 + 1) % D  # 7 is coprime to every day count used here: all days are in play
    kw = dict(num_intervals=W, evaluate=evaluate, starting_balance=2000)
    ref = fo.OracleEnv(P, LR, env_indices=idx, redraw_mode=1, seed=9, auto_emit=False, **kw)
    ref.redraw_counter[0] = 1
    env = fe.TimeSeriesEnv(tables=(P, LR), env_indices=idx, redraw="device", seed=9, **kw)
    return ref, env


def test_lstm_activations_equal_oracle_bit_for_bit(fe, fo):
    """sigmoid / tanh of the LSTM head on 1.3 M inputs (dense sweep, normal draws, edge values): device == oracle
    bit for bit, and both within 1e-7 of the f64 functions."""
    from finenvs_amd import _lib

    rng = np.random.default_rng(0)
    x = np.concatenate([np.linspace(-100, 100, 1_000_001), rng.normal(0, 3, 300_000), rng.normal(0, 1e-3, 10_000),
                        [0.0, -0.0, 1e-8, -1e-8, 1e-38, 88.0, -88.0, 1e30, -1e30, np.inf, -np.inf]]).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    sig, tnh = torch.empty_like(xd), torch.empty_like(xd)
    lib = _lib.load()
    _lib.check(lib.fe_lstm_activations(xd.data_ptr(), sig.data_ptr(), tnh.data_ptr(), x.size, torch.cuda.current_stream().cuda_stream))
    s_ref, t_ref = fo.lstm_activations(x)
    assert_bits(t2n(sig), s_ref, "sigmoid")
    assert_bits(t2n(tnh), t_ref, "tanh")
    x64 = x.astype(np.float64)
    with np.errstate(over="ignore"):
        assert np.abs(s_ref - 1.0 / (1.0 + np.exp(-x64))).max() < 1e-7
    assert np.abs(t_ref - np.tanh(x64)).max() < 1e-7


@pytest.mark.parametrize("N,A,W,H,days,bars,drop,evaluate", [
    (300, 1, 4, 32, 6, 40, 0.0, False),       # the reference scripts' window (num_intervals=4)
    (500, 1, 4, 128, 5, 60, 0.05, False),     # PPOAgentLSTM's default hidden_dim = 128
    (77, 3, 7, 64, 4, 40, 0.0, False),        # odd W, several sleeves
    (21, 30, 4, 32, 5, 40, 0.1, True),        # DJIA-shaped sleeves, evaluate mode
    (40, 30, 5, 128, 5, 40, 0.0, False),      # 30 sleeves, 2 envs per 64-pair tile
    (131, 5, 1, 64, 6, 45, 0.1, False),       # W = 1: no recurrent step at all
    (260, 1, 16, 32, 5, 50, 0.0, False),      # partial last tile (260 = 2 x 128 + 4)
    (40, 1, 390, 32, 18, 30, 0.0, False),     # the reference's default window num_intervals=390 (TSE:19): 390 recurrent steps
])
def test_lstm_rollout_equals_oracle_loop_bit_for_bit(fe, fo, N, A, W, H, days, bars, drop, evaluate):
    from finenvs_amd.rollout import FusedLSTMRollout

    ref, env = _make(fe, fo, N, A, W, days, bars, drop, evaluate, seed=3 * N + W)
    lstm, lin = _modules(H, seed=W + H)
    whh, wx, wout, bout = _packed(fo, lstm, lin)
    roll = FusedLSTMRollout.from_modules(env, lstm, lin)
    assert_bits(t2n(roll.whh), whh, "packed Whh")
    assert_bits(t2n(roll.wx), wx, "packed Wx")
    obs = ref.reset().copy()
    assert_bits(t2n(roll.observation()), obs, "initial obs")
    K, reps = 5, 2 * (bars + 3) // 5 + 1
    seen = set()
    for rep in range(reps):
        acts, rews, dones = roll.run(K)
        for k in range(K):
            a_ref = fo.policy_lstm(obs, whh, wx, wout, bout)
            seen.update(np.unique(np.clip(np.rint(a_ref * 5.5), -5, 5)).tolist())
            obs, r_ref, d_ref, _ = ref.step(a_ref)
            obs = obs.copy()
            what = f"replay {rep} step {k}"
            assert_bits(t2n(acts[k]), a_ref, what + " actions")
            assert_bits(t2n(rews[k]), r_ref, what + " rewards")
            assert_bits(t2n(dones[k]), d_ref, what + " dones")
        assert_bits(t2n(env.cash), ref.cash, f"replay {rep} cash")
        assert_bits(t2n(env.margin), ref.margin, f"replay {rep} margin")
        assert_bits(t2n(env.env_indices), ref.env_idx, f"replay {rep} env_idx")
        assert_bits(t2n(env.env_spots[:, 0]), ref.spot0, f"replay {rep} spot0")
        assert_bits(t2n(roll.observation()), obs, f"replay {rep} observation()")
        if evaluate and int(ref.n_terminated[0]) == N:
            env.reset_evaluation_metrics()
            ref.terminated[:] = 0; ref.episode_returns[:] = 0; ref.n_terminated[0] = 0
    assert len(seen) >= 4, f"the policy must actually trade in both directions (share changes seen: {sorted(seen)})"


def test_lstm_head_against_torch_nn_lstm_fp32(fe, fo):
    """tanh(Linear(LSTM(states.float())[:, -1, :])) with torch's own modules, per asset, on the rendered observation:
    within 1e-5 absolute of the in-kernel head (measured ~1e-7); clamp output form as well."""
    from finenvs_amd.rollout import FusedLSTMRollout

    N, A, W, H = 500, 3, 4, 128
    ref, env = _make(fe, fo, N, A, W, 5, 100, 0.0, False, seed=21)
    lstm, lin = _modules(H, seed=0)
    roll = FusedLSTMRollout.from_modules(env, lstm, lin)
    worst = 0.0
    for t in range(40):
        obs = roll.observation().float().cpu()
        with torch.no_grad():
            want = torch.stack([torch.tanh(lin(lstm(obs[:, :, 5 * a:5 * a + 5])[0][:, -1, :])).squeeze(1) for a in range(A)], dim=1)
        acts, _, _ = roll.run(1)
        worst = max(worst, float((acts[0].cpu() - want).abs().max()))
        torch.testing.assert_close(acts[0].cpu(), want, rtol=0, atol=1e-5)  # tolerance: 1e-5 absolute
    assert float(want.abs().max()) > 0.05 and float(want.std()) > 0.01
    print(f"worst |kernel - torch nn.LSTM| action difference {worst:.3g}")
    clamp = FusedLSTMRollout.from_modules(env, lstm, lin, output_activation="clamp")
    obs = clamp.observation().float().cpu()
    with torch.no_grad():
        want = torch.stack([lin(lstm(obs[:, :, 5 * a:5 * a + 5])[0][:, -1, :]).squeeze(1) for a in range(A)], dim=1).clamp(-1, 1)
    acts, _, _ = clamp.run(1)
    torch.testing.assert_close(acts[0].cpu(), want, rtol=0, atol=1e-5)


def test_lstm_rollout_argument_errors(fe, fo):
    from finenvs_amd._lib import FinEnvsNativeError
    from finenvs_amd.rollout import FusedLSTMRollout

    ref, env = _make(fe, fo, 10, 1, 4, 5, 40, 0.0, False, seed=1)
    z = torch.zeros
    with pytest.raises(ValueError):
        FusedLSTMRollout(env, z((4 * 48, 5)), z((4 * 48, 48)), z(4 * 48), z(4 * 48), z(48))      # H not a supported size
    with pytest.raises(ValueError):
        FusedLSTMRollout(env, z((128, 4)), z((128, 32)), z(128), z(128), z(32))                  # wrong input size
    with pytest.raises(ValueError):
        FusedLSTMRollout(env, z((128, 5)), z((128, 32)), z(128), z(128), z(32), output_activation="relu")
    with pytest.raises(ValueError):
        FusedLSTMRollout.from_modules(env, torch.nn.LSTM(5, 32, num_layers=2), torch.nn.Linear(32, 1))
    # more sleeves per env than a workgroup tile holds is refused, not mis-tiled
    P, LR = _tables(fo, 3, 70, 30, 4)
    wide = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=4, num_envs=4, redraw="device")
    lstm, lin = _modules(128, seed=1)
    roll = FusedLSTMRollout.from_modules(wide, lstm, lin)
    with pytest.raises(FinEnvsNativeError, match="assets"):
        roll.run(1)


def test_lstm_evaluation_loop_returns_match_stepwise_oracle(fe, fo):
    """FusedLSTMRollout.evaluate_returns = the reference's evaluation loop (PPO_LSTM_testing_SPY.py:43-52:
    ``actions = test_actor.forward(states.float())`` until ``"returns" in info``), K steps per launch: the per-env
    episode returns equal the oracle stepped one action at a time, bit for bit."""
    from finenvs_amd.rollout import FusedLSTMRollout

    N, A, W, H = 90, 1, 4, 64
    ref, env = _make(fe, fo, N, A, W, 6, 40, 0.1, True, seed=5)
    lstm, lin = _modules(H, seed=2)
    whh, wx, wout, bout = _packed(fo, lstm, lin)
    roll = FusedLSTMRollout.from_modules(env, lstm, lin)
    got = t2n(roll.evaluate_returns(chunk=16))
    ref.auto_emit = True
    obs = ref.reset().copy()
    want = None
    for _ in range(2000):
        obs, _, _, info = ref.step(fo.policy_lstm(obs, whh, wx, wout, bout))
        obs = obs.copy()
        if "returns" in info:
            want = info["returns"]
            break
    assert want is not None
    assert_bits(got, want, "episode returns")


@pytest.mark.parametrize("N,A,W,H", [(150, 1, 4, 64), (45, 3, 5, 32)])
def test_lstm_training_rollout_samples_and_fills_the_trajectory(fe, fo, N, A, W, H):
    """The K-step form of the reference's TRAINING loop (PPO_LSTM_training_SPY.py:22-28 with agent.step of
    PPO_agent.py:98-108): actions = clamp(mean + std * noise), the eval env (last env, training mode) acts on the mean,
    and agent.store's fields -- states as descriptors, actions, rewards, dones -- land in the trajectory chunk.
    Everything equals the oracle stepped one action at a time, bit for bit; the stored states render to the
    observations the oracle's policy saw."""
    from finenvs_amd.rollout import FusedLSTMRollout
    from finenvs_amd.trajectory import TrajectoryBuffer

    ref, env = _make(fe, fo, N, A, W, 6, 40, 0.05, False, seed=N + H)
    assert env._eval_env == N - 1  # training mode: the last env is the evaluation env
    lstm, lin = _modules(H, seed=H)
    whh, wx, wout, bout = _packed(fo, lstm, lin)
    roll = FusedLSTMRollout.from_modules(env, lstm, lin)
    K, std = 12, np.float32(0.6)
    traj = TrajectoryBuffer(K, N, A, states=True)
    g = torch.Generator(device="cuda").manual_seed(11)
    obs = ref.reset().copy()
    sampled_beyond_clamp = 0
    for chunk in range(8):
        noise = torch.randn((K, N, A), generator=g, device="cuda")
        acts, rews, dones = roll.run(K, noise=noise, std=float(std), record_means=True, trajectory=traj)
        assert acts.data_ptr() == traj.actions.data_ptr() and len(traj) == K and traj.full()
        z = t2n(noise)
        for k in range(K):
            mean = fo.policy_lstm(obs, whh, wx, wout, bout)
            a_ref = np.clip((mean + (std * z[k]).astype(np.float32)).astype(np.float32), np.float32(-1), np.float32(1))
            a_ref[N - 1] = mean[N - 1]
            sampled_beyond_clamp += int((np.abs(mean + std * z[k]) > 1).sum())
            what = f"chunk {chunk} step {k}"
            assert_bits(t2n(roll.means[k]), mean, what + " means")
            assert_bits(t2n(acts[k]), a_ref, what + " actions")
            want_state = obs
            assert_bits(t2n(traj.states(env, k)), want_state, what + " stored state")
            obs, r_ref, d_ref, _ = ref.step(a_ref)
            obs = obs.copy()
            assert_bits(t2n(rews[k]), r_ref, what + " rewards")
            assert_bits(t2n(dones[k]), d_ref, what + " dones")
        assert_bits(t2n(traj.states(env, K)), obs, f"chunk {chunk} bootstrap state")
        assert_bits(t2n(env.cash), ref.cash, f"chunk {chunk} cash")
        assert_bits(t2n(env.env_indices), ref.env_idx, f"chunk {chunk} env_idx")
        traj.clear()
    assert sampled_beyond_clamp > 0  # the clamp of the sample was exercised
    # log_prob as the agent computes it (PPO_agent.py:108) needs nothing but what the rollout returned
    lp = torch.distributions.Normal(roll.means, float(std)).log_prob(acts)
    assert torch.isfinite(lp).all()
    with pytest.raises(ValueError):
        roll.run(K, noise=torch.randn((K, N, A), device="cuda"))  # std missing
    with pytest.raises(ValueError):
        roll.run(K + 1, trajectory=traj)  # wrong length


def test_ppo_lstm_example_runs_on_the_fused_rollout(fe):
    """examples/ppo_lstm_fused.py: two PPO iterations (fused sampled rollout -> values on rendered states -> returns
    kernel -> minibatch updates from descriptors -> new weights into the kernel) run end to end and produce finite
    numbers; the actor the kernel evaluates afterwards is the updated one."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "ppo_lstm_fused.py")
    spec = importlib.util.spec_from_file_location("ppo_lstm_fused", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    history = mod.main(envs=512, steps=8, iters=2, hidden=32, window=4, quiet=True)
    assert len(history) == 2
    for critic_loss, mean_reward, log in history:
        assert np.isfinite(critic_loss) and np.isfinite(mean_reward)
        assert log["num_training_episodes"] >= 0


def test_lstm_rollout_shape_sweep_bit_for_bit(fe, fo):
    """A seeded sweep over the kernel's tiling cases -- H in {32, 64, 128} (row tiles per wavefront / column-tile split),
    1..30 sleeves (pairs per tile not a multiple of 32, envs per tile from 128 down to 2), W 1..9, env counts that leave
    partial tiles and partial column-tile groups, evaluate and training mode, sampled and mean actions: every action,
    reward, done and state array equals the oracle loop bit for bit."""
    from finenvs_amd.rollout import FusedLSTMRollout

    rng = np.random.default_rng(2026)
    cases = 0
    for H in (32, 64, 128):
        for _ in range(6):
            A = int(rng.choice([1, 1, 2, 3, 5, 7, 12, 30]))
            W = int(rng.integers(1, 10))
            N = int(rng.integers(1, 400 // A + 2))
            evaluate = bool(rng.integers(0, 2))
            sample = bool(rng.integers(0, 2))
            ref, env = _make(fe, fo, N, A, W, 5, 30, 0.05, evaluate, seed=int(rng.integers(1, 1000)))
            lstm, lin = _modules(H, seed=int(rng.integers(1, 1000)))
            whh, wx, wout, bout = _packed(fo, lstm, lin)
            roll = FusedLSTMRollout.from_modules(env, lstm, lin)
            obs = ref.reset().copy()
            g = torch.Generator(device="cuda").manual_seed(cases)
            std = np.float32(0.4)
            for rep in range(4):
                K = int(rng.integers(1, 7))
                noise = torch.randn((K, N, A), generator=g, device="cuda") if sample else None
                acts, rews, dones = roll.run(K, noise=noise, std=float(std) if sample else None)
                for k in range(K):
                    a_ref = fo.policy_lstm(obs, whh, wx, wout, bout)
                    if sample:
                        smp = np.clip((a_ref + (std * t2n(noise[k])).astype(np.float32)).astype(np.float32), np.float32(-1), np.float32(1))
                        if not evaluate:
                            smp[N - 1] = a_ref[N - 1]
                        a_ref = smp
                    obs, r_ref, d_ref, _ = ref.step(a_ref)
                    obs = obs.copy()
                    what = f"case H={H} A={A} W={W} N={N} eval={evaluate} sample={sample} rep {rep} step {k}"
                    assert_bits(t2n(acts[k]), a_ref, what + " actions")
                    assert_bits(t2n(rews[k]), r_ref, what + " rewards")
                    assert_bits(t2n(dones[k]), d_ref, what + " dones")
                assert_bits(t2n(env.cash), ref.cash, "cash")
                assert_bits(t2n(env.margin), ref.margin, "margin")
                assert_bits(t2n(env.env_indices), ref.env_idx, "env_idx")
                if evaluate and int(ref.n_terminated[0]) == N:
                    env.reset_evaluation_metrics()
                    ref.terminated[:] = 0; ref.episode_returns[:] = 0; ref.n_terminated[0] = 0
            cases += 1
    assert cases == 18


@pytest.mark.parametrize("N,A,W,H", [(65536, 1, 4, 128), (65536, 1, 64, 32), (16384, 30, 4, 64)])
def test_lstm_rollout_full_size_sampled_oracle_parity(fe, fo, N, A, W, H):
    """(env, asset) pairs are independent, so the oracle loop run on a SAMPLE of envs must equal the fused rollout's
    outputs for those envs at full size (BASELINE config 2's 65 536 envs, with the reference scripts' W = 4 and with
    config 2's own W = 64; 16 384 envs x 30 sleeves), bit for bit."""
    from finenvs_amd.data import synthetic
    from finenvs_amd.rollout import FusedLSTMRollout

    prices, day_id, _ = synthetic.synthetic_series(9, A, 390, 1234)
    P, LR, *_ = fo.tables_from_series(prices, day_id, W)
    D, L, _ = P.shape
    env = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=W, num_envs=N, evaluate=True)
    rng = np.random.default_rng(N + W)
    sample = np.unique(np.concatenate([np.arange(40), np.arange(N - 40, N), rng.integers(0, N, 300)]))
    sidx = torch.from_numpy(sample).to(env.device)
    spot = torch.from_numpy(rng.integers(0, L - W - 1, N)).to(env.device)
    spot[sidx[::5]] = L - W - 1 - torch.arange(len(sidx[::5]), device=env.device) % 3  # some right at their episode's end
    env._spot0.copy_(spot)
    ref = fo.OracleEnv(P, LR, W, env_indices=env.env_indices[sidx].cpu().numpy(), evaluate=True, nthreads=8)
    ref.spot0[:] = spot[sidx].cpu().numpy()
    lstm, lin = _modules(H, seed=H + W)
    whh, wx, wout, bout = _packed(fo, lstm, lin)
    roll = FusedLSTMRollout.from_modules(env, lstm, lin)
    obs = ref.reset().copy()
    K = 3
    acts, rews, dones = roll.run(K)
    for k in range(K):
        a_ref = fo.policy_lstm(obs, whh, wx, wout, bout)
        obs, r_ref, d_ref, _ = ref.step(a_ref)
        obs = obs.copy()
        what = f"N={N} A={A} W={W} H={H} step {k}"
        assert_bits(t2n(acts[k][sidx]), a_ref, what + " actions")
        assert_bits(t2n(rews[k][sidx]), r_ref, what + " rewards")
        assert_bits(t2n(dones[k][sidx]), d_ref, what + " dones")
    assert_bits(t2n(env.cash[sidx]), ref.cash, "cash")
    assert_bits(t2n(env.margin[sidx]), ref.margin, "margin")
    assert_bits(t2n(roll.observation()[sidx]), obs, "observation()")
    assert int(dones.sum()) > 0


@pytest.mark.parametrize("N,A,W,H,sample", [
    (70, 1, 4, 256, False),     # weights streamed from L2: 4 row tiles per wavefront; 70 = two full tiles + 6 pairs
    (21, 3, 3, 512, True),      # 3 sleeves: 10 envs per 32-pair tile (30 pairs), sampled actions
    (33, 1, 4, 1024, False),    # the reference example's hidden_dim = 1024 (PPO_LSTM_training_SPY.py:16, testing:28)
    (4, 30, 2, 256, False),     # 30 sleeves: one env per tile
])
def test_lstm_rollout_large_hidden_sizes_bit_for_bit(fe, fo, N, A, W, H, sample):
    """H = 256 / 512 / 1024: the recurrent weights no longer fit a workgroup's registers and are streamed from L2 in
    fragment-major order (fe_rollout_lstm_big_kernel) -- same k order, same activations, same oracle: bit for bit."""
    from finenvs_amd.rollout import FusedLSTMRollout

    ref, env = _make(fe, fo, N, A, W, 6, 30, 0.05, False, seed=N + H)
    lstm, lin = _modules(H, seed=H, gain=2.0)
    whh, wx, wout, bout = _packed(fo, lstm, lin)
    roll = FusedLSTMRollout.from_modules(env, lstm, lin)
    roll.split = False  # THIS test pins the fused large-H kernel (so few pairs would otherwise take the per-time-step path)
    # the device copy is fragment-major: [row tile][k group][lane][4] of the same packed rows
    frag = whh.reshape(4 * H // 32, 32, H // 8, 2, 4).transpose(0, 2, 3, 1, 4).reshape(4 * H, H)
    assert_bits(t2n(roll.whh), np.ascontiguousarray(frag), "fragment-major Whh")
    obs = ref.reset().copy()
    g = torch.Generator(device="cuda").manual_seed(3)
    std = np.float32(0.5)
    seen = set()
    for rep in range(3):
        K = 4
        noise = torch.randn((K, N, A), generator=g, device="cuda") if sample else None
        acts, rews, dones = roll.run(K, noise=noise, std=float(std) if sample else None)
        for k in range(K):
            a_ref = fo.policy_lstm(obs, whh, wx, wout, bout)
            seen.update(np.unique(np.round(a_ref, 2)).tolist())
            if sample:
                smp = np.clip((a_ref + (std * t2n(noise[k])).astype(np.float32)).astype(np.float32), np.float32(-1), np.float32(1))
                smp[N - 1] = a_ref[N - 1]
                a_ref = smp
            obs, r_ref, d_ref, _ = ref.step(a_ref)
            obs = obs.copy()
            what = f"H={H} rep {rep} step {k}"
            assert_bits(t2n(acts[k]), a_ref, what + " actions")
            assert_bits(t2n(rews[k]), r_ref, what + " rewards")
            assert_bits(t2n(dones[k]), d_ref, what + " dones")
        assert_bits(t2n(env.cash), ref.cash, f"rep {rep} cash")
        assert_bits(t2n(roll.observation()), obs, f"rep {rep} observation()")
    assert len(seen) > 5, "the policy's outputs must vary across envs"


@pytest.mark.parametrize("H,A", [(64, 1), (128, 3), (256, 1)])
def test_lstm_forward_on_descriptors_is_the_critic_without_observations(fe, fo, H, A):
    """fe_lstm_forward: the same head on ANY descriptors, env untouched -- here a critic (no output activation) valuing
    all K + 1 states of a trajectory chunk in ONE launch.  Equal bit for bit to the oracle head on the rendered
    observations, within 1e-5 relative of torch's nn.LSTM + nn.Linear, and the env's state does not move."""
    from finenvs_amd.rollout import FusedLSTMRollout
    from finenvs_amd.trajectory import TrajectoryBuffer

    N, W, K = 120, 4, 6
    ref, env = _make(fe, fo, N, A, W, 6, 40, 0.05, False, seed=H + A)
    actor = _modules(32, seed=1)
    critic_lstm, critic_lin = _modules(H, seed=2, gain=3.0)
    roll = FusedLSTMRollout.from_modules(env, *actor)
    critic = FusedLSTMRollout.from_modules(env, critic_lstm, critic_lin, output_activation="none")
    traj = TrajectoryBuffer(K, N, A, states=True)
    roll.run(K, trajectory=traj)
    state_before = [t.clone() for t in (env.cash, env.margin, env.env_indices, env.env_spots[:, 0])]
    values = critic.forward(traj.obs_src, traj.obs_pos).reshape(K + 1, N, A)
    for t_, b in zip((env.cash, env.margin, env.env_indices, env.env_spots[:, 0]), state_before):
        assert torch.equal(t_, b), "forward() must not touch the env"
    whh, wx, wout, bout = _packed(fo, critic_lstm, critic_lin)
    for k in range(K + 1):
        obs = traj.states(env, k)
        want = fo.policy_lstm(t2n(obs), whh, wx, wout, bout, out_act=2)
        assert_bits(t2n(values[k]), want, f"values of state {k}")
        with torch.no_grad():
            o32 = obs.float().cpu()
            tv = torch.stack([critic_lin(critic_lstm(o32[:, :, 5 * a:5 * a + 5])[0][:, -1, :]).squeeze(1) for a in range(A)], 1)
        torch.testing.assert_close(values[k].cpu(), tv, rtol=1e-5, atol=1e-5)  # tolerance: 1e-5
    assert float(values.std()) > 1e-3  # the states differ and so do their values
    # a gathered minibatch of descriptors, any size
    idx = torch.randint(0, (K + 1) * N, (77,), device="cuda")
    mb = critic.forward(traj.obs_src.reshape(-1)[idx], traj.obs_pos.reshape(-1, A)[idx])
    assert torch.equal(mb, values.reshape(-1, A)[idx])
    assert critic.forward(traj.obs_src[:0], traj.obs_pos[:0]).shape == (0, A)
    from finenvs_amd._lib import FinEnvsNativeError
    with pytest.raises(RuntimeError, match="stale"):
        critic.run(1)  # `roll` advanced the env: the critic object's own descriptors are behind (forward() takes its descriptors as arguments)
    critic.sync_from_env()
    with pytest.raises(FinEnvsNativeError, match="out_activation"):
        critic.run(1)  # an action needs bounds: "none" is for forward() only


@pytest.mark.parametrize("N,A,W,H,sample,evaluate", [
    (9, 1, 4, 1024, False, True),     # the reference's own evaluation: one env per trading day of SPY dummy, hidden_dim 1024
    (70, 1, 4, 256, True, False),     # three column tiles (one partial), sampled actions, training mode
    (21, 3, 3, 512, False, False),    # sleeves: pairs = 63
    (150, 1, 1, 256, False, False),   # W = 1: a single gate launch per step; more column tiles than one workgroup row
])
def test_lstm_split_path_equals_oracle_and_fused_path(fe, fo, N, A, W, H, sample, evaluate):
    """fe_env_rollout_lstm_split (one launch per LSTM time step, gate-row tiles over the whole GPU, h / c in global
    memory): the same arithmetic as the fused kernel -- bit for bit against the oracle loop, trajectory rows included."""
    from finenvs_amd.rollout import FusedLSTMRollout
    from finenvs_amd.trajectory import TrajectoryBuffer

    ref, env = _make(fe, fo, N, A, W, 6, 30, 0.05, evaluate, seed=N + H)
    lstm, lin = _modules(H, seed=H + 1, gain=2.0)
    whh, wx, wout, bout = _packed(fo, lstm, lin)
    roll = FusedLSTMRollout.from_modules(env, lstm, lin)
    assert roll.split is None and N * A < roll.SPLIT_BELOW_PAIRS  # the automatic choice takes the split path here
    obs = ref.reset().copy()
    g = torch.Generator(device="cuda").manual_seed(3)
    std = np.float32(0.5)
    for rep in range(3):
        K = 4
        traj = TrajectoryBuffer(K, N, A, states=True)
        noise = torch.randn((K, N, A), generator=g, device="cuda") if sample else None
        acts, rews, dones = roll.run(K, noise=noise, std=float(std) if sample else None, trajectory=traj, record_means=True)
        assert roll._workspace is not None
        for k in range(K):
            assert_bits(t2n(traj.states(env, k)), obs, f"rep {rep} stored state {k}")
            mean = fo.policy_lstm(obs, whh, wx, wout, bout)
            assert_bits(t2n(roll.means[k]), mean, f"rep {rep} step {k} means")
            a_ref = mean
            if sample:
                a_ref = np.clip((mean + (std * t2n(noise[k])).astype(np.float32)).astype(np.float32), np.float32(-1), np.float32(1))
                if not evaluate:
                    a_ref[N - 1] = mean[N - 1]
            obs, r_ref, d_ref, _ = ref.step(a_ref)
            obs = obs.copy()
            what = f"H={H} rep {rep} step {k}"
            assert_bits(t2n(acts[k]), a_ref, what + " actions")
            assert_bits(t2n(rews[k]), r_ref, what + " rewards")
            assert_bits(t2n(dones[k]), d_ref, what + " dones")
        assert_bits(t2n(traj.states(env, K)), obs, f"rep {rep} bootstrap state")
        assert_bits(t2n(env.cash), ref.cash, f"rep {rep} cash")
        assert_bits(t2n(env.env_indices), ref.env_idx, f"rep {rep} env_idx")
        if evaluate and int(ref.n_terminated[0]) == N:
            env.reset_evaluation_metrics()
            ref.terminated[:] = 0; ref.episode_returns[:] = 0; ref.n_terminated[0] = 0
    # the fused kernel, forced, continues the same trajectory identically
    roll.split = False
    acts, rews, dones = roll.run(2)
    for k in range(2):
        a_ref = fo.policy_lstm(obs, whh, wx, wout, bout)
        obs, r_ref, d_ref, _ = ref.step(a_ref)
        obs = obs.copy()
        assert_bits(t2n(acts[k]), a_ref, f"fused continuation step {k} actions")
        assert_bits(t2n(rews[k]), r_ref, f"fused continuation step {k} rewards")


def test_lstm_evaluation_example_three_ways_agree(fe):
    """examples/lstm_evaluation.py: the reference's evaluation loop eager, graphed and fused -- the graphed loop equals the
    eager one bit for bit; the fused kernel's own (bit-reproducible) LSTM arithmetic differs from torch's by ~1e-7 per action,
    which moves a day's return by far less than a cent on a 10 000 balance."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "lstm_evaluation.py")
    spec = importlib.util.spec_from_file_location("lstm_evaluation", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.main(days=12, hidden=256, window=4, quiet=True)
    eager, graph, fused = (res[k][0] for k in ("eager", "graph", "fused"))
    assert eager.numel() == 12 and torch.equal(eager, graph)
    assert float(eager.abs().sum()) > 0
    torch.testing.assert_close(fused, eager, rtol=0, atol=0.5)  # dollars of episode return; share changes are integers
