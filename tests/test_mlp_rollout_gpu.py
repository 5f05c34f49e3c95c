"""GPU parity of the fused rollout's MLP head (fe_env_rollout_mlp, first layer on the matrix cores).

* ReLU: the whole K-step rollout -- actions, rewards, dones, state -- equals the oracle's loop
  ``actions = fo.policy_mlp(obs); obs, r, d = step(actions)`` BIT FOR BIT: the v_mfma_f32_32x32x2_f32 accumulation is
  an fmaf chain whose order oracle/fe_oracle.c:fo_policy_mlp restates.
* ELU (the reference's default activation): the device's v_exp_f32 differs from libm's expm1f by an ulp or two, so the
  actions are compared with a tolerance of 2e-6 absolute (actions live in [-1, 1]); the env is kept in lock-step by
  feeding the DEVICE's actions to the oracle, and everything downstream of the actions stays bit-exact.
* tanh: the exact-operation form shared with the LSTM head -- bit for bit, like ReLU.
* against a plain fp32 PyTorch ``nn.Sequential(Flatten, Linear, ELU, Linear)`` on the rendered observation: 1e-5
  absolute (a different summation order: that is the tolerance north_star states for floating point).
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


def _weights(W, H, seed):
    rng = np.random.default_rng(seed)
    # log-returns are ~5e-2 and the position feature ~1e-1: scale the first layer so that the pre-activations
    # vary by O(1) from env to env and the actions cover [-1, 1] without saturating everywhere
    W1 = (rng.normal(0, 1.0, (5 * W, H)) * (12.0 / np.sqrt(W))).astype(np.float32)
    W1[4::5, :] = (rng.normal(0, 1.0, (W, H)) * (3.0 / W)).astype(np.float32)
    b1 = rng.normal(0, 0.3, H).astype(np.float32)
    W2 = (rng.normal(0, 1.0, H) * (0.9 / np.sqrt(H))).astype(np.float32)
    return W1, b1, W2, np.float32(0.03)


def _make(fe, fo, N, A, W, days, bars, drop, evaluate, seed):
    P, LR = _tables(fo, days, A, bars, W, seed=seed, drop=drop)
    D = P.shape[0]
    idx = (np.arange(N) * 5 + 1) % D
    kw = dict(num_intervals=W, evaluate=evaluate, starting_balance=2000)
    ref = fo.OracleEnv(P, LR, env_indices=idx, redraw_mode=1, seed=9, auto_emit=False, **kw)
    ref.redraw_counter[0] = 1
    env = fe.TimeSeriesEnv(tables=(P, LR), env_indices=idx, redraw="device", seed=9, **kw)
    return ref, env


@pytest.mark.parametrize("N,A,W,H,days,bars,drop,evaluate", [
    (300, 1, 8, 32, 6, 40, 0.0, False),
    (1000, 1, 64, 64, 5, 100, 0.05, False),   # BASELINE config-2 window, H = 64
    (77, 3, 7, 64, 4, 40, 0.0, False),        # odd W: the last row group is half empty
    (50, 30, 16, 32, 5, 40, 0.1, True),       # DJIA-shaped sleeves, evaluate mode
    (200, 1, 16, 128, 5, 40, 0.0, False),     # four hidden tiles
    (131, 5, 4, 32, 6, 45, 0.1, False),       # pairs per workgroup not a multiple of 32
])
def test_mlp_rollout_relu_equals_oracle_loop_bit_for_bit(fe, fo, N, A, W, H, days, bars, drop, evaluate):
    from finenvs_amd.rollout import FusedMLPRollout

    ref, env = _make(fe, fo, N, A, W, days, bars, drop, evaluate, seed=3 * N + W)
    W1, b1, W2, b2 = _weights(W, H, seed=W + H)
    w1t, wpos = fo.mlp_pack(W1, W)
    roll = FusedMLPRollout(env, torch.from_numpy(W1), torch.from_numpy(b1), torch.from_numpy(W2), float(b2), activation="relu")
    assert_bits(t2n(roll.w1t), w1t, "packed W1t")
    assert_bits(t2n(roll.wpos), wpos, "position weights")
    obs = ref.reset().copy()
    assert_bits(t2n(roll.observation()), obs, "initial obs")
    K, reps = 5, 2 * (bars + 3) // 5 + 1
    seen = set()
    for rep in range(reps):
        acts, rews, dones = roll.run(K)
        for k in range(K):
            a_ref = fo.policy_mlp(obs, w1t, wpos, b1, W2, b2, act=1)
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
    assert len(seen) >= 5, "the policy must actually trade in both directions"


@pytest.mark.parametrize("activation,act,H,atol", [("elu", 0, 64, 2e-6), ("tanh", 2, 64, 0.0), ("tanh", 2, 128, 0.0)])
def test_mlp_rollout_elu_tanh_within_tolerance_in_lockstep(fe, fo, activation, act, H, atol):
    """ELU: tolerance 2e-6 absolute (v_exp_f32 vs expm1f).  tanh: the exact-operation form shared with the LSTM head
    (fe_activations.h; restated by the oracle as fo_lstm_tanh) -- the actions are the oracle's bit for bit."""
    from finenvs_amd.rollout import FusedMLPRollout

    N, A, W = 400, 2, 32
    ref, env = _make(fe, fo, N, A, W, 5, 60, 0.05, False, seed=11)
    W1, b1, W2, b2 = _weights(W, H, seed=5)
    w1t, wpos = fo.mlp_pack(W1, W)
    roll = FusedMLPRollout(env, torch.from_numpy(W1), torch.from_numpy(b1), torch.from_numpy(W2), float(b2), activation=activation)
    obs = ref.reset().copy()
    worst = 0.0
    for t in range(130):
        acts, rews, dones = roll.run(1)
        a_dev = t2n(acts[0])
        a_ref = fo.policy_mlp(obs, w1t, wpos, b1, W2, b2, act=act)
        worst = max(worst, float(np.abs(a_dev.astype(np.float64) - a_ref).max()))
        np.testing.assert_allclose(a_dev, a_ref, rtol=0, atol=atol, err_msg=f"step {t} actions")  # tolerance: 2e-6 absolute (ELU), 0 (tanh)
        obs, r_ref, d_ref, _ = ref.step(a_dev)  # lock-step on the device's actions
        obs = obs.copy()
        assert_bits(t2n(rews[0]), r_ref, f"step {t} rewards")
        assert_bits(t2n(dones[0]), d_ref, f"step {t} dones")
        assert_bits(t2n(env.cash), ref.cash, f"step {t} cash")
    assert 0.05 < float(np.median(np.abs(a_ref))) < 0.95  # mostly unsaturated: the comparison means something
    print(f"{activation}: worst |device - oracle| action difference {worst:.3g}")


def test_mlp_head_against_plain_fp32_pytorch(fe, fo):
    """nn.Sequential(Flatten, Linear(5W, H), ELU, Linear(H, 1)) per asset on the rendered observation (.float(), as
    PPO_agent.py:101 does), clamped like PPO_agent.py:104: within 1e-5 absolute of the in-kernel MFMA head."""
    from finenvs_amd.rollout import FusedMLPRollout

    N, A, W, H = 500, 3, 64, 64
    ref, env = _make(fe, fo, N, A, W, 5, 100, 0.0, False, seed=21)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(5 * W, H), torch.nn.ELU(), torch.nn.Linear(H, 1))
    with torch.no_grad():
        net[1].weight.mul_(4.0)
        net[1].weight[:, 4::5].mul_(10.0)
    W1 = net[1].weight.detach().t().contiguous()  # (5W, H)
    roll = FusedMLPRollout(env, W1, net[1].bias.detach(), net[3].weight.detach().reshape(H), float(net[3].bias), activation="elu")
    for t in range(40):
        obs = roll.observation().float().cpu()  # what the next in-kernel policy evaluation sees
        with torch.no_grad():
            want = torch.stack([net(obs[:, :, 5 * a:5 * a + 5]).squeeze(1) for a in range(A)], dim=1).clamp(-1, 1)
        acts, _, _ = roll.run(1)
        torch.testing.assert_close(acts[0].cpu(), want, rtol=0, atol=1e-5)  # tolerance: 1e-5 absolute
    assert float(want.abs().max()) > 0.05


def test_mlp_rollout_argument_errors(fe, fo):
    from finenvs_amd._lib import FinEnvsNativeError
    from finenvs_amd.rollout import FusedMLPRollout

    ref, env = _make(fe, fo, 10, 1, 8, 5, 40, 0.0, False, seed=1)
    W1, b1, W2, b2 = _weights(8, 32, seed=1)
    with pytest.raises(ValueError):
        FusedMLPRollout(env, torch.zeros((40, 48)), torch.zeros(48), torch.zeros(48))   # H not a tile multiple
    with pytest.raises(ValueError):
        FusedMLPRollout(env, torch.zeros((39, 32)), torch.zeros(32), torch.zeros(32))   # wrong row count
    with pytest.raises(ValueError):
        FusedMLPRollout(env, torch.from_numpy(W1), torch.from_numpy(b1), torch.from_numpy(W2), activation="gelu")
    # a first layer that does not fit the 160 KiB LDS is refused, not truncated
    P, LR = _tables(fo, 3, 1, 500, 390)
    big = fe.TimeSeriesEnv(tables=(P, LR), num_intervals=390, num_envs=4, redraw="device")
    roll = FusedMLPRollout(big, torch.zeros((5 * 390, 128)), torch.zeros(128), torch.zeros(128))
    with pytest.raises(FinEnvsNativeError, match="LDS"):
        roll.run(1)


def test_fused_rollouts_refuse_stale_descriptors(fe, fo):
    """A fused rollout object keeps its own observation descriptors; the account state lives in the env.  Stepping the
    env behind its back (env.step, another rollout object) leaves the descriptors one observation behind: run() must
    refuse until sync_from_env() is called, and then equal a fresh rollout object on the same state."""
    from finenvs_amd.rollout import FusedLinearRollout, FusedMLPRollout

    ref, env = _make(fe, fo, 40, 1, 8, 5, 40, 0.0, False, seed=1)
    W1, b1, W2, b2 = _weights(8, 32, seed=1)
    mlp = FusedMLPRollout(env, torch.from_numpy(W1), torch.from_numpy(b1), torch.from_numpy(W2), b2, activation="relu")
    lin = FusedLinearRollout(env, torch.ones((8, 5), dtype=torch.float64) * 0.1, 0.0)
    mlp.run(2)
    mlp.run(1)  # its own runs keep it fresh
    with pytest.raises(RuntimeError, match="stale"):
        lin.run(1)  # built before mlp advanced the env
    env.step(torch.zeros((40, 1), device=env.device))
    with pytest.raises(RuntimeError, match="stale"):
        mlp.run(1)
    mlp.sync_from_env()
    state = (env.cash, env.margin, env.long_shares, env.short_shares, env._spot0, env.env_indices, env._counters)
    saved = [t.clone() for t in state]
    a1, r1, d1 = mlp.run(3)
    # replay from the same state with a rollout object created there
    for dst, src in zip(state, saved):
        dst.copy_(src)
    fresh = FusedMLPRollout(env, torch.from_numpy(W1), torch.from_numpy(b1), torch.from_numpy(W2), b2, activation="relu")
    a2, r2, d2 = fresh.run(3)
    assert torch.equal(a1, a2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    # a hipGraph replay advances the env without passing through env.step(): it must count as "someone else" too
    from finenvs_amd.rollout import GraphedRollout

    env2 = fe.TimeSeriesEnv(tables=(env.price_environments.cpu().numpy(), env.log_return_environments.cpu().numpy()),
                            num_intervals=8, num_envs=40, redraw="device", obs_buffers=1, seed=1)
    lin2 = FusedLinearRollout(env2, torch.ones((8, 5), dtype=torch.float64) * 0.1, 0.0)
    zeros = torch.zeros((40, 1), device=env2.device)
    graphed = GraphedRollout(env2, lambda obs, k: zeros, 2)
    lin2.sync_from_env()
    lin2.run(1)
    graphed.run()
    with pytest.raises(RuntimeError, match="stale"):
        lin2.run(1)


def test_mlp_evaluation_loop_returns_match_stepwise_oracle(fe, fo):
    """FusedMLPRollout.evaluate_returns = the reference's evaluation loop (PPO_LSTM_testing_SPY.py:43-52) with an MLP
    actor, K steps per launch: the per-env episode returns equal the oracle stepped one action at a time (ReLU head:
    bit for bit)."""
    from finenvs_amd.rollout import FusedMLPRollout

    N, A, W, H = 90, 2, 8, 32
    ref, env = _make(fe, fo, N, A, W, 6, 40, 0.1, True, seed=5)
    W1, b1, W2, b2 = _weights(W, H, seed=2)
    w1t, wpos = fo.mlp_pack(W1, W)
    roll = FusedMLPRollout(env, torch.from_numpy(W1), torch.from_numpy(b1), torch.from_numpy(W2), float(b2), activation="relu")
    got = t2n(roll.evaluate_returns(chunk=16))
    ref.auto_emit = True
    obs = ref.reset().copy()
    want = None
    for _ in range(2000):
        obs, _, _, info = ref.step(fo.policy_mlp(obs, w1t, wpos, b1, W2, b2, act=1))
        obs = obs.copy()
        if "returns" in info:
            want = info["returns"]
            break
    assert want is not None
    assert_bits(got, want, "episode returns")
