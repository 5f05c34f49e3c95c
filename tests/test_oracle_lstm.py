"""CPU: the oracle's LSTM head (oracle/fe_oracle.c:fo_policy_lstm, the order-exact restatement the fused rollout's
LSTM kernel is compared with bit for bit) pinned against the REFERENCE's own actor network.

tests/golden/lstm_actor.npz was produced by oracle/make_goldens.py:lstm_actor_case, which runs the reference's
ContinuousActorLSTM (finenvs/agents/PPO/continuous_actor.py:104-126 over finenvs/agents/networks/lstm.py:7-57) on
reference-env observations the way examples/time_series/PPO_LSTM_testing_SPY.py:46 does.  Floating point, a
different summation order and libm's exp/tanh on the reference side: tolerance 2e-6 absolute on actions in (-1, 1).
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fo():
    from oracle import fe_oracle

    fe_oracle.build()
    return fe_oracle


@pytest.mark.parametrize("tag", ["h128_w4", "h32_w8"])
def test_oracle_lstm_head_matches_the_reference_actor(fo, tag):
    g = np.load(os.path.join(GOLD, "lstm_actor.npz"))
    H, W = int(g[f"{tag}_H"]), int(g[f"{tag}_W"])
    whh, wx = fo.lstm_pack(g[f"{tag}_weight_ih"], g[f"{tag}_weight_hh"], g[f"{tag}_bias_ih"], g[f"{tag}_bias_hh"])
    wout, bout = g[f"{tag}_weight_out"].reshape(H), float(g[f"{tag}_bias_out"].reshape(()))
    obs, want = g[f"{tag}_obs"], g[f"{tag}_actions"]
    assert obs.shape[2:] == (W, 5) and want.shape == obs.shape[:2] + (1,)
    worst = 0.0
    for t in range(obs.shape[0]):
        got = fo.policy_lstm(obs[t], whh, wx, wout, bout)
        worst = max(worst, float(np.abs(got - want[t]).max()))
        np.testing.assert_allclose(got, want[t], rtol=0, atol=2e-6)  # tolerance: 2e-6 absolute
    assert want.std() > 0.1  # the recorded actor really moves
    print(f"{tag}: worst |oracle - reference actor| = {worst:.3g}")


def test_lstm_pack_row_order_is_a_permutation_grouping_gates_by_unit(fo):
    for H in (32, 64, 128):
        order = fo.lstm_row_order(H)
        assert sorted(order.tolist()) == list(range(4 * H))
        R = np.arange(4 * H)
        gate, unit = order // H, order % H
        assert np.array_equal(gate, R % 4)  # four consecutive packed rows = the four gates i, f, g, o ...
        assert np.array_equal(unit, 8 * (R // 32) + 4 * ((R % 8) // 4) + (R % 32) // 8)  # ... of this hidden unit


def test_lstm_activation_forms_are_accurate(fo):
    rng = np.random.default_rng(0)
    x = np.concatenate([np.linspace(-100, 100, 400_001), rng.normal(0, 3, 100_000), rng.normal(0, 1e-3, 10_000),
                        [0.0, -0.0, 1e-8, -1e-8, 88.0, -88.0, 1e30, -1e30]]).astype(np.float32)
    sig, tnh = fo.lstm_activations(x)
    x64 = x.astype(np.float64)
    with np.errstate(over="ignore"):
        assert np.abs(sig - 1.0 / (1.0 + np.exp(-x64))).max() < 1e-7
    assert np.abs(tnh - np.tanh(x64)).max() < 1e-7
    assert np.all((sig >= 0) & (sig <= 1)) and np.all(np.abs(tnh) <= 1)
    assert np.array_equal(np.signbit(tnh), np.signbit(x))  # odd, including -0.0
    # NaN is not propagated (it acts like a pre-activation of -60 of either sign): the host refuses non-finite weights
    nan_s, nan_t = fo.lstm_activations(np.array([np.nan], dtype=np.float32))
    assert np.isfinite(nan_s[0]) and abs(nan_t[0]) == 1.0


def test_oracle_lstm_head_matches_torch_modules(fo):
    """Random nn.LSTM / nn.Linear modules, several shapes incl. multi-asset observations and W = 1."""
    import torch

    torch.manual_seed(0)
    rng = np.random.default_rng(1)
    for H, W, A, N in ((32, 4, 1, 40), (128, 4, 2, 12), (64, 7, 3, 10), (64, 1, 1, 10)):
        lstm, lin = torch.nn.LSTM(5, H, batch_first=True), torch.nn.Linear(H, 1)
        with torch.no_grad():
            lstm.weight_ih_l0.mul_(8.0)
        obs = rng.normal(0, 0.05, (N, W, 5 * A))
        whh, wx = fo.lstm_pack(*(p.detach().numpy() for p in (lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)))
        got, h = fo.policy_lstm(obs, whh, wx, lin.weight.detach().numpy().reshape(H), float(lin.bias.detach()), return_h=True)
        o32 = torch.from_numpy(obs).float()
        with torch.no_grad():
            hw = torch.stack([lstm(o32[:, :, 5 * a:5 * a + 5])[0][:, -1, :] for a in range(A)], 1)
            want = torch.tanh(lin(hw)).squeeze(-1)
        np.testing.assert_allclose(h, hw.numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(got, want.numpy(), rtol=0, atol=1e-6)  # tolerance: 1e-6 absolute
