"""GPU: a trajectory that keeps the PPO buffer's ``states`` as descriptors.

The reference's buffer stores every step's observation (finenvs/agents/PPO/buffer.py:33-56) and training indexes
minibatches out of it (PPO_agent.py:175-188).  ``TrajectoryBuffer(states=True)`` stores 8 + 8A bytes per env-step
instead -- written by the step kernel itself (fe_env_step_traj) -- and renders on demand; every rendered state
must equal, bit for bit, the observation the loop actually fed to the policy, including terminal windows on done
steps and the bootstrap state after the last step, and the oracle's observation at the same step.
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


def _make(fe, fo, N, A, W, days, bars, drop, seed, obs_dtype=torch.float64):
    from finenvs_amd.data import synthetic

    prices, day_id, _ = synthetic.synthetic_series(days, A, bars, seed, drop)
    P, LR, *_ = fo.tables_from_series(prices, day_id, W)
    idx = (np.arange(N) * 7 + 1) % P.shape[0]
    kw = dict(num_intervals=W, starting_balance=1500)
    ref = fo.OracleEnv(P, LR, env_indices=idx, redraw_mode=1, seed=4, auto_emit=False, **kw)
    ref.redraw_counter[0] = 1
    env = fe.TimeSeriesEnv(tables=(P, LR), env_indices=idx, redraw="device", seed=4, obs_dtype=obs_dtype, **kw)
    return ref, env


@pytest.mark.parametrize("N,A,W,T,obs_dtype", [
    (200, 1, 8, 25, torch.float64),    # chunks of 25 steps over 40-bar days: every chunk crosses day ends
    (61, 3, 5, 16, torch.float64),
    (37, 30, 4, 9, torch.float32),     # f32 observations render from the f32 table
])
def test_states_kept_as_descriptors_render_the_observations_the_policy_saw(fe, fo, N, A, W, T, obs_dtype):
    from finenvs_amd.trajectory import TrajectoryBuffer

    ref, env = _make(fe, fo, N, A, W, 6, 40, 0.05, seed=N + W, obs_dtype=obs_dtype)
    traj = TrajectoryBuffer(T, N, A, states=True)
    g = torch.Generator(device="cuda").manual_seed(3)
    obs = env.reset()
    obs_ref = ref.reset().copy()
    ndone = 0
    for chunk in range(-(-80 // T)):  # 80+ steps: every env passes at least one day end (40-bar days)
        if chunk == 0:
            traj.begin(env)
        seen, seen_ref = [], []
        for t in range(T):
            seen.append(obs.clone())          # what the policy is looking at
            seen_ref.append(obs_ref.copy())
            a_slot, r_slot, d_slot = traj.next_slot()
            a_slot.copy_(torch.rand((N, A), generator=g, device="cuda") * 2 - 1)
            obs, r, d, _ = env.step(a_slot, rewards_out=r_slot, dones_out=d_slot, descriptors_out=traj.state_slot())
            obs_ref, r_ref, d_ref, _ = ref.step(t2n(a_slot))
            obs_ref = obs_ref.copy()
            assert_bits(t2n(r), r_ref, f"chunk {chunk} step {t} rewards")
            ndone += int(d_ref.sum())
        assert traj.full()
        for t in range(T):
            got = traj.states(env, t)
            assert torch.equal(got, seen[t]), f"chunk {chunk} state {t}"
            want = seen_ref[t] if obs_dtype is torch.float64 else seen_ref[t].astype(np.float32)
            assert_bits(t2n(got), want, f"chunk {chunk} state {t} vs oracle")
        assert torch.equal(traj.states(env, T), obs)  # the bootstrap state (PPO_agent.py:171 `current_states`)
        # minibatches, numbered like the reference's reshaped buffer: sample = env * steps + step
        idx = torch.randint(0, N * T, (97,), generator=g, device="cuda")
        mb = traj.minibatch_states(env, idx)
        stacked = torch.stack(seen, dim=1)  # (N, T, W, 5A): the reference's container["states"] layout
        assert torch.equal(mb, stacked.reshape(N * T, W, 5 * A)[idx])
        traj.clear()  # carries the bootstrap state over as row 0 of the next chunk
    assert ndone >= N, "the rollout must cross episode ends (terminal windows are the interesting states)"


def test_state_descriptor_api_errors_and_plain_describe_render(fe, fo):
    from finenvs_amd.trajectory import TrajectoryBuffer

    ref, env = _make(fe, fo, 50, 2, 6, 5, 40, 0.0, seed=9)
    # describe / render round trip on the current state, any batch size, out= reuse
    obs = env.reset()
    src, pos = env.describe()
    assert torch.equal(env.render(src, pos), obs)
    pick = torch.tensor([3, 3, 49, 0, 17], device="cuda")
    out = torch.empty((5, 6, 10), dtype=torch.float64, device="cuda")
    assert env.render(src[pick], pos[pick], out=out) is out and torch.equal(out, obs[pick])
    assert env.render(src[:0], pos[:0]).shape == (0, 6, 10)
    with pytest.raises(ValueError):
        env.render(src, pos, out=torch.empty((50, 6, 10), dtype=torch.float32, device="cuda"))
    plain = TrajectoryBuffer(4, 50, 2)
    with pytest.raises(RuntimeError, match="states=True"):
        plain.begin(env)
    traj = TrajectoryBuffer(4, 50, 2, states=True)
    a, r, d = traj.next_slot()
    with pytest.raises(RuntimeError, match="begin"):
        traj.state_slot()
    traj.clear()
    # a step that did not record its observation leaves nothing to begin a chunk from
    env.step(torch.zeros((50, 2), device="cuda"))
    with pytest.raises(RuntimeError, match="descriptors_out"):
        traj.begin(env)
    env.reset()
    traj.begin(env)
    with pytest.raises(ValueError):
        env.step(torch.zeros((50, 2), device="cuda"), descriptors_out=(torch.empty(50, device="cuda"), torch.empty((50, 2), device="cuda")))


def test_another_ranks_descriptors_render_on_this_rank(fe, fo):
    """Tables are replicated across ranks, so descriptors gathered from another rank's shard render on
    this rank's env object: here two shard envs of one 40-env job on one GPU stand in for two ranks."""
    from finenvs_amd.data import synthetic
    from finenvs_amd.environments.time_series_env import shard_range
    from finenvs_amd.trajectory import TrajectoryBuffer

    N, A, W, T = 40, 2, 6, 12
    prices, day_id, _ = synthetic.synthetic_series(6, A, 40, 3, 0.0)
    P, LR, *_ = fo.tables_from_series(prices, day_id, W)
    idx = (np.arange(N) * 7 + 1) % P.shape[0]
    shards = []
    for r in range(2):
        lo, hi = shard_range(N, r, 2)
        env = fe.TimeSeriesEnv(tables=(P, LR), env_indices=idx[lo:hi], redraw="device", seed=4, num_intervals=W, starting_balance=1500)
        shards.append((env, TrajectoryBuffer(T, hi - lo, A, states=True)))
    g = torch.Generator(device="cuda").manual_seed(5)
    seen = [[], []]
    for r, (env, traj) in enumerate(shards):
        obs = env.reset()
        traj.begin(env)
        for t in range(T):
            seen[r].append(obs.clone())
            a, rew, d = traj.next_slot()
            a.copy_(torch.rand(a.shape, generator=g, device="cuda") * 2 - 1)
            obs, *_ = env.step(a, rewards_out=rew, dones_out=d, descriptors_out=traj.state_slot())
    # "rank 0" renders "rank 1"'s states from nothing but the descriptors it would have received in the all-gather
    env0, (_, traj1) = shards[0][0], shards[1]
    for t in range(T):
        got = env0.render(traj1.obs_src[t].clone(), traj1.obs_pos[t].clone())
        assert torch.equal(got, seen[1][t]), f"state {t}"
    # bytes: what the states cost per env-step against the observations they stand for
    assert traj1._layout["obs_src"][1] + traj1._layout["obs_pos"][1] == (T + 1) * traj1.C * (8 + 8 * A)
    assert seen[1][0][0].numel() * 8 == 40 * W * A


def test_graphed_rollout_keeps_states_as_descriptors(fe, fo):
    """A hipGraph of K x (policy -> step) with a states=True trajectory: no store launch in the graph (the step kernel
    writes actions, rewards, dones and descriptors into the slots), row 0 of every replay's chunk is the previous
    replay's bootstrap row, and every stored state renders to the observation an eager loop sees at that step."""
    from finenvs_amd.data import synthetic
    from finenvs_amd.rollout import GraphedRollout
    from finenvs_amd.trajectory import TrajectoryBuffer

    N, A, W, K = 300, 2, 6, 8
    prices, day_id, _ = synthetic.synthetic_series(6, A, 40, 3, 0.05)
    P, LR, *_ = fo.tables_from_series(prices, day_id, W)
    g = torch.Generator().manual_seed(1)
    ring = [(torch.rand((N, A), generator=g) * 2 - 1).float().cuda() for _ in range(K)]
    mk = lambda: fe.TimeSeriesEnv(tables=(P, LR), num_intervals=W, num_envs=N, redraw="device", seed=5, obs_buffers=2)
    eager, graphed = mk(), mk()
    traj = TrajectoryBuffer(K, N, A, states=True)
    roll = GraphedRollout(graphed, lambda obs, k: ring[k], K, trajectory=traj, warmup=0)
    obs_e = eager.reset().clone()
    for rep in range(10):  # 80 steps: crosses day ends
        roll.run()
        for k in range(K):
            assert torch.equal(traj.states(graphed, k), obs_e), f"replay {rep} state {k}"
            obs_e, rew_e, done_e, _ = eager.step(ring[k])
            obs_e = obs_e.clone()
            assert torch.equal(traj.rewards[k], rew_e) and torch.equal(traj.dones[k], done_e) and torch.equal(traj.actions[k], ring[k])
        assert torch.equal(traj.states(graphed, K), obs_e), f"replay {rep} bootstrap state"


def test_graphed_evaluation_loop_with_a_torch_lstm_actor(fe, fo):
    """The reference's evaluation loop (PPO_LSTM_testing_SPY.py:43-52) with the torch nn.LSTM actor ITSELF as the
    policy, K steps per hipGraph replay (evaluate mode: the "all terminated?" host read moves to the end of a replay):
    the episode returns equal the eager step-by-step loop's, bit for bit."""
    from finenvs_amd.data import synthetic
    from finenvs_amd.rollout import GraphedRollout

    N, W, H, K = 64, 4, 32, 8
    prices, day_id, _ = synthetic.synthetic_series(7, 1, 40, 3, 0.05)
    P, LR, *_ = fo.tables_from_series(prices, day_id, W)
    torch.manual_seed(0)
    lstm, lin = torch.nn.LSTM(5, H, batch_first=True).cuda(), torch.nn.Linear(H, 1).cuda()
    with torch.no_grad():
        lstm.weight_ih_l0.mul_(30.0)

    @torch.no_grad()
    def actor(states, k=0):
        return torch.tanh(lin(lstm(states.float())[0][:, -1, :]))

    mk = lambda: fe.TimeSeriesEnv(tables=(P, LR), num_intervals=W, num_envs=N, evaluate=True, obs_buffers=2)
    eager, graphed = mk(), mk()
    states = eager.reset()
    want = None
    for _ in range(400):
        states, _, _, info = eager.step(actor(states))
        if "returns" in info:
            want = info["returns"]
            break
    assert want is not None
    roll = GraphedRollout(graphed, actor, K, warmup=0)
    got = roll.evaluate_returns()
    assert torch.equal(got, want)
    assert float(got.abs().sum()) > 0


def test_foreign_descriptors_can_be_checked_before_they_are_rendered(fe, fo):
    """obs_src is a raw offset into the log-return table: the kernels do not range-check it.  For descriptors that
    did not come from this env object (another rank's gathered chunk, an uninitialised caller buffer) the debug check
    names the first bad one instead of letting the render fault."""
    from finenvs_amd._lib import FinEnvsNativeError

    _, env = _make(fe, fo, 50, 2, 8, 5, 40, 0.0, seed=2)
    D, L, _ = env.price_environments.shape
    src, pos = env.describe()
    env.check_descriptors(src)                      # its own descriptors are fine
    obs = env.render(src, pos, check=True)
    assert torch.equal(obs, env.reset())
    last_valid = torch.tensor([(D * L - 8) * 8], dtype=torch.int64, device="cuda")  # the last window of the last day
    env.check_descriptors(last_valid)
    for bad_value, why in ((-8, "negative"), (12, "not a row start"), ((D * L - 7) * 8, "window past the table"),
                           (1 << 40, "garbage")):
        bad = src.clone()
        bad[17] = bad_value
        with pytest.raises(FinEnvsNativeError, match=r"obs_src\[17\]"):
            env.check_descriptors(bad)
        with pytest.raises(FinEnvsNativeError, match="descriptors lie outside"):
            env.render(bad, pos, check=True)
    two = src.clone()
    two[40], two[3] = -1, -1
    with pytest.raises(FinEnvsNativeError, match=r"2 of 50 .*obs_src\[3\]"):
        env.check_descriptors(two)
    assert [int(x) for x in env.geometry()] == [D, L, 8, 2]
