"""K rollout steps as ONE hipGraph launch.

The reference's rollout loop (examples/time_series/PPO_LSTM_training_SPY.py:22-30) pays
Python + launch overhead per step: agent.step -> env.step -> agent.store.  At 64k envs the fused
env kernel takes ~35 us, the same order as that overhead.  ``GraphedRollout`` captures K
iterations of  policy(obs) -> env.step(actions) -> trajectory.store(...)  into a single
torch.cuda.CUDAGraph (a hipGraph on ROCm); ``run()`` replays it with one launch call.

Requirements: env.redraw == "device" and training or evaluate-free stepping (nothing on the
captured path may synchronise with the host), and K a multiple of env.obs_buffers so that the
observation a replay ends on is the buffer the next replay starts from.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from .trajectory import TrajectoryBuffer


class GraphedRollout:
    def __init__(self, env, policy: Callable[[torch.Tensor, int], torch.Tensor], num_steps: int,
                 trajectory: Optional[TrajectoryBuffer] = None, warmup: int = 2):
        if env.redraw != "device" and not env.evaluate:
            raise ValueError('GraphedRollout needs redraw="device" (redraw="torch" syncs on the done flag of the eval env)')
        # evaluate mode: the reference reads "have all envs terminated?" on the host every step (TSE:531); a captured
        # replay cannot, so that read moves to the end of the K steps (self.info, evaluate_returns) -- the kernel keeps the
        # per-env bookkeeping, and steps past an env's termination cannot change its return (TSE:526-528)
        if env.obs_buffers < 1 or num_steps % env.obs_buffers != 0:
            raise ValueError("num_steps must be a multiple of env.obs_buffers (>= 1)")
        if trajectory is not None and trajectory.T != num_steps:
            raise ValueError("trajectory buffer must hold exactly num_steps steps")
        self.env, self.policy, self.K, self.traj = env, policy, int(num_steps), trajectory
        self.rewards: List[torch.Tensor] = []
        self.dones: List[torch.Tensor] = []
        dev = env._dev
        # warm up on a side stream (allocator + lazy init), as torch.cuda.graph requires
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            self.obs = env.reset()
            if trajectory is not None and trajectory.has_states:
                trajectory.clear()
                trajectory.begin(env)  # row 0 of the very first chunk ...
                # ... and a copy in row K: every iteration (the captured one included) starts by carrying row K over
                trajectory.obs_src[self.K].copy_(trajectory.obs_src[0])
                trajectory.obs_pos[self.K].copy_(trajectory.obs_pos[0])
            for _ in range(warmup):
                self._iterate(record=False)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        env._sync_public_views()  # (pending in-place edits of handed-out views are applied now, not frozen into the graph as a copy)
        # THREAD-LOCAL capture mode: only this thread's calls are held to the capture rules.  In a torch.distributed (RCCL) job the
        # process group's watchdog thread polls its collectives' events (hipEventQuery) all the time; under the default GLOBAL mode
        # such a query from another thread while this one captures fails with "operation not permitted when stream is capturing",
        # the watchdog rethrows and std::terminate()s the process -- seen in one of ~10 rehearsals of bench.py's N > 1 path
        # (round 6), i.e. whenever a poll happened to fall into a capture.  Everything captured here is issued by this thread.
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._iterate(record=True)
        # what the graph froze: the kernel env.step dispatched to (f32 or promoted share arithmetic) and the pointers of the
        # env's bound state -- run() refuses to replay once either has changed
        self._captured_promoted = bool(env.shares_promoted)
        self._captured_epoch = env._binding_epoch
        # self.obs now names the static buffer holding the newest observation after each replay

    def _iterate(self, record: bool) -> None:
        env, traj = self.env, self.traj
        # the host read of evaluate mode is deferred only while THESE steps are issued (warm-up / capture); the env object
        # behaves as usual for anyone stepping it directly afterwards
        env._defer_evaluation_check = bool(env.evaluate)
        try:
            self._iterate_steps(env, traj, record)
        finally:
            env._defer_evaluation_check = False

    def _iterate_steps(self, env, traj, record: bool) -> None:
        if traj is not None:
            if traj.has_states:  # unconditionally, so that the captured graph contains it whatever t was at capture time
                traj._carry_state(traj._cur, traj._cur)
            traj.t = 0
        obs = self.obs
        rews, dones = [], []
        for k in range(self.K):
            actions = self.policy(obs, k)
            if traj is None:
                obs, rew, done, _ = env.step(actions)
            else:  # agent.store's fields are written by the step kernel itself: no store launch in the graph
                a_slot, r_slot, d_slot = traj.next_slot()
                obs, rew, done, _ = env.step(actions, rewards_out=r_slot, dones_out=d_slot, actions_out=a_slot,
                                             descriptors_out=traj.state_slot() if traj.has_states else None)
            rews.append(rew)
            dones.append(done)
        self.obs = obs
        if record:
            self.rewards, self.dones = rews, dones

    def run(self) -> torch.Tensor:
        """Replay the K captured steps; returns the newest observation (a static buffer).  In evaluate mode
        ``self.info`` then holds ``{"returns": ...}`` if every env has finished its episode by now (TSE:523-536), else {}."""
        env = self.env
        if env.shares_promoted != self._captured_promoted:
            raise RuntimeError("the env was stepped with float64 actions after this graph was captured: the reference computes its "
                               "commissions in f64 from then on (fe_env_step_promoted) while the captured launches run the f32 "
                               "arithmetic -- capture a new GraphedRollout")
        if env._binding_epoch != self._captured_epoch:
            raise RuntimeError("the env was resized (env_indices assigned with another length) after this graph was captured: its "
                               "launches point at the old state -- capture a new GraphedRollout")
        env._sync_public_views()  # in-place edits through handed-out views (promoted share mirrors, terminated_episodes) reach the replay
        self.graph.replay()
        # the replay advanced the env K steps without passing through env.step(): fused rollout objects sharing this env
        # must see their observation descriptors as stale (_FusedEvaluation._begin_run).  Callers that step through
        # the C ABI directly bypass this guard.
        self.env._generation += self.K
        if self.traj is not None:
            self.traj.t = self.K
        self.info = self.env.record_evaluation_metrics() if self.env.evaluate else {}
        return self.obs

    def evaluate_returns(self, max_steps: int = 1_000_000) -> torch.Tensor:
        """The reference's evaluation loop (examples/time_series/PPO_LSTM_testing_SPY.py:43-52) with ANY capturable
        policy -- e.g. the torch ``nn.LSTM`` actor itself -- K steps per graph replay and one host read per replay
        instead of one per step; returns the per-env episode returns."""
        if not self.env.evaluate:
            raise ValueError("evaluate_returns needs an env constructed with evaluate=True")
        steps = 0
        while steps < max_steps:
            self.run()
            steps += self.K
            if "returns" in self.info:
                return self.info["returns"]
        raise RuntimeError("episodes did not all terminate within max_steps")


class DoubleBufferedRollout:
    """The env batch as ``shards`` contiguous shards (``rank = i, world_size = shards`` on ONE device: the very same envs,
    env n -> day n mod D, the evaluation env in the last shard), one ``GraphedRollout`` per shard, each replayed on its
    own HIP stream.  Envs are independent, so when the policy acts per env (the reference's actors do) the shards need
    not wait for each other: while one shard's launch boundary, first-tile chain and policy run, the other shard's
    observation stream keeps HBM busy -- the ~4 us per step that a single 30 us launch cannot hide (DESIGN.md section
    5: 29.5 -> 26.3 us per step of all 65 536 envs; two shards is the sweet spot, four gain nothing).

    ``make_env(rank, world_size)`` must return a ``TimeSeriesEnv`` built with exactly those ``rank`` / ``world_size``
    (and ``redraw="device"``, ``obs_buffers >= 1``: what ``GraphedRollout`` needs).  ``policy(obs, k)`` is called with
    ONE shard's observation.  ``rewards`` / ``dones`` are per-shard lists of the K per-step tensors of the last replay
    (``joined_rewards()`` / ``joined_dones()``: (K, N) in global env order).

    Stream semantics: the shards run on their own streams and must be allowed to DRIFT APART -- if they are brought
    together after every replay they start every replay in lock-step, both in their start-up at the same time, and
    the overlap is gone (measured: 0.99 x instead of 1.12 x).  So ``run()`` only forks from the caller's stream the
    first time (and after a ``join()``); it does not make the caller's stream wait.  Call ``join()`` before consuming
    ``obs`` / ``rewards`` / ``dones`` / env state on the current stream, typically once per chunk of replays."""

    def __init__(self, make_env: Callable[[int, int], object], policy: Callable[[torch.Tensor, int], torch.Tensor],
                 num_steps: int, shards: int = 2, stagger: bool = True):
        if shards < 1:
            raise ValueError("shards must be >= 1")
        self.envs = [make_env(r, shards) for r in range(shards)]
        for r, env in enumerate(self.envs):
            if env.rank != r or env.world_size != shards:
                raise ValueError("make_env(rank, world_size) must pass both on to TimeSeriesEnv")
            if env._dev != self.envs[0]._dev:
                raise ValueError("all shards live on one device (for several GPUs use one process per GPU)")
        self._dev = self.envs[0]._dev
        self.streams = [torch.cuda.Stream(device=self._dev) for _ in self.envs]
        self.rolls = [GraphedRollout(env, policy, num_steps) for env in self.envs]
        self.K = int(num_steps)
        self.num_envs = sum(env.num_envs for env in self.envs)
        self._forked = False
        # Stagger: shard i starts i / shards of a step late, so that the shards run in anti-phase from the first replay
        # on (two equal graphs started together would march in lock-step, both in their start-up at the same time).
        # One shard's step time is measured here (outputs unaffected: the replays below are ordinary rollout steps).
        self._stagger_cycles = 0
        if stagger and shards > 1 and hasattr(torch.cuda, "_sleep"):  # (a private torch helper: without it, no stagger)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            probe = GraphedRollout(make_env(0, shards), policy, num_steps)  # a throw-away twin of shard 0
            probe.run()
            e0.record()
            probe.run()
            e1.record()
            e1.synchronize()
            step_ms = e0.elapsed_time(e1) / self.K
            # torch.cuda._sleep counts device clock ticks of unknown rate: calibrate it with events
            torch.cuda._sleep(1_000_000)
            e0.record()
            torch.cuda._sleep(4_000_000)
            e1.record()
            e1.synchronize()
            ticks_per_ms = 4_000_000 / max(e0.elapsed_time(e1), 1e-3)
            self._stagger_cycles = int(step_ms / shards * ticks_per_ms)
            del probe

    def run(self) -> None:
        """One replay (K steps) of every shard, asynchronously on the shards' streams."""
        if not self._forked:  # whatever the caller queued so far (e.g. new policy weights) comes first
            cur = torch.cuda.current_stream(self._dev)
            for i, s in enumerate(self.streams):
                s.wait_stream(cur)
                if i and self._stagger_cycles:
                    with torch.cuda.stream(s):
                        torch.cuda._sleep(i * self._stagger_cycles)
            self._forked = True
        for roll, s in zip(self.rolls, self.streams):
            with torch.cuda.stream(s):
                roll.run()

    def join(self) -> List[torch.Tensor]:
        """Make the current stream wait for all shards; returns the shards' newest observations."""
        cur = torch.cuda.current_stream(self._dev)
        for s in self.streams:
            cur.wait_stream(s)
        self._forked = False
        return [roll.obs for roll in self.rolls]

    @property
    def rewards(self) -> List[List[torch.Tensor]]:
        return [roll.rewards for roll in self.rolls]

    @property
    def dones(self) -> List[List[torch.Tensor]]:
        return [roll.dones for roll in self.rolls]

    def joined_rewards(self) -> torch.Tensor:
        return torch.cat([torch.stack(r) for r in self.rewards], dim=1)

    def joined_dones(self) -> torch.Tensor:
        return torch.cat([torch.stack(d) for d in self.dones], dim=1)


class _FusedEvaluation:
    """Shared by the fused rollouts (each has ``env``, ``obs_src (N,)``, ``obs_pos (N, A)`` and ``run``): descriptor
    upkeep, observation on demand, and the reference's evaluation loop on the device."""

    def sync_from_env(self) -> None:
        """Point the descriptors at the observation ``env.reset()`` would render now."""
        from . import _lib

        self._check_epoch()
        _lib.check(self.env._lib.fe_env_describe(self.env._handle, self.obs_src.data_ptr(), self.obs_pos.data_ptr(),
                                                 self.env._stream()))
        self._seen_generation = self.env._generation

    def _check_epoch(self) -> None:
        """This object's descriptor arrays are sized for the env as it was when the object was made (its constructor's
        ``sync_from_env`` records the env's binding epoch): after a resize they are the wrong length for every kernel."""
        if getattr(self, "_epoch", None) is None:
            self._epoch = self.env._binding_epoch
        elif self._epoch != self.env._binding_epoch:
            raise RuntimeError("the env was resized (env_indices assigned with another length) after this rollout object was made: "
                               "construct a new one")

    def _begin_run(self) -> None:
        """A fused rollout keeps its OWN observation descriptors (what its policy sees next) but reads the account state
        from the env's shared arrays.  If anything else advanced the env since this object last looked -- ``env.step``,
        another rollout object's ``run`` -- the two no longer belong together and the policy would act on a stale
        observation while the accounting uses the current state: refuse instead of running on silently."""
        self._check_epoch()
        self.env._sync_public_views()
        if getattr(self.env, "shares_promoted", False):
            raise RuntimeError("this env was stepped with float64 actions: the reference computes its commissions in f64 from then on "
                               "(fe_env_step_promoted); the fused rollouts run the f32 arithmetic only")
        if getattr(self, "_seen_generation", None) != self.env._generation:
            raise RuntimeError("the env was stepped by someone else since this rollout object's last run(): its observation "
                               "descriptors are stale.  Call sync_from_env() (= look at the state as env.reset() renders it) "
                               "or drive the env through ONE rollout object.")

    def _end_run(self) -> None:
        self.env._generation += 1
        self._seen_generation = self.env._generation

    def observation(self) -> torch.Tensor:
        """The (N, W, 5A) observation the next policy evaluation will see."""
        from . import _lib

        self._check_epoch()
        obs = self.env._next_obs()
        _lib.check(self.env._lib.fe_env_render(self.env._handle, self.obs_src.data_ptr(), self.obs_pos.data_ptr(),
                                               obs.data_ptr(), self.env._stream()))
        return obs


    def evaluate_returns(self, chunk: int = 64, max_steps: int = 1_000_000) -> torch.Tensor:
        """The reference's evaluation loop (examples/time_series/PPO_LSTM_testing_SPY.py:43-52): step the
        evaluate-mode env with the actor until ``info["returns"]`` would appear, i.e. every env has finished one
        episode; returns those per-env episode returns.  Runs ``chunk`` steps per launch; steps past an env's
        termination cannot change its return (TSE:526-528 zeroes their rewards).  (The reference feeds the actor's
        output unclamped; the env's share-change clamp, TSE:298-302, makes that equal to the clamped action.)"""
        env = self.env
        if not env.evaluate:
            raise ValueError("evaluate_returns needs an env constructed with evaluate=True")
        steps = 0
        while steps < max_steps:
            self.run(chunk, record_actions=False)
            steps += chunk
            if int(env._counters[0].item()) == env.num_envs:
                returns = env.episode_returns.clone()
                env.reset_evaluation_metrics()
                return returns
        raise RuntimeError("episodes did not all terminate within max_steps")


class FusedLinearRollout(_FusedEvaluation):
    """K env steps per launch with an in-kernel linear policy (SURVEY.md 8f.2, C ABI
    ``fe_env_rollout_linear``): the whole loop

        states = env.reset()
        for k in range(K): actions = policy(states); states, r, d, _ = env.step(actions)

    (examples/time_series/PPO_LSTM_training_SPY.py:22-28) inside one kernel, for the policy
    ``clamp(bias + <obs window of the asset, weights (W, 5)>, -1, 1)``.  Observations are never
    written to HBM during the rollout; ``observation()`` materialises the current one on demand.
    State, evaluate-mode metrics and EpisodeStats advance exactly as K calls of ``env.step``."""

    def __init__(self, env, weights: torch.Tensor, bias: float = 0.0, form: str = "window"):
        """``form="window"``: the policy re-reads the window every step and sums all 5 features in
        observation order.  ``form="table"``: the log-return part is precomputed once per weight
        update as an indicator table (one number per day, window start and asset), a step then
        costs two 8-byte lookups per account -- an order of magnitude faster; its sum is split
        (table + position * sum of position weights), so its actions can differ from the window
        form's in the last bit."""
        W = env.num_intervals
        if tuple(weights.shape) != (W, 5):
            raise ValueError(f"weights must be ({W}, 5): one weight per window row and feature")
        if form not in ("window", "table"):
            raise ValueError('form must be "window" or "table"')
        self.form = form
        if env.redraw != "device" and not env.evaluate:
            raise ValueError('the fused rollout needs redraw="device" (or evaluate mode): no host in the loop')
        self.env = env
        self.weights = weights.detach().to(device=env._dev, dtype=torch.float64).contiguous()
        self.bias = float(bias)
        self.obs_src = torch.empty((env.num_envs,), dtype=torch.int64, device=env._dev)
        self.obs_pos = torch.empty((env.num_envs, env.num_assets), dtype=torch.float64, device=env._dev)
        self.table = None
        if form == "table":
            D, L, _ = env.price_environments.shape
            self.table = torch.empty((D, L, env.num_assets), dtype=torch.float64, device=env._dev)
            self._wsum = torch.empty((1,), dtype=torch.float64, device=env._dev)
            self.set_weights(self.weights, self.bias)
        self.sync_from_env()

    def set_weights(self, weights: torch.Tensor, bias: float = None) -> None:
        """New policy parameters (the table form rebuilds its indicator table: one launch)."""
        from . import _lib

        self.weights = weights.detach().to(device=self.env._dev, dtype=torch.float64).contiguous()
        if bias is not None:
            self.bias = float(bias)
        if self.form == "table":
            _lib.check(self.env._lib.fe_policy_table(self.env._handle, self.weights.data_ptr(), self.table.data_ptr(),
                                                     self._wsum.data_ptr(), self.env._stream()))

    def run(self, num_steps: int, record_actions: bool = True):
        """Returns (actions (K, N, A) f32 or None, rewards (K, N) f64, dones (K, N) int32)."""
        from . import _lib

        env, K = self.env, int(num_steps)
        N, A = env.num_envs, env.num_assets
        self._begin_run()
        actions = torch.empty((K, N, A), dtype=torch.float32, device=env._dev) if record_actions else None
        rewards = torch.empty((K, N), dtype=torch.float64, device=env._dev)
        dones = torch.empty((K, N), dtype=torch.int32, device=env._dev)
        if self.form == "table":
            _lib.check(env._lib.fe_env_rollout_table(
                env._handle, self.table.data_ptr(), self._wsum.data_ptr(), self.bias, K, self.obs_src.data_ptr(),
                self.obs_pos.data_ptr(), actions.data_ptr() if record_actions else None, rewards.data_ptr(),
                dones.data_ptr(), env._stream()))
        else:
            _lib.check(env._lib.fe_env_rollout_linear(
                env._handle, self.weights.data_ptr(), self.bias, K, self.obs_src.data_ptr(), self.obs_pos.data_ptr(),
                actions.data_ptr() if record_actions else None, rewards.data_ptr(), dones.data_ptr(), env._stream()))
        self._end_run()
        return actions, rewards, dones


class FusedMLPRollout(_FusedEvaluation):
    """K env steps per launch with an in-kernel two-layer perceptron policy on the flattened observation window of
    every (env, asset) pair (SURVEY.md 8f.2 "linear/MLP head"; C ABI ``fe_env_rollout_mlp``):

        actions = clamp(W2 . act(W1^T . flatten(states.float()) + b1) + b2, -1, 1)

    i.e. ``nn.Sequential(nn.Flatten(), nn.Linear(5W, H), act, nn.Linear(H, 1))`` of the reference's MLP networks
    (finenvs/agents/networks/multilayer_perceptron.py:17-25, default ELU) applied per asset.  The first layer is a
    dense (pairs x 5W x H) contraction on the MFMA units (f32 in / f32 accumulate); observations are never written
    to HBM during the rollout.  ``W1`` is (5W, H) with rows in observation order (row 5j+c), ``b1``/``W2`` (H,),
    ``H`` in {32, 64, 128}."""

    ACTIVATIONS = {"elu": 0, "relu": 1, "tanh": 2}

    def __init__(self, env, W1: torch.Tensor, b1: torch.Tensor, W2: torch.Tensor, b2: float = 0.0, activation: str = "elu"):
        W = env.num_intervals
        if W1.dim() != 2 or W1.shape[0] != 5 * W:
            raise ValueError(f"W1 must be ({5 * W}, H): one row per flattened observation element")
        H = int(W1.shape[1])
        if H not in (32, 64, 128):
            raise ValueError("H must be 32, 64 or 128")
        if activation not in self.ACTIVATIONS:
            raise ValueError(f"activation must be one of {sorted(self.ACTIVATIONS)}")
        if env.redraw != "device" and not env.evaluate:
            raise ValueError('the fused rollout needs redraw="device" (or evaluate mode): no host in the loop')
        self.env, self.H, self.act = env, H, self.ACTIVATIONS[activation]
        dev = env._dev
        self.obs_src = torch.empty((env.num_envs,), dtype=torch.int64, device=dev)
        self.obs_pos = torch.empty((env.num_envs, env.num_assets), dtype=torch.float64, device=dev)
        # the f32 copy of the log-return table the first layer streams from (what states.float() would hold)
        self._lr32 = getattr(env, "_log_return_f32", None)
        if self._lr32 is None:
            self._lr32 = env.log_return_environments.float().contiguous()
        self.set_weights(W1, b1, W2, b2)
        self.sync_from_env()

    def set_weights(self, W1: torch.Tensor, b1: torch.Tensor, W2: torch.Tensor, b2: float) -> None:
        W, H, dev = self.env.num_intervals, self.H, self.env._dev
        w = W1.detach().to(dtype=torch.float32, device="cpu").reshape(W, 5, H)
        self.w1t = w[:, :4, :].reshape(4 * W, H).t().contiguous().to(dev)
        wpos = torch.zeros((H,), dtype=torch.float32)
        for j in range(W):  # sequential f32 sum, j ascending: part of the contract (oracle: mlp_pack)
            wpos = wpos + w[j, 4, :]
        self.wpos = wpos.to(dev)
        self.b1 = b1.detach().to(dtype=torch.float32, device=dev).reshape(H).contiguous()
        self.w2 = W2.detach().to(dtype=torch.float32, device=dev).reshape(H).contiguous()
        self.b2 = float(b2)

    def run(self, num_steps: int, record_actions: bool = True):
        """Returns (actions (K, N, A) f32 or None, rewards (K, N) f64, dones (K, N) int32)."""
        from . import _lib

        env, K = self.env, int(num_steps)
        N, A = env.num_envs, env.num_assets
        actions = torch.empty((K, N, A), dtype=torch.float32, device=env._dev) if record_actions else None
        rewards = torch.empty((K, N), dtype=torch.float64, device=env._dev)
        dones = torch.empty((K, N), dtype=torch.int32, device=env._dev)
        self._begin_run()
        _lib.check(env._lib.fe_env_rollout_mlp(
            env._handle, self._lr32.data_ptr(), self.w1t.data_ptr(), self.wpos.data_ptr(), self.b1.data_ptr(),
            self.w2.data_ptr(), self.b2, self.H, self.act, K, self.obs_src.data_ptr(), self.obs_pos.data_ptr(),
            actions.data_ptr() if record_actions else None, rewards.data_ptr(), dones.data_ptr(), env._stream()))
        self._end_run()
        return actions, rewards, dones


def lstm_row_order(H: int) -> torch.Tensor:
    """Row of ``weight_hh_l0`` / ``weight_ih_l0`` (torch order: gate * H + unit, gates i, f, g, o) that packed row
    R = 32*mt + 8*b + 4*half + gate holds: the hidden unit of R is 8*mt + 4*half + b, so that one accumulator lane of
    the kernel's 32 x 32 MFMA tile carries all four gates of four units (include/finenvs_amd.h, fe_env_rollout_lstm)."""
    R = torch.arange(4 * H)
    mt, rho = R // 32, R % 32
    gate, half, b = rho % 4, (rho % 8) // 4, rho // 8
    return gate * H + 8 * mt + 4 * half + b


class FusedLSTMRollout(_FusedEvaluation):
    """K env steps per launch with the LSTM actor of the reference's time-series scripts evaluated in the kernel
    (C ABI ``fe_env_rollout_lstm``):

        actions = tanh(Linear(H, 1)(LSTM(5, H)(states.float())[:, -1, :]))         per (env, asset) pair

    i.e. ``LSTMNetwork((5, H, 1), sequence_length=W, output_activation=nn.Tanh)`` of
    finenvs/agents/networks/lstm.py:28-57 / finenvs/agents/PPO/continuous_actor.py:104-126 -- the network
    examples/time_series/PPO_LSTM_testing_SPY.py:43-52 steps the evaluate-mode env with; ``evaluate_returns()`` is
    that loop.  The gate contractions run on the MFMA units in f32, observations are never written to HBM.
    Parameters are ``nn.LSTM``'s ``weight_ih_l0 (4H, 5)``, ``weight_hh_l0 (4H, H)``, ``bias_ih_l0``, ``bias_hh_l0``
    and the output layer's ``weight (1, H)`` / ``bias``; ``H`` in {32, 64, 128} (recurrent weights in registers) or
    {256, 512, 1024} (streamed from L2; the reference example trains ``hidden_dim=1024``).  The large sizes are a
    throughput kernel (one 32-pair tile per workgroup walks the whole matrix); below ``SPLIT_BELOW_PAIRS`` (env, asset)
    pairs -- an evaluation over a few hundred trading days -- ``run`` issues one launch per LSTM time step instead."""

    OUTPUT_ACTIVATIONS = {"tanh": 0, "clamp": 1, "none": 2}  # "none": a critic (forward() only; an action needs bounds)
    # H >= 256 below this many (env, asset) pairs: one launch per LSTM time step with the gate-row tiles spread over the
    # whole GPU (fe_env_rollout_lstm_split) instead of one tile per CU for a whole step; ``self.split`` = True / False
    # overrides the choice (measured cross-over: profiles/r02_microbench/lstm_split.txt)
    SPLIT_BELOW_PAIRS = 5120

    def __init__(self, env, weight_ih: torch.Tensor, weight_hh: torch.Tensor, bias_ih: torch.Tensor, bias_hh: torch.Tensor,
                 weight_out: torch.Tensor, bias_out: float = 0.0, output_activation: str = "tanh"):
        if weight_hh.dim() != 2 or weight_hh.shape[0] != 4 * weight_hh.shape[1]:
            raise ValueError("weight_hh must be (4H, H)")
        H = int(weight_hh.shape[1])
        if H not in (32, 64, 128, 256, 512, 1024):
            raise ValueError("H must be 32, 64, 128 (weights in registers) or 256, 512, 1024 (weights streamed from L2)")
        if tuple(weight_ih.shape) != (4 * H, 5):
            raise ValueError(f"weight_ih must be ({4 * H}, 5): the four log-returns and the position feature")
        if output_activation not in self.OUTPUT_ACTIVATIONS:
            raise ValueError(f"output_activation must be one of {sorted(self.OUTPUT_ACTIVATIONS)}")
        if env.redraw != "device" and not env.evaluate:
            raise ValueError('the fused rollout needs redraw="device" (or evaluate mode): no host in the loop')
        self.env, self.H, self.out_act = env, H, self.OUTPUT_ACTIVATIONS[output_activation]
        self.split, self._workspace = None, None
        dev = env._dev
        self.obs_src = torch.empty((env.num_envs,), dtype=torch.int64, device=dev)
        self.obs_pos = torch.empty((env.num_envs, env.num_assets), dtype=torch.float64, device=dev)
        self._lr32 = getattr(env, "_log_return_f32", None)
        if self._lr32 is None:
            self._lr32 = env.log_return_environments.float().contiguous()
        self.set_weights(weight_ih, weight_hh, bias_ih, bias_hh, weight_out, bias_out)
        self.sync_from_env()

    @classmethod
    def from_modules(cls, env, lstm: "torch.nn.LSTM", linear: "torch.nn.Linear", output_activation: str = "tanh"):
        """From the two modules of the reference's LSTMNetwork (``.lstm`` and ``.last_layer[0]``)."""
        if lstm.num_layers != 1 or lstm.bidirectional or lstm.input_size != 5 or linear.out_features != 1:
            raise ValueError("expected nn.LSTM(5, H, num_layers=1) and nn.Linear(H, 1)")
        return cls(env, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0, linear.weight,
                   float(linear.bias.detach()), output_activation)

    def set_weights(self, weight_ih, weight_hh, bias_ih, bias_hh, weight_out, bias_out: float) -> None:
        H, dev = self.H, self.env._dev
        f32 = dict(dtype=torch.float32, device="cpu")
        order = lstm_row_order(H)
        w_ih, w_hh = weight_ih.detach().to(**f32), weight_hh.detach().to(**f32)
        for name, t in (("weight_ih", w_ih), ("weight_hh", w_hh), ("bias_ih", bias_ih), ("bias_hh", bias_hh), ("weight_out", weight_out)):
            if not bool(torch.isfinite(t.detach()).all()):  # the kernel's activations do not propagate NaN
                raise ValueError(f"{name} has non-finite entries")
        bias = bias_ih.detach().to(**f32).reshape(4 * H) + bias_hh.detach().to(**f32).reshape(4 * H)  # one f32 add
        wx = torch.zeros((4 * H, 8), dtype=torch.float32)
        wx[:, :5] = w_ih[order]
        wx[:, 5] = bias[order]
        whh = w_hh[order].contiguous()  # packed row order
        if H > 128:  # fragment-major for the streaming kernel: [row tile][k group][lane = (row & 31) + 32 * k half][4]
            whh = whh.reshape(4 * H // 32, 32, H // 8, 2, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(4 * H, H)
        self.whh = whh.to(dev)
        self.wx = wx.to(dev)
        self.wout = weight_out.detach().to(dtype=torch.float32, device=dev).reshape(H).contiguous()
        self.bout = float(bias_out)

    def forward(self, obs_src: torch.Tensor, obs_pos: torch.Tensor, out: Optional[torch.Tensor] = None,
                check: bool = False) -> torch.Tensor:
        """The head evaluated on ANY B observation descriptors (``obs_src (B,)`` int64, ``obs_pos (B, A)`` float64 --
        rows of a ``TrajectoryBuffer(states=True)``, gathered ones included) without stepping the env and without
        materialising the observations: ``(B, A)`` float32.  With ``output_activation="none"`` this is the critic of the
        reference's PPO (finenvs/agents/PPO/critic.py, CriticLSTM): the values of all K + 1 states of a chunk in one
        launch, e.g. ``critic.forward(traj.obs_src, traj.obs_pos).reshape(K + 1, N)``.  ``check=True`` validates the
        descriptors first (``env.check_descriptors``: for descriptors that came from another rank or a caller's buffer)."""
        from . import _lib

        env, A = self.env, self.env.num_assets
        B = int(obs_src.numel())
        src = obs_src.reshape(B).to(device=env._dev, dtype=torch.int64).contiguous()
        pos = obs_pos.reshape(B, A).to(device=env._dev, dtype=torch.float64).contiguous()
        if check:
            env.check_descriptors(src)
        if out is None:
            out = torch.empty((B, A), dtype=torch.float32, device=env._dev)
        elif out.dtype is not torch.float32 or out.numel() != B * A or not out.is_contiguous() or out.device != env._dev:
            raise ValueError(f"out must be a contiguous float32 tensor of {B} x {A} elements on {env._dev}")
        if B:
            _lib.check(env._lib.fe_lstm_forward(
                env._handle, self._lr32.data_ptr(), self.whh.data_ptr(), self.wx.data_ptr(), self.wout.data_ptr(), self.bout,
                self.H, self.out_act, src.data_ptr(), pos.data_ptr(), B, out.data_ptr(), env._stream()))
        return out

    def run(self, num_steps: int, record_actions: bool = True, noise: Optional[torch.Tensor] = None,
            std: Optional[float] = None, record_means: bool = False, trajectory=None):
        """Returns (actions (K, N, A) f32 or None, rewards (K, N) f64, dones (K, N) int32).

        Training rollouts (``agent.step`` of finenvs/agents/PPO/PPO_agent.py:98-108): pass ``noise`` -- (K, N, A) f32
        standard-normal draws from the caller's generator -- and ``std = exp(log_standard_deviation)``; the action is
        ``clamp(mean + std * noise, -1, 1)``, the eval env of a training-mode env acts on the mean.  (An evaluate-mode
        env has no evaluation env, so there every env samples; the reference's ``agent.step`` overwrites the LAST row with
        the mean whatever the env's mode, PPO_agent.py:104-106 -- a caller who wants that passes ``noise[:, -1] = 0``,
        which is the same action bit for bit.)  ``record_means``
        keeps the means in ``self.means`` ((K, N, A), what ``log_prob`` needs).  ``trajectory``: an empty
        ``TrajectoryBuffer(K, N, A, states=True)`` without capacity padding -- the kernel writes actions, rewards,
        dones and the K + 1 state descriptors straight into its chunk (``agent.store`` for K steps at once; the
        returned tensors are then views of it)."""
        from . import _lib

        env, K = self.env, int(num_steps)
        N, A = env.num_envs, env.num_assets
        dev = env._dev
        self._begin_run()
        src_out = pos_out = None
        if trajectory is not None:
            tr = trajectory
            if not (tr.has_states and tr.T == K and tr.N == N and tr.C == N and tr.A == A and len(tr) == 0 and tr.device == dev):
                raise ValueError("trajectory must be an empty TrajectoryBuffer(K, N, A, states=True) on the env's device "
                                 "without capacity padding")
            actions, rewards, dones = tr.actions, tr.rewards, tr.dones
            src_out, pos_out = tr.obs_src, tr.obs_pos
        else:
            actions = torch.empty((K, N, A), dtype=torch.float32, device=dev) if record_actions else None
            rewards = torch.empty((K, N), dtype=torch.float64, device=dev)
            dones = torch.empty((K, N), dtype=torch.int32, device=dev)
        if noise is not None:
            if std is None or not float(std) >= 0.0:
                raise ValueError("noise needs std >= 0 (= exp(log_standard_deviation))")
            if noise.dtype is not torch.float32 or noise.numel() != K * N * A or noise.device != dev:
                raise ValueError(f"noise must be ({K}, {N}, {A}) float32 on {dev}")
            noise = noise.contiguous()
        self.means = torch.empty((K, N, A), dtype=torch.float32, device=dev) if record_means else None
        args = (env._handle, self._lr32.data_ptr(), self.whh.data_ptr(), self.wx.data_ptr(), self.wout.data_ptr(), self.bout,
                self.H, self.out_act, K, self.obs_src.data_ptr(), self.obs_pos.data_ptr(),
                noise.data_ptr() if noise is not None else None, float(std) if noise is not None else 0.0,
                actions.data_ptr() if actions is not None else None, self.means.data_ptr() if record_means else None,
                rewards.data_ptr(), dones.data_ptr(), src_out.data_ptr() if src_out is not None else None,
                pos_out.data_ptr() if pos_out is not None else None)
        use_split = self.split if self.split is not None else (self.H > 128 and N * A < self.SPLIT_BELOW_PAIRS)
        if use_split:
            if self.H <= 128:
                raise ValueError("split=True is for H in {256, 512, 1024}")
            if self._workspace is None:
                n = int(env._lib.fe_lstm_split_workspace_floats(self.H, N * A))
                self._workspace = torch.empty((n,), dtype=torch.float32, device=dev)
            _lib.check(env._lib.fe_env_rollout_lstm_split(*args, self._workspace.data_ptr(), env._stream()))
        else:
            _lib.check(env._lib.fe_env_rollout_lstm(*args, env._stream()))
        self._end_run()
        if trajectory is not None:
            trajectory.mark_filled(K)
        return actions, rewards, dones

