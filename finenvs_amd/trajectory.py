"""Device-resident trajectory buffer + RCCL all-gather.

Replaces the compact-field half of the reference's PPO buffer
(finenvs/agents/PPO/buffer.py): ``store`` is one slot write into preallocated
time-major tensors instead of a ``torch.cat`` per step (buffer.py:33-56, O(T^2)
bytes moved), and ``returns_and_advantages`` is one reverse-scan kernel per env
instead of a Python loop over T (buffer.py:80-100), with the reference's dtype
discipline (f32 discount factor, f64 carry, f32 results).

Multi-GPU: envs are sharded contiguously, one process per GPU; the only
exchange is ``all_gather`` of these compact fields -- actions (T, n, A) f32,
rewards (T, n) f64, dones (T, n) i32 -- once per T-step chunk, as ONE collective
over a single packed byte buffer (RCCL over xGMI when the process group is
"nccl").  Observations are never gathered: they stay sharded with their learner.

``states=True`` adds the reference buffer's ``states`` field (buffer.py:46, one (N, 1, W, 5) f64 ``torch.cat`` per
step) as DESCRIPTORS: per env-step the window offset ``obs_src`` (i64) and the position features ``obs_pos`` (A f64)
-- 8 + 8A bytes instead of 40WA (16 B against 2 560 B at W = 64; 248 B against 153 600 B for 30 assets at W = 128,
where T steps of a million envs would not fit any memory).  ``states`` / ``minibatch_states`` render them through
``env.render`` when the learner asks (PPO_agent.py:175-188), and because the tables are replicated the gathered
descriptors of OTHER ranks render too: the all-gather then carries the complete PPO sample, states included.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class TrajectoryBuffer:
    def __init__(self, num_steps: int, num_envs: int, num_assets: int = 1, device: str = "cuda:0",
                 host_rehearsal: bool = False, capacity: Optional[int] = None, states: bool = False):
        """``capacity`` (>= num_envs, same on every rank) sizes the env axis of the buffers, so that ranks
        owning shards of different sizes (N not divisible by the world size) can still exchange chunks with
        one all-gather; ``join_shards`` drops the padding again."""
        self.T, self.N, self.A = int(num_steps), int(num_envs), int(num_assets)
        self.C = int(capacity) if capacity is not None else self.N
        if self.C < self.N:
            raise ValueError("capacity must be >= num_envs")
        self.device = torch.device(device)
        if self.device.type != "cuda" and not host_rehearsal:
            raise RuntimeError("TrajectoryBuffer lives in HBM; host tensors are accepted only with host_rehearsal=True, "
                               "which exists to rehearse the all-gather plumbing over gloo (no kernels run there)")
        self.has_states = bool(states)
        T, N, A = self.T, self.C, self.A  # the env axis is laid out with `capacity` slots
        # one allocation per chunk, typed views, 8-byte fields first:
        #   [rewards f64 | obs_pos f64, obs_src i64 (states=True) | actions f32 | dones i32];
        # two chunks so that a chunk can be in flight on the collective stream while the next fills
        # state descriptors have T + 1 rows: row t is the observation the policy saw at step t, row T the one the
        # last step returned (the `current_states` PPO_agent.py:171 bootstraps its values from)
        fields = [("rewards", torch.float64, (), T)]
        if self.has_states:
            fields += [("obs_pos", torch.float64, (A,), T + 1), ("obs_src", torch.int64, (), T + 1)]
        fields += [("actions", torch.float32, (A,), T), ("dones", torch.int32, (), T)]
        self._layout, off = {}, 0
        for name, dt, trail, rows in fields:
            nb = rows * N * (A if trail else 1) * torch.empty((), dtype=dt).element_size()
            self._layout[name] = (off, nb, dt, trail, rows)
            off += nb
        self._nbytes = off
        self._chunks = [torch.zeros((self._nbytes,), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._views = [self._typed(c) for c in self._chunks]
        # per-slot views, built once (tensor indexing costs microseconds of host time per call)
        n = self.N
        self._slots = [[(v[0][t, :n], v[1][t, :n], v[2][t, :n]) for t in range(T)] for v in self._views]
        self._state_views = [self._typed_states(c) for c in self._chunks] if self.has_states else None
        self._state_slots = ([[(v[0][t, :n], v[1][t, :n]) for t in range(T + 1)] for v in self._state_views]
                             if self.has_states else None)
        self._begun = False            # states=True: row 0 of the first chunk has been set (begin / a fused rollout)
        self._pending = [None, None]   # outstanding collective per chunk
        self._deferred = None          # (chunk, group) handed over with all_gather_async(defer=True), not started yet
        self._gathered = [None, None]  # its output buffer
        self._cur = 0
        self.t = 0
        self._native = self.device.type == "cuda"
        self._lib = _lib.load() if self._native else None

    def _field(self, packed: torch.Tensor, name: str, lead: Tuple[int, ...] = ()) -> torch.Tensor:
        N = self.C
        off, nb, dt, trail, rows = self._layout[name]
        if lead:
            flat = packed.reshape(-1, self._nbytes)
            return flat[:, off:off + nb].contiguous().view(dt).view(flat.shape[0], rows, N, *trail)
        return packed[off:off + nb].view(dt).view(rows, N, *trail)

    def _typed(self, packed: torch.Tensor, lead: Tuple[int, ...] = ()):
        """(actions, rewards, dones) views of a packed chunk (of G packed chunks when ``lead``)."""
        return tuple(self._field(packed, k, lead) for k in ("actions", "rewards", "dones"))

    def _typed_states(self, packed: torch.Tensor, lead: Tuple[int, ...] = ()):
        """(obs_src, obs_pos) views of a packed chunk."""
        return tuple(self._field(packed, k, lead) for k in ("obs_src", "obs_pos"))

    # the chunk being filled
    @property
    def actions(self) -> torch.Tensor:
        return self._views[self._cur][0][:, : self.N]

    @property
    def rewards(self) -> torch.Tensor:
        return self._views[self._cur][1][:, : self.N]

    @property
    def dones(self) -> torch.Tensor:
        return self._views[self._cur][2][:, : self.N]

    @property
    def obs_src(self) -> torch.Tensor:
        """(T + 1, N) int64 window offsets of the stored states (``states=True``); row T = the last returned one."""
        return self._state_views[self._cur][0][:, : self.N]

    @property
    def obs_pos(self) -> torch.Tensor:
        """(T + 1, N, A) float64 position features of the stored states (``states=True``)."""
        return self._state_views[self._cur][1][:, : self.N]

    @property
    def _packed(self) -> torch.Tensor:
        return self._chunks[self._cur]

    def _stream(self) -> int:
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if _raw_stream is not None:
            return _raw_stream(idx)
        return torch.cuda.current_stream(self.device).cuda_stream

    def __len__(self) -> int:
        return self.t

    def full(self) -> bool:
        return self.t >= self.T

    def clear(self) -> None:
        if self.t == self.T:
            self._carry_state(self._cur, self._cur)
        self.t = 0

    def store(self, actions: torch.Tensor, rewards: torch.Tensor, dones: torch.Tensor) -> None:
        if self.t >= self.T:
            raise IndexError("trajectory buffer is full; call clear()")
        if self._native:
            if actions.dtype is not torch.float32:
                actions = actions.float()
            st = self._stream()
            sa, sr, sd = self._slots[self._cur][self.t]  # slot t starts at row t of the capacity-strided buffers
            _lib.check(self._lib.fe_traj_store(
                0, self.N, self.A, actions.contiguous().data_ptr(), rewards.data_ptr(), dones.data_ptr(),
                sa.data_ptr(), sr.data_ptr(), sd.data_ptr(), st))
        else:  # host tensors: only the gloo rehearsal of the collective uses this
            self.actions[self.t].copy_(actions.reshape(self.N, self.A))
            self.rewards[self.t].copy_(rewards)
            self.dones[self.t].copy_(dones)
        self.t += 1

    def begin(self, env_or_descriptors) -> None:
        """Row 0 of the state descriptors: the observation the policy is looking at when the chunk starts.  Pass the
        env (``last_observation_descriptors()``: the current state after ``reset()``, else what the last step
        recorded) or a ``(obs_src (N,), obs_pos (N, A))`` pair.  When a chunk follows another (``clear()`` /
        ``all_gather_async()``) row 0 is carried over from the previous chunk's row T automatically."""
        if not self.has_states:
            raise RuntimeError("this TrajectoryBuffer was built without states=True")
        if self.t != 0:
            raise RuntimeError("begin() belongs at the start of a chunk")
        src, pos = self._state_slots[self._cur][0]
        d = env_or_descriptors if isinstance(env_or_descriptors, tuple) else env_or_descriptors.last_observation_descriptors()
        if d[0].data_ptr() != src.data_ptr():
            src.copy_(d[0].reshape(self.N))
            pos.copy_(d[1].reshape(self.N, self.A))
        self._begun = True

    def state_slot(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """``descriptors_out`` for the ``env.step`` that fills the slot ``next_slot()`` / ``store()`` just handed out:
        row t of the state descriptors (t already advanced) = the observation that step returns = the state the
        policy sees at step t (row T: the bootstrap state)."""
        if not self.has_states:
            raise RuntimeError("this TrajectoryBuffer was built without states=True")
        if not self._begun:
            raise RuntimeError("call begin(env) at the start of the first chunk: row 0 of the states is not set")
        if self.t < 1:
            raise RuntimeError("state_slot() follows next_slot() / store()")
        return self._state_slots[self._cur][self.t]

    def _carry_state(self, src_chunk: int, dst_chunk: int) -> None:
        """Row T of a finished chunk is row 0 of the next one."""
        if self.has_states and self._begun:
            (s_from, p_from), (s_to, p_to) = self._state_slots[src_chunk][self.T], self._state_slots[dst_chunk][0]
            s_to.copy_(s_from)
            p_to.copy_(p_from)

    def mark_filled(self, num_steps: int) -> None:
        """A kernel wrote ``num_steps`` whole slots of the current chunk (and, with ``states=True``, descriptor rows
        0..num_steps) behind the buffer's back -- a fused rollout given ``trajectory=``."""
        if not 0 <= num_steps <= self.T:
            raise ValueError("num_steps out of range")
        self.t = int(num_steps)
        self._begun = True

    def states(self, env, t: int) -> torch.Tensor:
        """The (N, W, 5A) observation the policy saw at step t of the chunk being filled (t = len(self): the one the
        last step returned), rendered now."""
        if not self.has_states:
            raise RuntimeError("this TrajectoryBuffer was built without states=True")
        if not 0 <= t <= self.t:
            raise IndexError(f"step {t} is not filled (t = {self.t})")
        src, pos = self._state_slots[self._cur][t]
        return env.render(src, pos)

    def minibatch_states(self, env, sample_indices: torch.Tensor) -> torch.Tensor:
        """Observations (B, W, 5A) of the samples ``sample_indices`` of the filled part of the chunk, numbered as
        the reference's buffer numbers them after ``reshape`` (buffer.py:102-109: sample = env * steps + step) --
        what ``batch_states[mini_batch_indices]`` is in PPO_agent.py:182-186, rendered instead of stored."""
        if not self.has_states:
            raise RuntimeError("this TrajectoryBuffer was built without states=True")
        if self.t == 0:
            raise IndexError("the chunk being filled is empty")
        idx = sample_indices.reshape(-1).to(device=self.device, dtype=torch.int64)
        n, t = idx // self.t, idx % self.t
        return env.render(self.obs_src[t, n], self.obs_pos[t, n])

    def next_slot(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Zero-copy form of ``store``: views of slot t -- actions (N, A), rewards (N,), dones (N,) --
        for the policy and ``env.step(actions, rewards_out=..., dones_out=...)`` to write into
        directly; advances t."""
        if self.t >= self.T:
            raise IndexError("trajectory buffer is full; call clear()")
        t = self.t
        self.t += 1
        return self._slots[self._cur][t]

    def returns_and_advantages(self, values: torch.Tensor, last_values: torch.Tensor, gamma: float = 0.99
                               ) -> Tuple[torch.Tensor, torch.Tensor]:
        """(returns, advantages), both (T, N) f32, per buffer.py:80-100.  values (T, N) f32,
        last_values (N,) f32."""
        if not self._native:
            raise RuntimeError("returns_and_advantages runs on the GPU only (no CPU path)")
        T, N = self.t, self.N
        values = values.reshape(T, N).float().contiguous()
        last_values = last_values.reshape(N).float().contiguous()
        ret = torch.empty((T, N), dtype=torch.float32, device=self.device)
        adv = torch.empty((T, N), dtype=torch.float32, device=self.device)
        st = self._stream()
        rew, don = self.rewards[:T], self.dones[:T]
        if self.C != self.N:  # the scan kernel wants dense (T, N) inputs
            rew, don = rew.contiguous(), don.contiguous()
        _lib.check(self._lib.fe_traj_returns(rew.data_ptr(), don.data_ptr(), values.data_ptr(),
                                             last_values.data_ptr(), T, N, float(gamma), ret.data_ptr(),
                                             adv.data_ptr(), st))
        return ret, adv

    # ------------------------------------------------------------------ multi-GPU exchange
    @staticmethod
    def check_geometry(env, group=None) -> None:
        """Once, before the first gather of a ``states=True`` trajectory: every rank's env must have the same table
        geometry (D, L, W, A), or the descriptors one rank gathers from another are offsets into a table of a different
        shape -- out-of-bounds reads inside ``env.render`` / ``FusedLSTMRollout.forward``, not an error code.  One tiny
        all-gather; raises ``ValueError`` on every rank if they differ."""
        import torch.distributed as dist

        G = dist.get_world_size(group)
        mine = env.geometry()
        dev = env._dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
        out = torch.empty((G, 4), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out.view(-1), mine.to(dev), group=group)
        out = out.cpu()
        if not bool((out == out[0]).all()):
            raise ValueError(f"ranks disagree on the table geometry (D, L, W, A): {out.tolist()}")

    def all_gather(self, group=None, out: Optional[torch.Tensor] = None, with_states: bool = False):
        """One blocking collective: every rank receives every rank's packed chunk.

        Returns ``(actions (G, T, n, A), rewards (G, T, n), dones (G, T, n), packed)``
        (all ranks must own the same n); ``with_states=True`` returns
        ``(actions, rewards, dones, obs_src (G, T + 1, n), obs_pos (G, T + 1, n, A), packed)``."""
        import torch.distributed as dist

        self.issue_deferred()  # collectives leave in program order on every rank: a deferred chunk goes first
        G = dist.get_world_size(group)
        if out is None:
            out = torch.empty((G, self._nbytes), dtype=torch.uint8, device=self.device)
        dist.all_gather_into_tensor(out.view(-1), self._packed, group=group)
        a, r, d = self._typed(out, lead=(G,))
        if with_states:
            if not self.has_states:
                raise RuntimeError("this TrajectoryBuffer was built without states=True")
            return (a, r, d) + self._typed_states(out, lead=(G,)) + (out,)
        return a, r, d, out

    def all_gather_async(self, group=None, defer: bool = False) -> None:
        """Start gathering the chunk just filled on the collective's own stream and switch to the
        other chunk, so the exchange over xGMI overlaps the next T env steps.  ``wait_gathered``
        returns the result; a chunk is waited for automatically before it is refilled.

        ``defer=True`` only switches chunks now; the collective itself is started by ``issue_deferred()`` (or by the
        next wait / drain / gather -- the blocking ``all_gather`` included --, whichever comes first).  Starting a collective costs the HOST 20 - 30 us; a loop
        that calls it before it has queued the next steps leaves the GPU idle for that long, so a rollout loop defers it
        until a couple of steps of the new chunk are in the queue (bench.py: the world-1 RCCL path 1.99 -> 2.1 G env-steps/s
        together with the stream-ordered fence)."""
        self.issue_deferred()
        i = self._cur
        if not defer:
            self._issue(i, group)
        self._cur = 1 - i
        self._wait(self._cur)  # the chunk about to be refilled must have left
        if self.t == self.T:
            self._carry_state(i, self._cur)
        self.t = 0
        if defer:
            self._deferred = (i, group)

    def _issue(self, i: int, group) -> None:
        import torch.distributed as dist

        G = dist.get_world_size(group)
        if self._gathered[i] is None:
            self._gathered[i] = torch.empty((G, self._nbytes), dtype=torch.uint8, device=self.device)
        self._pending[i] = dist.all_gather_into_tensor(self._gathered[i].view(-1), self._chunks[i], group=group,
                                                       async_op=True)

    def issue_deferred(self) -> None:
        """Start the collective of a chunk handed over with ``all_gather_async(defer=True)`` (no-op if there is none)."""
        d = self._deferred
        if d is not None:
            self._deferred = None
            self._issue(*d)

    def _wait(self, i: int) -> None:
        d = self._deferred
        if d is not None and d[0] == i:
            self.issue_deferred()
        if self._pending[i] is not None:
            self._pending[i].wait()  # stream-level wait for NCCL/RCCL; blocks the host only for gloo
            self._pending[i] = None

    def wait_gathered(self, with_states: bool = False):
        """(actions, rewards, dones) of the most recently started gather, as (G, T, n, ...) tensors;
        ``with_states=True`` appends (obs_src (G, T, n), obs_pos (G, T, n, A)) -- renderable on this rank with
        ``env.render`` although they describe other ranks' envs."""
        i = 1 - self._cur
        self._wait(i)
        if self._gathered[i] is None:
            raise RuntimeError("no gather has been started")
        lead = (self._gathered[i].shape[0],)
        out = self._typed(self._gathered[i], lead=lead)
        if with_states:
            if not self.has_states:
                raise RuntimeError("this TrajectoryBuffer was built without states=True")
            out = out + self._typed_states(self._gathered[i], lead=lead)
        return out

    def drain(self) -> None:
        self.issue_deferred()
        for i in (0, 1):
            self._wait(i)

    @staticmethod
    def join_shards(x: torch.Tensor, total_envs: int) -> torch.Tensor:
        """(G, T, capacity, ...) gathered field -> (T, total_envs, ...): shards concatenated in rank order,
        capacity padding dropped (shards as produced by ``shard_range``)."""
        from .environments.time_series_env import shard_range

        G = x.shape[0]
        parts = []
        for r in range(G):
            lo, hi = shard_range(total_envs, r, G)
            parts.append(x[r][:, : hi - lo])
        return torch.cat(parts, dim=1)
