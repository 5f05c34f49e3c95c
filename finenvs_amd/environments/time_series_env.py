"""TimeSeriesEnv: the reference's Gym-like vectorised trading env, MI355X-native.

Drop-in for ``finenvs.environments.time_series_env.TimeSeriesEnv`` ("TSE"): same
constructor arguments (TSE:15-29), same ``reset()`` / ``step(actions)`` /
``get_env_args()`` protocol (TSE:236-243, 277-296, 423-435), same attributes
callers read (``num_envs, num_obs, num_acts, device, action_space,
observation_space, cash, margin, long_shares, short_shares, env_indices,
env_pointers, env_spots``).  The per-step state transition is one fused HIP
kernel behind the C ABI of include/finenvs_amd.h; this class only allocates
torch tensors, hands their device pointers to that ABI and launches on torch's
current stream.  There is no CPU path.

Keyword-only extensions: ``num_envs`` (env n -> day n mod D), ``num_assets`` /
``prices`` / ``day_id`` (tensor input, multi-asset "sleeve" contract of
DESIGN.md), ``tables`` (ready-made (D,L,4A) price/log-return tables),
``obs_dtype``, ``cast_actions`` (cast non-f32 actions to f32; default: float64 actions take the reference's f64
promotion of its share tensors, other dtypes raise ValueError), ``backend`` (only ``"hip"``: there is no CPU path), ``obs_buffers`` (opt-in ring of env-owned observation buffers), ``obs_audition`` (ring mode:
extra candidate buffers to try at construction -- within a quarter of the free memory --, the fastest stay), ``redraw``, ``seed``, ``env_indices``, ``rank`` /
``world_size`` (contiguous env shards, one process per GPU).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .. import _lib
from ..data import loader
from ..device_utils import set_device
from ..rng import redraw_day
from ..spaces import Box


# raw hipStream_t of torch's current stream without building a torch.cuda.Stream object (~0.3 us vs ~4 us)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class HostFlagTimeout(RuntimeError):
    """The step kernel never reported through its host flag (see ``poll_host_word``)."""


def poll_host_word(read, matches, timeout_s: float, on_timeout=None, spin: int = 2000, clock=None, sleep=None) -> int:
    """Wait until ``matches(read())`` and return that word: the host half of fe_env_step_notify (the reference's
    per-step host read, TSE:510 / TSE:531, without its device-to-host copy).  The kernel writes the word a few
    microseconds after it starts, so the first ``spin`` polls are a tight loop; after that the thread yields
    (``sleep(0)``), and from 20 ms on it sleeps 50 us per poll -- a stream that is merely slow (queued work ahead, a
    profiler, a shared GPU) does not cost a core.  After ``timeout_s``: ``on_timeout()`` (the caller synchronises its
    stream there, which surfaces a launch / kernel error and lets a slow launch finish), then ONE more read -- only if
    the word still does not match is ``HostFlagTimeout`` raised."""
    import time

    clock = clock or time.monotonic
    sleep = sleep or time.sleep
    v = read()
    if matches(v):
        return v
    for _ in range(spin):
        v = read()
        if matches(v):
            return v
    t0 = clock()
    while True:
        v = read()
        if matches(v):
            return v
        el = clock() - t0
        if el > timeout_s:
            if on_timeout is not None:
                on_timeout()
            v = read()
            if matches(v):
                return v
            raise HostFlagTimeout(f"the step kernel did not report through its host flag within {timeout_s:g} s "
                                  f"(last word {v:#x})")
        sleep(0.0 if el < 0.02 else 50e-6)


def shard_range(num_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block of envs owned by ``rank``."""
    base, rem = divmod(num_envs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class TimeSeriesEnv:
    def __init__(
        self,
        instrument_name: Union[str, Sequence[str]] = "synthetic",
        dataset_key: str = "dummy",
        num_intervals: int = 390,
        max_shares: int = 5,
        starting_balance: float = 10000,
        per_share_commission: float = 0.01,
        initial_margin_requirement: float = 1.5,
        maintenance_margin_requirement: float = 0.25,
        evaluate: bool = False,
        device_id: int = 0,
        *,
        num_envs: Optional[int] = None,
        num_assets: Optional[int] = None,
        prices=None,
        day_id=None,
        tables=None,
        obs_dtype: torch.dtype = torch.float64,
        obs_buffers: int = 0,
        obs_audition: int = 0,
        redraw: str = "torch",
        cast_actions: bool = False,
        backend: str = "hip",
        seed: int = 0,
        env_indices=None,
        rank: int = 0,
        world_size: int = 1,
        _native=None,
    ):
        self.instrument_name = instrument_name if isinstance(instrument_name, str) else "+".join(instrument_name)
        self.num_intervals = int(num_intervals)
        self.max_shares = max_shares
        self.starting_balance = starting_balance
        self.per_share_commission = per_share_commission
        self.initial_margin_requirement = initial_margin_requirement
        self.maintenance_margin_requirement = maintenance_margin_requirement
        self.log_return_scale_factor = 100
        self.evaluate = bool(evaluate)
        if backend != "hip":
            # SURVEY 8(b) names a `backend=` keyword ("hip" | a CPU restatement); this product has no CPU path on purpose
            raise ValueError(f"backend={backend!r}: finenvs_amd runs the HIP path only (the CPU restatement is test infrastructure, oracle/)")
        if redraw not in ("torch", "device"):
            raise ValueError("redraw must be 'torch' (reference RNG stream) or 'device' (Philox, no host sync)")
        if obs_dtype not in (torch.float64, torch.float32):
            raise ValueError("obs_dtype must be torch.float64 or torch.float32")
        self.redraw = redraw
        self.cast_actions = bool(cast_actions)
        self.seed = int(seed)
        self.obs_dtype = obs_dtype
        self.rank, self.world_size = int(rank), int(world_size)
        # dataset-key / file errors come first, exactly as in the reference (TSE:38-40)
        from_files = prices is None and tables is None
        if from_files:
            names = [instrument_name] if isinstance(instrument_name, str) else list(instrument_name)
            self.file_key = loader.determine_file_key(dataset_key)
            self.data_dir_name = loader.get_data_dir_name(names[0])
            self.filename = [loader.find_file_by_key(loader.get_data_dir_name(n), self.file_key) for n in names]
        self.device = set_device(device_id)
        # _native: a library handle from _lib.load(path) -- how tools/ A/B an experiment build against the
        # product in one process; everything else gets the product library
        self._lib = _native if _native is not None else _lib.load()
        self._dev = torch.device(self.device)
        self._dev_index = int(device_id)
        with torch.cuda.device(self._dev):
            self._build_tables(from_files, prices, day_id, tables, num_assets)
            self.set_spaces()
            self.set_environment_params(num_envs, env_indices, obs_buffers)
            if obs_audition > 0 and self.obs_buffers > 0:
                self.audition_ring(int(obs_audition))

    # ------------------------------------------------------------------ init path
    def _stream(self) -> int:
        if _raw_stream is not None:
            return _raw_stream(self._dev_index)
        return torch.cuda.current_stream(self._dev).cuda_stream

    def _build_tables(self, from_files, prices, day_id, tables, num_assets) -> None:
        """process_data / set_up_environments (TSE:75-101, 165-216) with the transform and the
        NaN-padded slicing done on the GPU."""
        W = self.num_intervals
        if tables is not None:
            P, LR = tables
            self.price_environments = torch.as_tensor(P, dtype=torch.float64).to(self._dev).contiguous()
            self.log_return_environments = torch.as_tensor(LR, dtype=torch.float64).to(self._dev).contiguous()
            if self.price_environments.shape != self.log_return_environments.shape or self.price_environments.dim() != 3:
                raise ValueError("tables must be two (D, L, 4*A) arrays of equal shape")
            pad = torch.isnan(self.price_environments[:, :, 0]).sum(dim=1).tolist()
            self._padding_rows = [int(x) for x in pad]
        else:
            if from_files:
                series, day_id, _ = loader.read_csv_portfolio(self.filename)
            else:
                series = prices.detach().cpu().numpy() if isinstance(prices, torch.Tensor) else np.asarray(prices)
                series = np.ascontiguousarray(series, dtype=np.float64)
                if day_id is None:
                    raise ValueError("prices= needs day_id= (one date label per row, market hours only)")
                day_id = day_id.detach().cpu().numpy() if isinstance(day_id, torch.Tensor) else np.asarray(day_id)
            if series.ndim != 2 or series.shape[1] % 4 != 0:
                raise ValueError("price series must be (T, 4*A): O,H,L,C per asset")
            starts, stops, L = loader.episode_bounds(day_id, W)
            if len(starts) == 0:
                raise Exception("no trading day has num_intervals bars of history before it")
            self._padding_rows = loader.padding_rows(starts, stops, L)
            T, c4 = series.shape
            A = c4 // 4
            st = self._stream()
            self.dataset = torch.from_numpy(series).to(self._dev)
            self.log_return_dataset = torch.empty_like(self.dataset)
            _lib.check(self._lib.fe_build_logret(self.dataset.data_ptr(), self.log_return_dataset.data_ptr(), T, A, st))
            d_starts = torch.from_numpy(starts).to(self._dev)
            d_stops = torch.from_numpy(stops).to(self._dev)
            D = len(starts)
            self.price_environments = torch.empty((D, L, c4), dtype=torch.float64, device=self._dev)
            self.log_return_environments = torch.empty((D, L, c4), dtype=torch.float64, device=self._dev)
            for src, dst in ((self.dataset, self.price_environments), (self.log_return_dataset, self.log_return_environments)):
                _lib.check(self._lib.fe_build_tables(src.data_ptr(), d_starts.data_ptr(), d_stops.data_ptr(), D, L, A,
                                                     dst.data_ptr(), st))
            torch.cuda.current_stream(self._dev).synchronize()  # d_starts/d_stops go out of scope
        D, L, c4 = self.price_environments.shape
        self.num_assets = c4 // 4
        if num_assets is not None and int(num_assets) != self.num_assets:
            raise ValueError(f"num_assets={num_assets} but the price data has {self.num_assets} assets")
        if not (1 <= self.num_assets <= _lib.FE_MAX_ASSETS):
            raise ValueError(f"num_assets must be in 1..{_lib.FE_MAX_ASSETS}")
        if L <= W:
            raise ValueError("every episode needs at least one bar after the window")
        self.values_per_interval = 4
        if self.redraw == "torch":
            # the reference burns global-generator draws while filling its NaN padding
            # (TSE:207-210); replay them so a seeded run sees the same stream afterwards
            for rem in self._padding_rows:
                if rem > 0:
                    torch.rand((rem, 4), device=self._dev)

    def set_spaces(self) -> None:
        """TSE:218-234, widened to A assets: per asset 4 log-returns + position."""
        A = self.num_assets
        self.num_obs = (self.values_per_interval + 1) * A
        self.num_acts = A
        self.action_space = Box(np.ones(self.num_acts) * -1.0, np.ones(self.num_acts) * +1.0, dtype=np.float64)
        self.observation_space = Box(
            np.ones((self.num_intervals, self.num_obs)) * -np.inf,
            np.ones((self.num_intervals, self.num_obs)) * +np.inf,
            dtype=np.float64,
        )

    def get_env_args(self) -> Dict:
        """TSE:236-243."""
        return {
            "env_name": self.instrument_name,
            "num_envs": self.num_envs,
            "num_observations": self.num_obs,
            "num_actions": self.num_acts,
            "sequence_length": self.num_intervals,
        }

    def set_environment_params(self, num_envs, env_indices, obs_buffers) -> None:
        """TSE:245-269: state tensors; plus the C-ABI env object."""
        dev = self._dev
        D = self.price_environments.shape[0]
        training = not self.evaluate
        if env_indices is not None:
            idx = torch.as_tensor(env_indices, dtype=torch.int64).to(dev).contiguous().clone()
            total = idx.shape[0]
            lo, hi = 0, total
            self._eval_env = total - 1 if training else -1
        else:
            total = int(num_envs) if num_envs is not None else D + (1 if training else 0)
            if total < 1:
                raise ValueError("num_envs must be >= 1")
            lo, hi = shard_range(total, self.rank, self.world_size)
            idx = (torch.arange(lo, hi, dtype=torch.int64, device=dev) % D).contiguous()
            # the last env overall is the evaluation env and starts on a random day (TSE:253-257)
            has_eval = training and hi == total and hi > lo
            self._eval_env = (hi - lo - 1) if has_eval else -1
            if training and self.redraw == "torch":
                first = torch.randint(0, D, (1,), device=dev)  # drawn on every rank to keep streams aligned
                if has_eval:
                    idx[-1:] = first
            elif has_eval:
                idx[-1] = redraw_day(self.seed, 0, D)
        self._allocate_state(idx, total, lo, obs_buffers)

    def _allocate_state(self, idx: torch.Tensor, total: int, lo: int, obs_buffers: int) -> None:
        """The state tensors of TSE:245-269 for the day indices ``idx`` + the C-ABI env object bound to them.  The env keeps
        PRIVATE references (``_cash`` ...) to the storage whose pointers ``fe_env`` holds: the public names are
        properties over them, so nothing a caller assigns can free or detach the memory the kernel writes."""
        dev = self._dev
        D, L, _ = self.price_environments.shape
        A, W = self.num_assets, self.num_intervals
        training = not self.evaluate
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= D):
            raise ValueError("env_indices out of range")
        self._env_indices = idx
        self._num_envs = int(idx.shape[0])
        self.global_num_envs = total
        self.env_offset = lo
        N = self._num_envs
        if N < 1:
            raise ValueError("this rank owns no envs")
        self._spot0 = torch.zeros((N,), dtype=torch.int64, device=dev)
        self._cash = self.starting_balance * torch.ones((N, A), device=dev)
        self._long = torch.zeros((N, A), device=dev)
        self._short = torch.zeros((N, A), device=dev)
        self._margin = torch.zeros((N, A), dtype=torch.float64, device=dev)
        self._terminated = torch.zeros((N,), dtype=torch.uint8, device=dev)
        self._returns = torch.zeros((N,), device=dev)
        # counters[0] = envs terminated so far (evaluate mode), counters[1] = redraw counter
        self._counters = torch.zeros((2,), dtype=torch.int64, device=dev)
        if training and self.redraw == "device":
            self._counters[1] = 1  # draw 0 chose the eval env's first day
        cfg = _lib.FeConfig(
            N, D, L, W, A, int(self.max_shares), int(self.evaluate), float(self.starting_balance),
            float(self.per_share_commission), float(self.initial_margin_requirement),
            float(self.maintenance_margin_requirement), int(self.obs_dtype == torch.float32),
            1 if self.redraw == "device" else 0, self.seed, self._eval_env,
        )
        handle = C.c_void_p()
        _lib.check(self._lib.fe_env_create(C.byref(cfg), self.price_environments.data_ptr(),
                                           self.log_return_environments.data_ptr(), C.byref(handle)))
        self._handle = handle
        _lib.check(self._lib.fe_env_bind_state(
            handle, self._env_indices.data_ptr(), self._spot0.data_ptr(), self._cash.data_ptr(),
            self._long.data_ptr(), self._short.data_ptr(), self._margin.data_ptr(),
            self._terminated.data_ptr(), self._returns.data_ptr(), self._counters.data_ptr()))
        if self.obs_dtype == torch.float32:
            # f32 observations stream from a pre-cast copy of the table (half the L2 reads, same values)
            if getattr(self, "_log_return_f32", None) is None:
                self._log_return_f32 = self.log_return_environments.float().contiguous()
            _lib.check(self._lib.fe_env_bind_f32_table(handle, self._log_return_f32.data_ptr()))
        # observation ring: 0 = fresh tensor per call (reference semantics), k = k env-owned buffers
        self.obs_buffers = int(obs_buffers)
        self._obs_ring = [torch.empty((N, W, 5 * A), dtype=self.obs_dtype, device=dev) for _ in range(self.obs_buffers)]
        self._obs_next = 0
        self._step_fn = self._lib.fe_env_step
        self._handle_v = handle.value
        # redraw="torch" with an evaluation env: the per-step host read of its done flag (TSE:510) goes through a
        # coherent host flag the kernel writes as soon as that env is accounted (fe_env_step_notify), not through a
        # device-to-host copy after the launch
        # the reference's share tensors become f64 at the first step() with float64 actions and stay f64 (step()): from then
        # on every step goes through fe_env_step_promoted
        self.shares_promoted = False
        self._mirrors = {}                  # promoted f64 share tensors handed out since the last launch (_promoted_mirror)
        self._terminated_view_out = False   # the bool view of the termination flags was handed out (_sync_public_views)
        self._flag = None
        self._flag_seq = 0
        self.flag_timeout_s = 60.0  # how long step() waits for the kernel's word before it synchronises and gives up
        if (self.redraw == "torch" and self._eval_env >= 0) or self.evaluate:
            # (evaluate mode: the per-step host read is "have all envs terminated?", TSE:531 -- the notify form's last
            # workgroup reports the count)
            flag = C.c_void_p()
            _lib.check(self._lib.fe_host_flag_create(C.byref(flag)))
            self._flag = flag
            self._flag_word = C.c_uint64.from_address(flag.value)
        # bumped by everything that advances the env (step, a fused rollout's run): the fused rollout objects keep their
        # own observation descriptors and refuse to run on ones that another caller has made stale (rollout.py)
        self._generation = getattr(self, "_generation", 0) + 1
        # bumped when the bound storage itself is replaced (a resize through the env_indices setter): captured graphs and
        # statistics objects hold the old pointers and refuse to run (rollout.GraphedRollout.run)
        self._binding_epoch = getattr(self, "_binding_epoch", -1) + 1
        self._last_descriptors, self._stepped = None, False

    def _release_native(self) -> None:
        """Destroy the C-ABI env object and its host flag (after the device has drained: the last launch may still be about
        to write either)."""
        h = getattr(self, "_handle", None)
        if h is None or getattr(self, "_lib", None) is None:
            return
        try:
            torch.cuda.synchronize(self._dev)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass
        if getattr(self, "_flag", None) is not None:
            self._lib.fe_host_flag_destroy(self._flag)
            self._flag = None
        self._lib.fe_env_destroy(h)
        self._handle = None
        self._handle_v = None

    def audition_ring(self, extra: int = 2, budget_bytes: Optional[int] = None, min_gain: float = 0.03) -> None:
        """Ring mode only (also what ``obs_audition=`` runs at construction).  HBM write bandwidth on MI355X depends on where a buffer lies (the same store kernel runs
        5.7 ... 6.5 TB/s on different 20 GB allocations, reproducibly per buffer; profiles/r02_microbench/placement_20g.txt, DESIGN.md
        section 4), and the step kernel is bound by exactly that.  So: allocate up to ``extra`` more candidate
        buffers than the ring needs, time the observation render into each, let a candidate replace the slowest ring
        member where it is faster by more than ``min_gain`` (3 %: less is timing noise) and give the rest back.  Values are unaffected; ``self.obs_audition`` records what was measured.

        The audition is BOUNDED: at most ``extra`` candidates (default 2), together at most ``budget_bytes`` (default:
        one quarter of the memory that is free right now), and never into the last 8 GiB of free memory -- constructing
        a 20-GB-per-buffer env must not transiently hold the whole card (12 x 20 GB at config 3)."""
        if self.obs_buffers < 1:
            raise ValueError("audition_ring needs ring mode (obs_buffers >= 1)")
        N, W, A = self.num_envs, self.num_intervals, self.num_assets
        nbytes = N * W * 5 * A * (4 if self.obs_dtype == torch.float32 else 8)
        free, _ = torch.cuda.mem_get_info(self._dev)
        budget = free // 4 if budget_bytes is None else int(budget_bytes)
        budget = max(0, min(budget, free - (8 << 30)))
        extra = int(max(0, min(int(extra), budget // max(nbytes, 1))))
        cands = list(self._obs_ring)
        try:
            for _ in range(extra):
                cands.append(torch.empty((N, W, 5 * A), dtype=self.obs_dtype, device=self._dev))
        except RuntimeError:  # out of memory: audition what we have
            pass
        if len(cands) <= self.obs_buffers:
            self.obs_audition = {"candidates": len(cands), "us": [], "kept": list(range(len(cands))), "budget_bytes": budget}
            return
        st = self._stream()

        def train(buf, k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(k):
                _lib.check(self._lib.fe_env_reset_obs(self._handle, buf.data_ptr(), st))
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e3 / k

        # Measured as the step is: trains of back-to-back renders (~2 ms each) after ~20 ms of settling launches (the box's
        # clock transient after an idle period is 15 - 25 %, more than the differences looked for), candidates interleaved
        # over three rounds, median per candidate.  (Round 4 timed single launches between host synchronisations: at 64k
        # envs it swapped ring members on noise -- twice out of two runs the "faster" ring was 1 % slower.)
        train(cands[0], 2)
        est_us = train(cands[0], 2)
        k = int(max(2, min(32, 2000.0 / max(est_us, 1e-3))))
        train(cands[0], int(max(2, min(1000, 20000.0 / max(est_us, 1e-3)))))
        rounds = [[train(buf, k) for buf in cands] for _ in range(3)]
        times = [sorted(r[i] for r in rounds)[1] for i in range(len(cands))]
        # a candidate replaces the slowest ring member only if it is faster by more than `min_gain`: at 64k envs all
        # candidates lie within 2 % of each other and a swap on noise made a driver-like run 2.5 % SLOWER (round 4)
        ring = list(range(self.obs_buffers))
        for c in sorted(range(self.obs_buffers, len(cands)), key=lambda i: times[i]):
            worst = max(ring, key=lambda i: times[i])
            if times[c] < times[worst] * (1.0 - min_gain):
                ring[ring.index(worst)] = c
        kept = sorted(ring)
        self._obs_ring = [cands[i] for i in kept]
        self._obs_next = 0
        self.obs_audition = {"candidates": len(cands), "us": [round(t, 2) for t in times], "kept": kept, "budget_bytes": budget,
                             "min_gain": min_gain}
        del cands
        torch.cuda.empty_cache()

    def print(self) -> None:
        """Attribute dump, the reference's BaseObject.print (finenvs/base_object.py:7-9); device tensors
        are summarised by shape and dtype instead of being copied to the host and printed."""
        public = ("num_envs", "env_indices", "env_pointers", "env_spots", "cash", "long_shares", "short_shares", "margin",
                  "terminated_episodes", "episode_returns")
        for key in sorted([k for k in vars(self) if not k.startswith("_")] + list(public)):
            val = getattr(self, key)
            if isinstance(val, torch.Tensor):
                val = f"Tensor{tuple(val.shape)} {str(val.dtype).replace('torch.', '')} on {val.device}"
            elif isinstance(val, (list, tuple)) and val and isinstance(val[0], torch.Tensor):
                val = f"[{len(val)} x Tensor{tuple(val[0].shape)}]"
            print(f"{key:32s} {val}")

    def launch_info(self) -> Dict[str, int]:
        g, b, t, l = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self._lib.fe_env_launch_info(self._handle, C.byref(g), C.byref(b), C.byref(t), C.byref(l)))
        return {"grid": g.value, "block": b.value, "tile_envs": t.value, "lds_bytes": l.value}

    def set_launch(self, tile_envs: int = 0, grid: int = 0, rollout_tile_envs: int = 0) -> Dict[str, int]:
        """Tuning hook for tools/: override the tile (envs per workgroup) / grid of the step kernel and the
        tile of the fused rollouts; 0 = automatic.  Returns the resulting launch_info()."""
        _lib.check(self._lib.fe_env_set_launch(self._handle, int(tile_envs), int(grid), int(rollout_tile_envs)))
        return self.launch_info()

    def __del__(self):
        self._release_native()

    def _poll_flag(self, shift: int, seq: int) -> int:
        """The host flag's word once its sequence field (bits ``shift`` and up) carries ``seq``."""
        word = self._flag_word
        for _ in range(2000):  # the usual case, inline: the word arrives 8 - 11 us after the launch call
            v = word.value
            if (v >> shift) == seq:
                return v
        return poll_host_word(lambda: word.value, lambda v: (v >> shift) == seq, self.flag_timeout_s,
                              on_timeout=lambda: torch.cuda.synchronize(self._dev), spin=0)

    def _eval_env_done(self, seq: int) -> bool:
        """Poll the host flag of fe_env_step_notify until it carries this step's sequence number; its low bit is the
        evaluation env's done flag (TSE:510).  The kernel writes it a few microseconds after it starts."""
        return bool(self._poll_flag(1, seq) & 1)

    # ------------------------------------------------------------------ public state (TSE:245-269)
    # The reference keeps its state in plain attributes and callers REBIND them (`env.cash = S * torch.ones(N, 1)`, SURVEY
    # Appendix B scales an env to N copies that way).  Here fe_env holds raw device pointers, so the public names are
    # properties over env-owned storage: the getter returns the bound tensor itself (in-place edits reach the kernel), the
    # setter COPIES the assigned values into it -- cast to the dtype the reference ends up holding (cash / shares f32,
    # margin f64, TSE:376-383) -- and the storage can neither be freed nor detached by an assignment.  Pointers stay
    # stable, so captured hipGraphs stay valid.  Assigning `env_indices` of another length resizes the env.
    def _assign(self, name: str, bound: torch.Tensor, value) -> None:
        v = torch.as_tensor(value)
        if v.numel() != bound.numel():
            raise ValueError(f"{name}: expected {bound.numel()} values ({tuple(bound.shape)}; the reference's shape is "
                             f"({self._num_envs}, 1) per asset), got {tuple(v.shape)}"
                             + ("" if name == "env_indices" else " -- assign env_indices first to resize the env"))
        bound.copy_(v.to(device=self._dev).reshape(bound.shape))  # copy_ casts (f32 margin zeros -> f64, bool -> u8 ...)

    @property
    def num_envs(self) -> int:
        return self._num_envs

    @num_envs.setter
    def num_envs(self, n) -> None:
        # (the reference's recipe assigns num_envs after env_indices: nothing to do then)
        if int(n) != self._num_envs:
            raise ValueError(f"num_envs is {self._num_envs}: it follows env_indices -- assign env_indices ({int(n)} day "
                             "indices) to resize the env")

    @property
    def env_indices(self) -> torch.Tensor:
        """(N,) int64 day index of every env (TSE:246-257)."""
        return self._env_indices

    @env_indices.setter
    def env_indices(self, value) -> None:
        idx = torch.as_tensor(value).to(device=self._dev, dtype=torch.int64).reshape(-1).contiguous()
        D = self.price_environments.shape[0]
        if idx.numel() < 1 or int(idx.min()) < 0 or int(idx.max()) >= D:
            raise ValueError(f"env_indices must be day indices in [0, {D})")
        if idx.numel() == self._num_envs:
            self._env_indices.copy_(idx)
            return
        # another length: the reference's way of scaling an env (rebinding every state tensor) -- here the env object is
        # rebuilt for the new N with every account in its initial state (the recipe's other assignments then match)
        if self.world_size != 1:
            raise ValueError("resizing a sharded env (world_size > 1) is not supported: construct it with num_envs=")
        self._resize(idx.clone())

    # everything _allocate_state (re)binds: what a failed resize puts back
    _STATE_ATTRS = ("_env_indices", "_num_envs", "global_num_envs", "env_offset", "_spot0", "_cash", "_long", "_short", "_margin",
                    "_terminated", "_returns", "_counters", "_handle", "_handle_v", "obs_buffers", "_obs_ring", "_obs_next", "_step_fn",
                    "shares_promoted", "_flag", "_flag_seq", "_flag_word", "flag_timeout_s", "_generation", "_binding_epoch",
                    "_last_descriptors", "_stepped", "_eval_env", "_mirrors", "_terminated_view_out")

    def _resize(self, idx: torch.Tensor) -> None:
        """Rebuild state + native env for another env count WITHOUT losing the env when that fails (out of memory while scaling
        up, a refused fe_env_create): the new state is built beside the old one and swapped in only once it is complete; the old
        native objects are released after that.  If memory is short the old OBSERVATION RING (output buffers, not state) is
        given up first and the build retried; a failure after that restores the old state with a fresh ring.  User-set
        ``flag_timeout_s`` and the device redraw counter carry over; an auditioned ring does not (the storage is new:
        ``audition_ring`` again if wanted)."""
        old = {k: self.__dict__[k] for k in self._STATE_ATTRS if k in self.__dict__}
        if self._dev.type == "cuda":
            torch.cuda.synchronize(self._dev)  # the last launch may still write the old state / flag
        redraw_counter = int(self._counters[1].item())

        def attempt():
            self._handle = self._flag = None  # (nothing below may release the OLD native objects)
            self._eval_env = idx.numel() - 1 if not self.evaluate else -1
            try:
                self._allocate_state(idx, idx.numel(), 0, old["obs_buffers"])
            except BaseException:
                h, f = self.__dict__.get("_handle"), self.__dict__.get("_flag")  # what the failed build got as far as creating
                if f is not None:
                    self._lib.fe_host_flag_destroy(f)
                if h is not None:
                    self._lib.fe_env_destroy(h)
                self.__dict__.update(old)
                raise

        try:
            attempt()
        except torch.cuda.OutOfMemoryError:
            ring_shape = [(tuple(t.shape), t.dtype) for t in old["_obs_ring"]]
            old["_obs_ring"] = self._obs_ring = []
            torch.cuda.empty_cache()
            try:
                attempt()
            except BaseException:
                # the env as it was, with a fresh ring (holders of the old buffers' pointers are told: new binding epoch)
                self._obs_ring = [torch.empty(sh, dtype=dt, device=self._dev) for sh, dt in ring_shape]
                self._obs_next = 0
                self._binding_epoch += 1
                raise
        # the new env is complete: now the old native objects can go
        if old.get("_flag") is not None:
            self._lib.fe_host_flag_destroy(old["_flag"])
        if old.get("_handle") is not None:
            self._lib.fe_env_destroy(old["_handle"])
        self.flag_timeout_s = old["flag_timeout_s"]
        self.obs_audition = None  # (what an earlier audition measured described the old ring)
        if not self.evaluate and self.redraw == "device":
            self._counters[1] = redraw_counter  # the Philox stream goes on where it was

    @property
    def cash(self) -> torch.Tensor:
        """(N, A) float32 (the reference: (N, 1))."""
        return self._cash

    @cash.setter
    def cash(self, value) -> None:
        self._assign("cash", self._cash, value)

    @property
    def margin(self) -> torch.Tensor:
        """(N, A) float64: what the reference's margin is from its first step on (TSE:376-383)."""
        return self._margin

    @margin.setter
    def margin(self, value) -> None:
        self._assign("margin", self._margin, value)

    @property
    def long_shares(self) -> torch.Tensor:
        """(N, A) float32 share counts; once a float64-action step has run (``shares_promoted``) a float64 COPY of them, as
        the reference's tensor is float64 from then on (TSE:361; it rebinds the attribute every step, so no caller can rely
        on aliasing it).  Write through assignment (``env.long_shares = t``), not into the returned copy."""
        return self._promoted_mirror("_long") if self.shares_promoted else self._long

    @long_shares.setter
    def long_shares(self, value) -> None:
        self._mirrors.pop("_long", None)
        self._assign("long_shares", self._long, value)

    @property
    def short_shares(self) -> torch.Tensor:
        """See ``long_shares`` (TSE:375)."""
        return self._promoted_mirror("_short") if self.shares_promoted else self._short

    @short_shares.setter
    def short_shares(self, value) -> None:
        self._mirrors.pop("_short", None)
        self._assign("short_shares", self._short, value)

    def _promoted_mirror(self, name: str) -> torch.Tensor:
        """The float64 tensor a promoted env hands out for ``long_shares`` / ``short_shares``: ONE tensor per attribute until the
        next launch, and whatever the caller wrote into it in place (``env.long_shares[mask] = 0``, ``.zero_()``: what the
        reference's callers do to its attribute) is copied back into the bound f32 storage before that launch
        (``_sync_public_views``) -- share counts are small integers, the cast is exact."""
        m = self._mirrors.get(name)
        if m is None:
            m = self._mirrors[name] = getattr(self, name).double()
        return m

    def _sync_public_views(self) -> None:
        """Before anything reads the bound state on the device: in-place edits made through handed-out VIEWS that are not the
        bound storage itself -- the promoted f64 share mirrors, the bool view of the termination flags (whose count the
        evaluate-mode step compares with num_envs, TSE:531) -- reach the kernel.  Costs nothing unless such a view was handed
        out since the last launch."""
        if self._mirrors:
            for name, m in self._mirrors.items():
                getattr(self, name).copy_(m)
            self._mirrors.clear()
        if self._terminated_view_out:
            self._counters[0] = self._terminated.sum()
            self._terminated_view_out = False

    @property
    def terminated_episodes(self) -> torch.Tensor:
        """(N,) bool view of the kernel's u8 flags (TSE:272-274); in-place writes into it are counted before the next step."""
        self._terminated_view_out = True
        return self._terminated.view(torch.bool)

    @terminated_episodes.setter
    def terminated_episodes(self, value) -> None:
        v = torch.as_tensor(value)
        self._assign("terminated_episodes", self._terminated, v != 0)
        self._counters[0] = self._terminated.sum()

    @property
    def episode_returns(self) -> torch.Tensor:
        """(N,) float32 (TSE:275)."""
        return self._returns

    @episode_returns.setter
    def episode_returns(self, value) -> None:
        self._assign("episode_returns", self._returns, value)

    @property
    def env_spots(self) -> torch.Tensor:
        """(N, W) window row indices; the reference stores this dense array (TSE:261-263).  Derived from the one int64 per
        env the kernel keeps: never streamed."""
        return self._spot0.unsqueeze(1) + torch.arange(self.num_intervals, device=self._dev)

    @env_spots.setter
    def env_spots(self, value) -> None:
        v = torch.as_tensor(value).to(device=self._dev, dtype=torch.int64)
        N, W = self._num_envs, self.num_intervals
        if tuple(v.shape) != (N, W):
            raise ValueError(f"env_spots must be ({N}, {W}), got {tuple(v.shape)}")
        first = v[:, 0].contiguous()
        L = self.price_environments.shape[1]
        if not torch.equal(v, first.unsqueeze(1) + torch.arange(W, device=self._dev)):
            raise ValueError("env_spots rows must be consecutive (spot0 + arange(W)): the window is the only shape the "
                             "reference's own step keeps (TSE:282, 515-521)")
        if int(first.min()) < 0 or int(first.max()) + W >= L:
            raise ValueError(f"env_spots out of range: every window needs one bar after it (spot0 + {W} < {L})")
        self._spot0.copy_(first)

    @property
    def env_pointers(self) -> torch.Tensor:
        """Steps since the last reset; always equal to env_spots[:, 0] (TSE:281-282, 514)."""
        return self._spot0.clone()

    @env_pointers.setter
    def env_pointers(self, value) -> None:
        # a write-only counter in the reference (TSE:258, 281, 514: never read); here it is env_spots[:, 0]
        v = torch.as_tensor(value).to(device=self._dev, dtype=torch.int64).reshape(-1)
        if v.numel() != self._num_envs or not torch.equal(v, self._spot0):
            raise ValueError("env_pointers is derived (== env_spots[:, 0]): assign env_spots to move the windows")

    def reset_evaluation_metrics(self) -> None:
        """TSE:271-275."""
        self._terminated.zero_()
        self._returns.zero_()
        self._counters[0] = 0

    # ------------------------------------------------------------------ hot path
    def _next_obs(self) -> torch.Tensor:
        if self.obs_buffers == 0:
            return torch.empty((self._num_envs, self.num_intervals, 5 * self.num_assets), dtype=self.obs_dtype,
                               device=self._dev)
        buf = self._obs_ring[self._obs_next]
        self._obs_next = (self._obs_next + 1) % self.obs_buffers
        return buf

    def reset(self) -> torch.Tensor:
        """Render the observation of the current state (TSE:423-435; it resets nothing)."""
        if self._mirrors or self._terminated_view_out:
            self._sync_public_views()
        obs = self._next_obs()
        _lib.check(self._lib.fe_env_reset_obs(self._handle, obs.data_ptr(), self._stream()))
        self._last_descriptors, self._stepped = None, False  # the caller now looks at the current state again
        return obs

    def last_observation_descriptors(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Descriptors of the observation most recently handed to the caller: what the last ``step`` recorded through
        ``descriptors_out``, or -- after ``reset()`` / construction -- the current state's.  (After a done step the
        returned observation is the terminal window, TSE:321, which the post-reset state no longer describes: a step
        without ``descriptors_out`` leaves nothing to return here.)"""
        if getattr(self, "_last_descriptors", None) is not None:
            return self._last_descriptors
        if getattr(self, "_stepped", False):
            raise RuntimeError("the last step() was not given descriptors_out: its observation cannot be described "
                               "after the fact (call reset() to look at the current state instead)")
        return self.describe()

    def describe(self, src_out: Optional[torch.Tensor] = None, pos_out: Optional[torch.Tensor] = None
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
        """The observation ``reset()`` would render now, as DESCRIPTORS: ``obs_src (N,) int64`` (window offset into
        the log-return table) and ``obs_pos (N, A) float64`` (position feature) -- 8 + 8A bytes per env instead of
        40WA.  ``render`` turns descriptors back into observations, on this or any other rank (the tables are
        replicated).  A trajectory of descriptors (``TrajectoryBuffer(states=True)``) is the ``states`` field of the
        reference's PPO buffer (finenvs/agents/PPO/buffer.py:33-56) without its bytes."""
        N, A = self.num_envs, self.num_assets
        src = src_out if src_out is not None else torch.empty((N,), dtype=torch.int64, device=self._dev)
        pos = pos_out if pos_out is not None else torch.empty((N, A), dtype=torch.float64, device=self._dev)
        for t, shape, dt in ((src, (N,), torch.int64), (pos, (N, A), torch.float64)):
            if tuple(t.shape) != shape or t.dtype is not dt or not t.is_contiguous() or t.device != self._dev:
                raise ValueError(f"descriptor outputs must be contiguous {shape} {dt} tensors on {self._dev}")
        _lib.check(self._lib.fe_env_describe(self._handle, src.data_ptr(), pos.data_ptr(), self._stream()))
        return src, pos

    def check_descriptors(self, obs_src: torch.Tensor) -> None:
        """Debug check for descriptors that did not come from this env object (another rank's gathered chunk, a
        caller-filled buffer): raises ``FinEnvsNativeError`` naming the first ``obs_src`` whose window does not lie inside
        this env's log-return table.  The kernels use ``obs_src`` as a raw table offset -- a rank built with another
        W / D / L, or an uninitialised row, would otherwise end in a GPU memory fault.  Synchronises the stream."""
        src = obs_src.reshape(-1).to(device=self._dev, dtype=torch.int64).contiguous()
        first_bad = C.c_int64(-1)
        if src.numel():
            _lib.check(self._lib.fe_env_check_descriptors(self._handle, src.data_ptr(), src.numel(), C.byref(first_bad),
                                                          self._stream()))

    def geometry(self) -> torch.Tensor:
        """(D, L, W, A) int64 on the host: what two ranks must agree on before one renders the other's descriptors
        (``TrajectoryBuffer.check_geometry`` exchanges it once)."""
        D, L, _ = self.price_environments.shape
        return torch.tensor([D, L, self.num_intervals, self.num_assets], dtype=torch.int64)

    def render(self, obs_src: torch.Tensor, obs_pos: torch.Tensor, out: Optional[torch.Tensor] = None,
               check: bool = False) -> torch.Tensor:
        """Observations ``(B, W, 5A)`` (this env's ``obs_dtype``) of ANY B descriptors -- e.g. a minibatch drawn from
        a trajectory of them (PPO_agent.py:175-188 indexes minibatches out of the stored states).  ``check=True`` runs
        ``check_descriptors`` first (one host synchronisation: for descriptors of foreign origin)."""
        B, A = int(obs_src.numel()), self.num_assets
        src = obs_src.reshape(B).to(device=self._dev, dtype=torch.int64).contiguous()
        pos = obs_pos.reshape(B, A).to(device=self._dev, dtype=torch.float64).contiguous()
        if check:
            self.check_descriptors(src)
        shape = (B, self.num_intervals, 5 * A)
        if out is None:
            out = torch.empty(shape, dtype=self.obs_dtype, device=self._dev)
        elif tuple(out.shape) != shape or out.dtype is not self.obs_dtype or not out.is_contiguous() or out.device != self._dev:
            raise ValueError(f"out must be a contiguous {shape} {self.obs_dtype} tensor on {self._dev}")
        if B:  # (empty tensors have no storage to point at)
            _lib.check(self._lib.fe_env_render_n(self._handle, src.data_ptr(), pos.data_ptr(), B, out.data_ptr(), self._stream()))
        return out

    def step(self, actions: torch.Tensor, rewards_out: Optional[torch.Tensor] = None,
             dones_out: Optional[torch.Tensor] = None,
             descriptors_out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
             actions_out: Optional[torch.Tensor] = None
             ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, dict]:
        """One fused launch of TSE:277-296.  Returns (obs (N,W,5A), rewards (N,) f64,
        dones (N,) int32, info).  ``rewards_out`` / ``dones_out`` let the kernel write straight into
        caller-owned storage (e.g. a TrajectoryBuffer slot) instead of fresh tensors;
        ``descriptors_out = (obs_src (N,) int64, obs_pos (N, A) float64)`` additionally receives the returned
        observation as descriptors (``render`` turns them back into it): the ``next_states`` of the reference's loop
        for a trajectory that keeps states without their bytes (``TrajectoryBuffer(states=True).state_slot()``);
        ``actions_out`` (N, A) f32 receives a copy of the actions -- ``agent.store``'s action field written by the kernel
        that reads the actions anyway, so the policy's output can stay where the policy wrote it."""
        N, A = self._num_envs, self.num_assets
        if self._mirrors or self._terminated_view_out:
            self._sync_public_views()
        act_f64 = False
        if actions.dtype is not torch.float32:
            if self.cast_actions:
                actions = actions.float()
            elif actions.dtype is torch.float64:
                # What the reference does with float64 actions (TSE:298-302, 353-374): the share change is computed in f64
                # and long_shares / short_shares are REBOUND to f64 tensors -- for the life of the env, also under later
                # f32 actions -- which makes its commission products and the liquidation fee f64 products
                # (fe_env_step_promoted; pinned by rollout_f64_actions.npz).  This env keeps its share tensors f32 (the
                # counts are small integers: same values) and remembers the promotion -- once the step has actually been
                # launched (a call refused by the validation below must not promote the env).
                act_f64 = True
            else:
                raise ValueError(f"actions must be float32 or float64, got {actions.dtype} (the reference would compute its "
                                 "share counts in that dtype; construct the env with cast_actions=True to have them cast to float32)")
        if actions.numel() != N * A or actions.device != self._dev:
            raise ValueError(f"actions must hold {N}x{A} values on {self.device}, got {tuple(actions.shape)} on {actions.device}")
        if not actions.is_contiguous():
            actions = actions.contiguous()
        obs = self._next_obs()
        if rewards_out is None:
            rewards = torch.empty((N,), dtype=torch.float64, device=self._dev)
        else:
            rewards = rewards_out
            if rewards.dtype is not torch.float64 or rewards.numel() != N or not rewards.is_contiguous() or rewards.device != self._dev:
                raise ValueError("rewards_out must be a contiguous float64 tensor of num_envs elements on the env's device")
        if dones_out is None:
            dones = torch.empty((N,), dtype=torch.int32, device=self._dev)
        else:
            dones = dones_out
            if dones.dtype is not torch.int32 or dones.numel() != N or not dones.is_contiguous() or dones.device != self._dev:
                raise ValueError("dones_out must be a contiguous int32 tensor of num_envs elements on the env's device")
        # (a hipGraph capture of evaluate-mode steps defers the host read to the end of the replay: plain launches there)
        notify = self._flag is not None and not (self.evaluate and getattr(self, "_defer_evaluation_check", False))
        if notify and torch.cuda.is_current_stream_capturing():
            # a captured launch does not run: the host flag would be polled for a kernel that is not executing (and the
            # reference's own `.item()` read fails under capture just the same)
            raise RuntimeError("TimeSeriesEnv.step() in its default mode reads a per-step host flag (the reference's "
                               "dones[-1].item(), TSE:510; evaluate mode: TSE:531) and cannot be captured into a "
                               "hipGraph: construct the env with redraw='device', or capture through "
                               "finenvs_amd.rollout.GraphedRollout (which defers the evaluate-mode read)")
        if notify:
            self._flag_seq = seq = (self._flag_seq + 1) & (0x7FFFFFFF if self.evaluate else 0x3FFFFFFFFFFFFFFF)
        # optional trajectory outputs (validated once, whichever entry point takes them)
        src = pos = None
        if descriptors_out is not None:
            src, pos = descriptors_out
            for t, count, dt in ((src, N, torch.int64), (pos, N * A, torch.float64)):
                if t.dtype is not dt or t.numel() != count or not t.is_contiguous() or t.device != self._dev:
                    raise ValueError("descriptors_out must be contiguous (int64 (N,), float64 (N, A)) tensors on the env's device")
        if actions_out is not None:
            if act_f64:
                raise ValueError("actions_out is a float32 copy of the actions: not available with float64 actions")
            if (actions_out.dtype is not torch.float32 or actions_out.numel() != N * A or not actions_out.is_contiguous()
                    or actions_out.device != self._dev):
                raise ValueError("actions_out must be a contiguous float32 tensor of num_envs x num_assets elements on the env's device")
        outs = (actions_out.data_ptr() if actions_out is not None else None, src.data_ptr() if src is not None else None,
                pos.data_ptr() if pos is not None else None)
        base = (self._handle_v, actions.data_ptr(), obs.data_ptr(), rewards.data_ptr(), dones.data_ptr())
        if act_f64 or self.shares_promoted:
            rc = self._lib.fe_env_step_promoted(self._handle_v, actions.data_ptr(), int(act_f64), *base[2:], *outs,
                                                self._flag if notify else None, seq if notify else 0, self._stream())
        elif descriptors_out is None and actions_out is None:
            if notify:
                rc = self._lib.fe_env_step_notify(*base, self._flag, seq, self._stream())
            else:
                rc = self._step_fn(*base, self._stream())
        elif notify:
            rc = self._lib.fe_env_step_traj_notify(*base, *outs, self._flag, seq, self._stream())
        else:
            rc = self._lib.fe_env_step_traj(*base, *outs, self._stream())
        self._last_descriptors = descriptors_out  # None: the observation just returned was not recorded
        self._stepped = True
        self._generation += 1
        if rc != 0:
            _lib.check(rc)
        if act_f64:
            self.shares_promoted = True  # (the reference has just rebound long_shares / short_shares to f64 tensors, TSE:361, 375)
        info: Dict = {}
        if self.evaluate and not getattr(self, "_defer_evaluation_check", False):
            # (GraphedRollout defers this host read to the end of a K-step replay: the per-env bookkeeping already ran in
            # the kernel, and steps past an env's termination cannot change its return, TSE:526-528)
            info = self.record_evaluation_metrics(self._terminated_count(seq) if notify else None)
        elif self.redraw == "torch" and self._eval_env >= 0:
            # TSE:504-513: the eval env redraws a day from torch's global generator when it finishes
            if self._eval_env_done(seq) if notify else bool(dones[self._eval_env].item()):
                D = self.price_environments.shape[0]
                self._env_indices[self._eval_env : self._eval_env + 1] = torch.randint(0, D, (1,), device=self._dev)
        return (obs, rewards, dones, info)

    def _terminated_count(self, seq: int) -> int:
        """Evaluate mode: poll the host flag until the launch with this sequence number has finished -- its last
        workgroup stores (seq << 32) | terminated-count there -- instead of copying the counter back (TSE:531)."""
        return int(self._poll_flag(32, seq) & 0xFFFFFFFF)

    def record_evaluation_metrics(self, terminated: Optional[int] = None) -> Dict:
        """TSE:523-536.  The per-env part ran inside the step kernel; this is the
        torch.all(terminated) test and the hand-over of the returns.  ``terminated``: the count if the caller already
        has it (step() reads it from the host flag); else one device-to-host read."""
        if (int(self._counters[0].item()) if terminated is None else terminated) == self.num_envs:
            info = {"returns": self._returns.clone()}
            self.reset_evaluation_metrics()
            return info
        return {}
