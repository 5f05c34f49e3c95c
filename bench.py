#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused TimeSeriesEnv.step() hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--config 2|3|4|5|1] [--repeats R] [--no-cpu] [--no-extra]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU; envs are sharded contiguously (weak scaling: every rank owns the
config's full env count), no collective inside step(); once per chunk of `steps` steps the
compact trajectory fields are all-gathered over RCCL.

A "step" is one env.step(actions) over all envs of the workload: synthetic GBM minute
bars (65 business days -> D=64 episodes of 390 bars), a ring of 8
pre-generated uniform action tensors already resident in HBM, training mode, f64
observations (the reference's dtype), a ring of two env-owned observation buffers (the
rollout loop keeps the previous observation alive while the next is written).

The timed region is EXACTLY K steps between two (barrier + synchronize) fences; it is repeated
R times and the MEDIAN block is reported (min / max beside it), because one K-step block at
64k envs is under a millisecond.  With N > 1 there are two legs per workload: steps with the
asynchronous trajectory all-gather (the headline `value`; one whole gather per timed block) and steps without it.

After the headline workload (BASELINE.json's "64k envs" configuration) the same process runs
the larger BASELINE configs for a few steps each and reports them under "extra_configs" -- on
one GPU configs 3 and 4 (the 1M-env north-star run), on N > 1 GPUs the per-GPU shard of
config 5 (4M envs over 8 GPUs); at N = 1 also four fused-rollout legs ("fused_rollouts": K steps per launch
with an in-kernel linear / MLP / LSTM (H = 128, 1024) policy; never part of `value`).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CONFIGS = {
    # BASELINE.json configs: name -> (envs per GPU, assets, window)
    1: ("1k envs, 1 asset, window=32", 1024, 1, 32),
    2: ("64k envs, 1 asset, window=64", 65536, 1, 64),
    3: ("256k envs, 30 assets, window=64", 262144, 30, 64),
    4: ("1M envs, 30 assets, window=128", 1048576, 30, 128),
    5: ("512k envs per GPU, 30 assets, window=128 (4M over 8 GPUs)", 524288, 30, 128),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
TRAJ_T = 16             # trajectory slots on one GPU (no exchange)
AUDITION_EXTRA = 10     # ring audition: at most this many candidate buffers beyond the ring's own ...
AUDITION_BUDGET = 48 << 30  # ... and at most this many bytes of them (config 2: 10 x 168 MB; config 3: 2 x 20 GB; configs 4 / 5: none)
WATCHDOG_S = float(os.environ.get("FE_BENCH_WATCHDOG_S", "300"))  # N > 1: seconds the strong-scaling leg + DeviceGuard check may take before every rank gives up on them
TRAJ_BUDGET = 48e9      # bytes of trajectory chunks + gathered copies a rank may hold (N > 1)
SETTLE_MS = 20.0        # untimed steps before every timed phase: the box's clock transient after an idle period (run_workload.settle)
# the reference's own PyTorch-CPU path, measured in the build container
REFERENCE_CPU_QUOTED = {"value": 40497, "unit": "env-steps/s", "cores": 8,
                        "what": "hmomin/FinEnvs TimeSeriesEnv.step, torch 2.10 CPU, 65536 envs x W64, build container"}


def survey_bytes(W: int, A: int) -> int:
    """SURVEY 8(d) formula: obs write 8*W*5A + window read 8*W*4A + 84 B per sleeve + 36 B per env.
    The window-read term is served by L2 / Infinity Cache (the tables are <= 64 MB), so this is NOT
    an HBM byte count; it is reported beside the roofline, never as `achieved`."""
    return 72 * W * A + 84 * A + 36


def hbm_bytes(W: int, A: int, obs_elem: int) -> int:
    """Bytes per env-step that must cross HBM: the observation write (obs_elem*W*5A) + sleeve state
    read/write, bar, NaN probe, action (84 B per sleeve) + indices, reward, done (36 B per env).
    For f64 observations this is survey_bytes - 32*W*A; the PMC counters agree with it to 1 %."""
    return obs_elem * W * 5 * A + 84 * A + 36


def make_series(A: int):
    from finenvs_amd.data import synthetic

    return synthetic.synthetic_series(65, A, 390, 1234)


def cpu_baseline(A: int, W: int, budget_s: float = 12.0):
    """The oracle (C restatement of the reference, OpenMP) on this box's host cores,
    on a bounded sample of the same workload."""
    from oracle import fe_oracle as fo

    fo.build()
    prices, day_id, _ = make_series(A)
    P, LR, *_ = fo.tables_from_series(prices, day_id, W)
    # the GPU box gives one GPU a 16-core share of the host (os.cpu_count() reports the whole machine)
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("FE_CPU_THREADS", "16")))
    n = 65536 if A == 1 else 4096
    g = torch.Generator().manual_seed(7)
    acts = [(torch.rand((n, A), generator=g) * 2 - 1).float().numpy() for _ in range(8)]

    def timed(threads, budget):
        env = fo.OracleEnv(P, LR, W, num_envs=n, redraw_mode=1, seed=1, nthreads=threads)
        env.step(acts[0])
        t0 = time.perf_counter()
        k = 0
        while True:
            env.step(acts[k % 8])
            k += 1
            el = time.perf_counter() - t0
            if el > budget:
                return n * k / el, k, el

    rate, k, el = timed(cores, budget_s)
    rate1, _, _ = timed(1, 3.0)
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    note = "" if A == 1 else "; NOTE the multi-asset sample is 4096 envs (cache-resident on the host, flatters the CPU)"
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{k} steps of {n} envs x {A} assets x W{W} (oracle/fe_oracle.c, OpenMP {cores} threads, {el:.1f} s){note}",
            "value_1thread": rate1, "cpu_model": model, "host_cpus_visible": os.cpu_count(),
            "reference_quoted": REFERENCE_CPU_QUOTED}


def pmc_traffic(config: int, f32: bool):
    """Fallback for roofline.traffic when the live counter passes (live_pmc_traffic) are unavailable or switched
    off: HBM traffic per launch from the committed PMC passes (profiles/hbm_traffic.json), with the profiling run
    named in `traffic_source` (tools/profile_box.sh)."""
    path = os.path.join(REPO, "profiles", "hbm_traffic.json")
    key = f"config{config}" + ("_f32" if f32 else "")
    try:
        ent = json.load(open(path)).get(key)
    except Exception:  # noqa: BLE001
        ent = None
    if not ent:
        return None, None
    return ent.get("bytes_per_launch"), f"profiles/hbm_traffic.json[{key}], run tag {ent.get('tag')} (separate --pmc FETCH_SIZE / WRITE_SIZE passes, 2*FETCH+WRITE)"


_LIVE_PMC_BROKEN = []  # first failure of a live counter pass: later workloads fall back to the committed figures
_PROFILER_ENV_PREFIXES = ("ROCP_", "ROCPROF", "ROCPROFILER_", "HSA_TOOLS_", "ROCTX_")


def is_step_kernel(name: str) -> bool:
    """fe_env_kernel<OT, VEC, SINGLE, RESET_ONLY, FORM> with RESET_ONLY = false (any form of the step; the reset()
    renderer is the same template with RESET_ONLY = true)."""
    import re

    return re.search(r"fe_env_kernel<[^>]*, (?:true|false), false, \d+>", name) is not None


def step_kernel_name(W: int, A: int, f32: bool, form: int) -> str:
    """The demangled instantiation fe_env_kernel<OT, VEC, SINGLE, RESET_ONLY, FORM> a step launch dispatches to (as
    rocprofv3's kernel trace names it, minus the anonymous namespace): OT / VEC as fe_env_create picks them (16-byte
    packs unless W*5*A is odd), FORM 0 lean (plain step, or with the action copy), 1 full (trajectory DESCRIPTORS / statistics /
    evaluate mode), 2 / 3 the same with the host flag (redraw='torch')."""
    vec = 4 if f32 else 2
    while vec > 1 and (W * 5 * A) % vec:
        vec //= 2
    return f"fe_env_kernel<{'float' if f32 else 'double'}, {vec}, {'true' if A == 1 else 'false'}, false, {form}>"


def under_profiler(environ=None) -> bool:
    """True when THIS process already runs under rocprofv3 / a rocprofiler tool library.  A nested
    `rocprofv3 --pmc` would inherit the preloaded tool library: it initialises the GPU in the child launcher, which
    then exec()s the application -- the exec-after-GPU-init case this pool's boxes do not survive."""
    env = os.environ if environ is None else environ
    if env.get("ROCP_TOOL_LIBRARIES") or env.get("HSA_TOOLS_LIB") or env.get("ROCPROFILER_LIBRARY_CTOR"):
        return True
    return "rocprof" in env.get("LD_PRELOAD", "").lower()


def pmc_child_env(environ=None) -> dict:
    """Environment for the counter-pass children: the parent's, minus everything a profiler may have put there."""
    env = dict(os.environ if environ is None else environ)
    for k in list(env):
        if k.startswith(_PROFILER_ENV_PREFIXES):
            del env[k]
    pre = [x for x in env.get("LD_PRELOAD", "").replace(":", " ").split() if "rocprof" not in x.lower()]
    if pre:
        env["LD_PRELOAD"] = ":".join(pre)
    else:
        env.pop("LD_PRELOAD", None)
    env["TMPDIR"] = "/tmp"
    return env


def live_pmc_traffic(config: int, f32: bool, redraw: str, no_audition: bool = False):
    """HBM traffic per launch MEASURED IN THIS RUN: two child processes of this very script (`--pmc-child`: the same
    workload, 16 launches) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, as
    MI355X_MICROARCH.md's HBM section prescribes; both counters are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B
    requests at 64 B).  Returns (bytes per launch, description) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if _LIVE_PMC_BROKEN:
        return None, _LIVE_PMC_BROKEN[0]
    if under_profiler():
        _LIVE_PMC_BROKEN.append("already under a profiler (no nested rocprofv3)")
        return None, _LIVE_PMC_BROKEN[0]
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        _LIVE_PMC_BROKEN.append("rocprofv3 not found")
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="fe_pmc_", dir="/tmp")
    vals = {}
    try:
        for kind in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", kind, "--output-format", "csv", "-d", os.path.join(out, kind), "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--config", str(config), "--redraw", redraw]
            if f32:
                cmd.append("--obs-f32")
            if no_audition:
                cmd.append("--no-audition")
            r = subprocess.run(cmd, cwd="/tmp", env=pmc_child_env(), stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=90)
            if r.returncode != 0:
                _LIVE_PMC_BROKEN.append(f"rocprofv3 --pmc {kind} child exited with {r.returncode}")
                return None, _LIVE_PMC_BROKEN[0]
            rows = []
            for f in glob.glob(os.path.join(out, kind, "*", "*_counter_collection.csv")):
                for row in csv.DictReader(open(f)):
                    if is_step_kernel(row["Kernel_Name"]) and row["Counter_Name"] == kind:
                        rows.append(float(row["Counter_Value"]))
            if len(rows) < 8:
                return None, f"no {kind} rows for the step kernel"
            rows = rows[len(rows) // 4:]  # drop the warm-up launches
            vals[kind] = (sum(rows) / len(rows), len(rows))
    except Exception as exc:  # noqa: BLE001  (incl. the 90 s timeout: never try again in this run)
        _LIVE_PMC_BROKEN.append(f"{type(exc).__name__}: {exc}")
        return None, _LIVE_PMC_BROKEN[0]
    finally:
        shutil.rmtree(out, ignore_errors=True)
    traffic = (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
    return traffic, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of bench.py "
                     f"({vals['FETCH_SIZE'][1]} / {vals['WRITE_SIZE'][1]} launches averaged; 2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes)")


def pmc_child(args):
    """`--pmc-child`: the headline workload's step kernel, a few launches, nothing else (what the rocprofv3 counter
    passes of live_pmc_traffic profile)."""
    import finenvs_amd

    name, N, A, W = CONFIGS[args.config]
    prices, day_id, _ = make_series(A)
    obs_bytes = N * W * 5 * A * (4 if args.obs_f32 else 8)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw=args.redraw, seed=1234,
                                    obs_buffers=2 if 2 * obs_bytes < 200e9 else 1, obs_dtype=torch.float32 if args.obs_f32 else torch.float64)
    if not args.no_audition:
        env.audition_ring(AUDITION_EXTRA, AUDITION_BUDGET)
    g = torch.Generator(device="cuda:0").manual_seed(7)
    actions = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
    env.reset()
    rew = torch.empty((N,), dtype=torch.float64, device="cuda:0")
    done = torch.empty((N,), dtype=torch.int32, device="cuda:0")
    act = torch.empty((N, A), dtype=torch.float32, device="cuda:0")
    for i in range(16):  # the timed loop's form of the step: trajectory outputs (+ the host flag with --redraw torch)
        env.step(actions[i % 8], rewards_out=rew, dones_out=done, actions_out=act)
    torch.cuda.synchronize()


class Dist:
    """torch.distributed plumbing; world == 1 needs none of it."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus and self.world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        # rehearsal knobs for a one-GPU box (never set by the driver): all ranks on device 0 + gloo
        if os.environ.get("FE_BENCH_SINGLE_DEVICE") == "1":
            self.local_rank = 0
        self.backend = os.environ.get("FE_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        # a launcher that narrows each rank's view to its own GPU (HIP_VISIBLE_DEVICES per rank) leaves LOCAL_RANK pointing
        # past the one device the rank can see
        ndev = torch.cuda.device_count()
        if ndev and self.local_rank >= ndev:
            self.local_rank %= ndev
        torch.cuda.set_device(self.local_rank)
        self.dev = f"cuda:{self.local_rank}"
        self.dist = None
        # rehearsal knob (never set by the driver): run the N > 1 code path -- RCCL process group, trajectory all-gather
        # legs, config-5 shard -- with the ONE rank a one-GPU box allows
        self.multi = self.world > 1 or os.environ.get("FE_BENCH_FORCE_DIST") == "1"
        if self.multi:
            import datetime

            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            kw = {"timeout": datetime.timedelta(minutes=5)}
            # RCCL's own account of what it chose (algorithm / protocol / channels) goes to a per-process file that rank 0
            # summarises into multi_gpu.rccl (NCCL_DEBUG_FILE keeps it off stdout / stderr); FE_BENCH_RCCL_DEBUG=0 switches it off
            self.rccl_log = None
            if self.backend == "nccl" and os.environ.get("FE_BENCH_RCCL_DEBUG", "1") != "0":
                self.rccl_log = f"/tmp/fe_bench_rccl_{os.getpid()}.log"
                os.environ["NCCL_DEBUG"] = "INFO"
                os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH,TUNING,ENV"  # (not COLL: no per-call logging inside timed regions)
                os.environ["NCCL_DEBUG_FILE"] = self.rccl_log
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(self.dev), **kw)
            else:
                dist.init_process_group(self.backend, **kw)
            self.dist = dist

    def rccl_report(self):
        """What RCCL said about itself on this rank (NCCL_DEBUG=INFO into a file): version, topology / channel lines and
        every distinct algorithm / protocol decision it logged.  None when the backend is not RCCL or logging is off."""
        path = getattr(self, "rccl_log", None)
        if not path:
            return None
        path = os.environ.get("NCCL_DEBUG_FILE", path)
        try:
            with open(path, errors="replace") as f:
                lines = [ln.strip() for ln in f]
        except OSError as exc:
            return {"log": path, "error": f"{type(exc).__name__}: {exc}"}
        import re

        def pick(pat, limit):
            seen, out = set(), []
            for ln in lines:
                if re.search(pat, ln, re.I):
                    body = re.sub(r"^\S+:\d+:\d+ \[\d+\] NCCL INFO ", "", ln)[:200]
                    if body not in seen:
                        seen.add(body)
                        out.append(body)
                    if len(out) >= limit:
                        break
            return out

        return {"log_lines": len(lines),
                "version": pick(r"RCCL version|HIP version|ROCm version", 3),
                # with more than one rank the TUNING subsystem logs its choice per collective size ("... Bytes -> Algo A proto P")
                "algorithm_protocol": pick(r"-> Algo|Algo \d|algorithm|protocol|Bytes ->", 12) or
                                      ["(none logged: RCCL logs no algorithm / protocol choice for a single rank)"],
                "topology": pick(r"Pattern \d|coll channels|P2P Chunksize|intraNodeP2pSupport|nNodes|XGMI|=== System", 10),
                "transport": pick(r" via |Connected all", 6),
                "rings": pick(r"NCCL INFO Ring 0*0 :", 2),
                "env": pick(r"NCCL_[A-Z_]+ set|RCCL_[A-Z_]+ set", 8)}

    def barrier(self):
        """The fence's barrier.  Over RCCL it is a one-element all-reduce ENQUEUED behind this rank's work (stream-ordered: it
        completes once every rank's stream has reached it); the caller's torch.cuda.synchronize() then waits for work and
        barrier together -- one host round trip per fence instead of dist.barrier()'s own device synchronise plus ours.
        Over gloo (CPU rehearsals) a plain host barrier."""
        if self.dist is None:
            return
        if self.backend == "nccl":
            if getattr(self, "_bar", None) is None:
                self._bar = torch.zeros((1,), dtype=torch.int32, device=self.dev)
            self.dist.all_reduce(self._bar)
        else:
            self.dist.barrier()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok: bool) -> bool:
        """True iff every rank says ok (so that no rank enters a collective loop alone)."""
        if self.dist is None:
            return ok
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)


class KernelTrain:
    """Back-to-back launches of the step kernel straight through the C ABI -- the SAME entry point and FORM the timed loop's
    env.step() dispatches to: fe_env_step_traj with the action copy (FORM 0: rewards / dones / action copy into trajectory slots
    need no more than the lean form) or, when this env polls a host flag (redraw='torch' on the rank that owns the evaluation
    env), fe_env_step_traj_notify (FORM 2) -- with preallocated outputs and
    no per-step Python work, so the queue never drains; ONE pair of HIP events on the launch stream brackets the train
    (torch's current stream is the stream the C ABI launches on).  interval = kernel + launch boundary.  Uses the env's own
    observation ring (the HBM / MALL regime of the timed region).  The host flag is written, never read, inside a train: the
    evaluation env simply keeps its day (a redraw is host work outside the kernel)."""

    def __init__(self, env, actions):
        N, dev = env.num_envs, env._dev
        self.env = env
        self.rew = torch.empty((N,), dtype=torch.float64, device=dev)
        self.done = torch.empty((N,), dtype=torch.int32, device=dev)
        self.act = torch.empty((N, env.num_assets), dtype=torch.float32, device=dev)  # agent.store's action copy
        self.aptr = [a.data_ptr() for a in actions]
        self.notify = env._flag is not None and not env.evaluate
        self.form = 2 if self.notify else 0

    def run(self, k: int) -> float:
        """ms per launch of one train of k launches."""
        from finenvs_amd import _lib as _fl

        env = self.env
        stream = torch.cuda.current_stream().cuda_stream
        obs = [t.data_ptr() for t in env._obs_ring]  # read per train: the placement audition may have replaced ring members
        h, rp, dp, ap, aptr, nb = env._handle_v, self.rew.data_ptr(), self.done.data_ptr(), self.act.data_ptr(), self.aptr, len(obs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rc = 0
        if self.notify:
            fn, flag, seq = env._lib.fe_env_step_traj_notify, env._flag, env._flag_seq
            e0.record()
            for i in range(k):
                seq = (seq + 1) & 0x3FFFFFFFFFFFFFFF
                rc = fn(h, aptr[i % 8], obs[i % nb], rp, dp, ap, None, None, flag, seq, stream) or rc  # (keeps a failure's code)
            e1.record()
            env._flag_seq = seq
        else:
            fn = env._lib.fe_env_step_traj
            e0.record()
            for i in range(k):
                rc = fn(h, aptr[i % 8], obs[i % nb], rp, dp, ap, None, None, stream) or rc
            e1.record()
        torch.cuda.synchronize()
        env._generation += k
        if rc:
            _fl.check(rc)
        return e0.elapsed_time(e1) / k


def launches_per_train(est_step_ms: float, steps: int) -> int:
    """Launches one HIP-event pair brackets: at least 40 (and `steps`), enough for ~10 ms of GPU time, at most 400 -- a short
    train of a 30 us kernel is dominated by its first launches (they start on a GPU whose XCDs wake up staggered, DESIGN.md
    section 5)."""
    return int(min(400, max(40, steps, round(10.0 / max(est_step_ms, 1e-6)))))


def auto_repeats(repeats: int, steps: int, est_step_s: float) -> int:
    if repeats > 0:
        return repeats
    # enough K-step blocks for ~0.15 s of timed work, between 5 and 40
    return int(min(40, max(5, round(0.15 / max(steps * est_step_s, 1e-9)))))


def run_workload(config: int, args, D: Dist, steps: int, warmup: int, repeats: int, with_cpu: bool, redraw: str = None):
    """Build the env of one BASELINE config on this rank, time it, tear it down.  Returns the result dict
    (on every rank; only rank 0 prints).  `redraw`: the evaluation env's redraw mode (default: --redraw)."""
    redraw = redraw or args.redraw
    import finenvs_amd
    from finenvs_amd import _lib as _fl
    from finenvs_amd.trajectory import TrajectoryBuffer

    world, rank, dev = D.world, D.rank, D.dev
    name, n_per_gpu, A, W = CONFIGS[config]
    obs_elem = 4 if args.obs_f32 else 8
    env = traj = actions = None
    err = None
    try:
        prices, day_id, _ = make_series(A)
        # config 4's observation is 153.6 GB: a single env-owned buffer (SURVEY section 7 "Capacity")
        obs_bytes = n_per_gpu * W * 5 * A * obs_elem
        obs_buffers = 2 if 2 * obs_bytes < 200e9 else 1
        # the previous workload's buffers (and those of its rocprofv3 child processes) are given back asynchronously:
        # a 150 GB allocation can take seconds to reappear as free memory after its owner has gone
        need = obs_buffers * obs_bytes + (6 << 30)
        t_wait = time.perf_counter()
        while torch.cuda.mem_get_info(D.dev)[0] < need and time.perf_counter() - t_wait < 60.0:
            gc.collect()
            torch.cuda.empty_cache()
            time.sleep(0.5)
        env = finenvs_amd.TimeSeriesEnv(
            prices=prices, day_id=day_id, num_intervals=W, num_envs=n_per_gpu * world, rank=rank, world_size=world,
            device_id=D.local_rank, redraw=redraw, seed=1234, obs_buffers=obs_buffers,
            obs_dtype=torch.float32 if args.obs_f32 else torch.float64)
        N = env.num_envs
        g = torch.Generator(device=dev).manual_seed(7 + rank)
        actions = [(torch.rand((N, A), generator=g, device=dev) * 2 - 1).float() for _ in range(8)]
        # Compact trajectory fields live in a device buffer; the step kernel writes rewards, dones and its copy of the
        # actions straight into slot t (fe_env_step_traj), so storing a step costs no extra launch.  (Round 2 found the
        # earlier scheme -- actions pre-stored in the slots -- to read 256 KB of COLD memory per step once a chunk is longer
        # than a few slots: +11 us per 30 us step at 64k envs, tools/cold_slots.py; a policy's output is hot.)  With N > 1 a chunk is `steps` slots long (capped by memory), i.e.
        # one timed block fills exactly one chunk, and the chunk filled by a block is all-gathered asynchronously
        # at the first step of the NEXT block: every timed block contains one whole all-gather (issued at its
        # start, drained by the fence at its end) that has the block's own steps to hide behind.
        T = TRAJ_T
        if D.multi:
            per_step = N * (8 + 4 * A + 4)
            T = max(1, min(steps, int(TRAJ_BUDGET // (per_step * (2 + 2 * world)))))
        traj = TrajectoryBuffer(T, N, A, device=dev)
        torch.cuda.synchronize()
    except Exception as exc:  # noqa: BLE001  (allocation / construction: before any collective)
        err = f"{type(exc).__name__}: {exc}"
    if not D.all_ok(err is None):
        del env, traj, actions
        gc.collect()
        torch.cuda.empty_cache()
        return {"workload": name, "config": config, "error": err or "another rank failed to build this workload"}

    gather = [D.multi]  # mutable: the timed legs flip it
    gather_due = [0]    # steps until the deferred collective of the chunk just handed over is started

    def one_step(i):
        if traj.full():  # checked at the START of a step: a chunk completed by a block goes out with the next block
            if gather[0]:
                # chunks are switched now; the collective itself is started two steps later, when the GPU has the new
                # chunk's first steps queued -- starting it costs the host 20 - 30 us (TrajectoryBuffer.all_gather_async).
                # It overlaps the following steps and is waited for before its chunk is reused / by the closing fence.
                traj.all_gather_async(defer=True)
                gather_due[0] = 2
            else:
                traj.clear()
        # the "policy" hands over its output buffer (a ring of 8 pre-generated action tensors: hot, as a policy's fresh
        # output is); agent.store's fields -- actions, rewards, dones -- are written into slot t by the step kernel itself
        a, r, d = traj.next_slot()
        obs, rew, done, _ = env.step(actions[i % 8], rewards_out=r, dones_out=d, actions_out=a)
        if gather_due[0]:
            gather_due[0] -= 1
            if gather_due[0] == 0:
                traj.issue_deferred()
        return obs

    roll = None
    if args.graph:
        if D.multi or steps % 8:
            sys.exit("--graph: single GPU only, and --steps must be a multiple of 8")
        warmup = (warmup + 7) // 8 * 8
        from finenvs_amd.rollout import GraphedRollout

        roll = GraphedRollout(env, lambda obs, k: actions[k % 8], 8)
        run_steps = lambda n: [roll.run() for _ in range(n // 8)]  # noqa: E731
    else:
        env.reset()
        run_steps = lambda n: [one_step(i) for i in range(n)]  # noqa: E731

    def fence():
        """drain + barrier + synchronize.  N > 1 over RCCL: the barrier is enqueued behind this rank's steps and gathers, so
        ONE synchronize covers "my work is done" and "everybody has arrived"; over gloo the host barrier needs this rank's
        work finished first."""
        traj.drain()  # outstanding gathers belong to the timed region
        if D.dist is not None and D.backend != "nccl":
            torch.cuda.synchronize()
        D.barrier()  # (one process: a no-op)
        torch.cuda.synchronize()

    trace = os.environ.get("FE_BENCH_TRACE") == "1"  # stderr: where a timed block's wall time goes (host issue / drain / fence)

    train = [None, 0]  # KernelTrain of this env (built once its launch rate is known), launches per train

    def timed_blocks(r):
        """r blocks of exactly `steps` steps, each bracketed by a full fence (drain + synchronize + barrier + synchronize) on
        both sides, as the driver's contract says; the block's time is the MAX over ranks.  Straight after each block's
        closing fence ONE kernel train runs (KernelTrain: back-to-back C-ABI launches of the loop's own kernel form between
        a pair of HIP events): block and train alternate, so `roofline.kernel_ms` and `ms_per_step` are medians over the
        same stretch of time, on the same buffers, in the same clock regime.  Returns (block seconds, train ms per launch)."""
        out, trains = [], []
        for _ in range(r):
            fence()
            t0 = time.perf_counter()
            run_steps(steps)
            if trace:
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            fence()
            out.append(D.max_over_ranks(time.perf_counter() - t0))
            if trace and rank == 0:
                t3 = time.perf_counter()
                print(f"[trace] issue {(t1 - t0) * 1e6:8.1f} us  drain {(t2 - t1) * 1e6:8.1f} us  fence+max {(t3 - t2) * 1e6:8.1f} us  "
                      f"({steps} steps)", file=sys.stderr, flush=True)
            if train[0] is not None:
                trains.append(train[0].run(train[1]))
        return out, trains

    run_steps(warmup)
    fence()
    t0 = time.perf_counter()
    run_steps(min(steps, 8))
    fence()
    est = D.max_over_ranks(time.perf_counter() - t0) / min(steps, 8)
    R = auto_repeats(repeats, steps, est)
    if roll is None:
        train[0], train[1] = KernelTrain(env, actions), launches_per_train(est * 1e3, steps)
    total_envs = env.global_num_envs if world > 1 else N

    def settle():
        """Untimed steps for SETTLE_MS of GPU time, straight before a timed phase.  After an idle period (env construction,
        allocations, the audition's event waits) this pool's GPUs run the same kernel at its settled duration for ~1 ms, then
        15 - 25 % longer for several ms, and are back after ~14 ms (profiles/r04_microbench/idle_transient.txt; identical for
        C-ABI trains and the Python loop): a 0.6 ms block timed inside that window measures the box's clock transient.  A
        rollout runs for minutes; the timed region is to describe that regime.  Same count on every rank, no collectives."""
        n = int(min(2000, max(0, round(SETTLE_MS * 1e-3 / max(est, 1e-9)))))
        n = (n + 7) // 8 * 8 if roll is not None else n
        keep = gather[0]
        gather[0] = False
        run_steps(n)
        gather[0] = keep
        return n

    settle_steps = settle()

    # The ring AS THE ALLOCATOR HANDED IT OUT is timed first, with the headline's own loop and fences (`as_allocated`: wall
    # value + kernel interval); then ring mode's placement audition (a product feature of the ring)
    # tries a BOUNDED number of further candidate buffers (AUDITION_EXTRA / AUDITION_BUDGET) and keeps the fastest -- the
    # headline runs on the auditioned ring.  Where the audition keeps the ring as it was (no candidate fits: configs 4 and 5;
    # none is > 3 % faster: usually config 2) both figures are the same timed blocks.
    as_allocated = None
    reuse_blocks = None  # the as-allocated blocks ARE the headline's when the audition kept the ring (on every rank)
    if not args.no_audition and roll is None:
        if D.multi:
            gather[0] = True
            fence()
            traj.clear()
            run_steps(T)  # one full chunk: the first timed block has something to gather
        aa_blocks, aa_trains = timed_blocks(R)
        aa_kern = statistics.median(aa_trains)
        ring_before = [t.data_ptr() for t in env._obs_ring]
        env.audition_ring(AUDITION_EXTRA, AUDITION_BUDGET)
        changed = [t.data_ptr() for t in env._obs_ring] != ring_before
        aa_block = statistics.median(aa_blocks)
        as_allocated = {"value": total_envs * steps / aa_block, "ms_per_step": aa_block / steps * 1e3,
                        "ms_per_step_min": min(aa_blocks) / steps * 1e3, "ms_per_step_max": max(aa_blocks) / steps * 1e3,
                        "blocks": len(aa_blocks), "kernel_ms": aa_kern,
                        "frac_of_8TBps_wall": hbm_bytes(W, A, obs_elem) * N / (aa_block / steps) / 1e9 / HBM_PEAK_GBPS,
                        "frac_of_8TBps_kernel": hbm_bytes(W, A, obs_elem) * N / (aa_kern * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "ring_changed_by_audition": changed,
                        "what": "the same timed loop and fences as `value`, on the observation ring as the allocator handed it out "
                                "(before the placement audition)"}
        if D.all_ok(not changed):
            reuse_blocks = (aa_blocks, aa_trains)
            as_allocated["same_run_as_value"] = True
        # warm the (possibly new) buffers: first touch, translations.  Unconditional and without collectives: the ranks
        # of an N > 1 run may keep different buffers, but must stay in step
        gather[0] = False
        fence()
        traj.clear()
        env.reset()
        run_steps(max(min(warmup, steps), 4))
        fence()
        traj.clear()
        gather[0] = D.multi
        settle()
    legs = {}
    if D.multi:
        gather[0] = True
        fence()
        traj.clear()
        run_steps(T)  # one full chunk: the first timed block has something to gather
        legs["with_all_gather"], kern = reuse_blocks if reuse_blocks is not None else timed_blocks(R)
        fence()
        gather[0] = False
        traj.clear()
        settle()
        legs["no_all_gather"], _ = timed_blocks(R)
        head = legs["with_all_gather"]
        # Gather-only leg: the all-gather of one filled chunk ALONE (issue + drain between fences, no env steps beside
        # it), so that "with / without all-gather" above can be decomposed: an exposed gather shows up as
        # with - without ~ gather_only, a hidden one as with ~ without.  SURVEY section 5 expects the direct
        # (all-links) algorithm on the fully connected xGMI node; `bus_GBps` = inbound bytes per GPU / time.
        fence()
        traj.clear()
        run_steps(T)  # a full chunk of real data
        gather_only = []
        for _ in range(max(R, 5)):
            fence()
            t0 = time.perf_counter()
            traj.mark_filled(T)  # the chunk counts as full again (same bytes; the data is not the point)
            traj.all_gather_async()
            traj.drain()
            torch.cuda.synchronize()
            gather_only.append(D.max_over_ranks(time.perf_counter() - t0))
        fence()
        traj.clear()
    else:
        legs["single_gpu"], kern = reuse_blocks if reuse_blocks is not None else timed_blocks(R)
        head = legs["single_gpu"]
    block = statistics.median(head)

    # Kernel duration for the roofline: the trains that alternated with the headline's timed blocks (timed_blocks).  --graph
    # has no C-ABI train of its own (the graph replays the lean form): three trains of the full form after the blocks.
    if not kern:
        train[0], train[1] = KernelTrain(env, actions), launches_per_train(est * 1e3, steps)
        settle()
        kern = [train[0].run(train[1]) for _ in range(3)]
    k2 = train[1]
    kern_ms = statistics.median(kern)
    form = train[0].form

    Bh = hbm_bytes(W, A, obs_elem)
    Bs = survey_bytes(W, A) - (4 * W * 5 * A if args.obs_f32 else 0)
    l2_read = (16 if args.obs_f32 else 32) * W * A  # window re-read per env-step, served by L2 / Infinity Cache
    achieved = Bh * N / (kern_ms * 1e-3) / 1e9
    traffic, traffic_source = pmc_traffic(config, args.obs_f32)
    res = {
        "workload": name, "config": config,
        "value": total_envs * steps / block,
        "ms_per_step": block / steps * 1e3,
        "steps": steps, "warmup": warmup,
        "envs_per_gpu": N, "num_assets": A, "window": W, "obs_buffers": obs_buffers,
        "obs_ring_audition": getattr(env, "obs_audition", None),
        "as_allocated": as_allocated,
        "launch": env.launch_info(),
        "launch_mode": "hipGraph x8 steps" if args.graph else "eager, one launch per step",
        "settle_steps": settle_steps,
        "repeats": {name_: {"blocks": len(v), "ms_per_step_median": statistics.median(v) / steps * 1e3,
                            "ms_per_step_min": min(v) / steps * 1e3, "ms_per_step_max": max(v) / steps * 1e3,
                            "value_median": total_envs * steps / statistics.median(v)}
                    for name_, v in legs.items()},
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
            # the instantiation the trains AND the timed loop launch (the row to look up in profiles/*_kernel_stats.csv):
            # env.step with rewards / dones / action copy into trajectory slots = fe_env_step_traj on the lean form (FORM 0), or --
            # redraw='torch' on the rank that owns the evaluation env -- fe_env_step_traj_notify (FORM 2: same arithmetic, the evaluation env's
            # tile first)
            "kernel": step_kernel_name(W, A, args.obs_f32, form),
            "timed_loop_kernel": step_kernel_name(W, A, args.obs_f32, 0 if roll is not None else form),  # (--graph: plain env.step, the lean form)
            "kernel_ms": kern_ms,
            "kernel_ms_regime": "median of the trains that ALTERNATE with the timed blocks (block, train, block, train ...): "
                                "back-to-back C-ABI launches of the loop's own kernel form, one HIP-event pair on the launch stream "
                                "around each train (kernel + launch boundary), same observation ring, same clock regime as the blocks",
            "kernel_ms_runs": kern if len(kern) <= 8 else [min(kern), statistics.median(kern), max(kern)],
            "kernel_launches_per_run": k2,
            # for tools/summarize_prof.py: how the LAST launches of this instantiation in the run are laid out
            "kernel_train_layout": {"repeats": len(kern), "loop_launches_per_block": steps, "train_launches": k2,
                                    "order": "block, train, block, train, ..."} if roll is None else None,
            # `achieved` counts only bytes that must cross HBM (observation write + state + outputs):
            "hbm_bytes_per_env_step": Bh, "units_per_launch": N,
            # the window re-read of the SURVEY 8(d) formula comes from L2 / Infinity Cache, reported apart:
            "l2_read_bytes_per_env_step": l2_read, "l2_GBps": l2_read * N / (kern_ms * 1e-3) / 1e9,
            "survey_8d_bytes_per_env_step": Bs,
            # whole-job check: HBM bytes / wall ms_per_step (must stay below the peak as well)
            "achieved_wall": Bh * N / (block / steps) / 1e9,
        },
    }
    if D.multi:
        forms = [None] * D.dist.get_world_size()
        D.dist.all_gather_object(forms, int(form))
        res["multi_gpu"] = {
            "ranks_seen": D.dist.get_world_size(), "collective_backend": D.backend,
            # the step kernel's FORM per rank: only the rank that owns the evaluation env polls the host flag (FORM 2 with
            # redraw='torch'); `roofline.kernel` above is rank 0's, `value` is the max over ranks
            "kernel_form_by_rank": forms,
            "trajectory_slots": T, "all_gather_every_steps": T,
            "all_gather_issue": "asynchronous; chunks switch at the first step after a chunk is full, the collective is started two steps later "
                                "(its host cost then hides behind queued steps); drained inside the timed block",
            "packed_bytes_per_rank_per_chunk": traj._nbytes,
            "gathered_bytes_per_rank_per_chunk": traj._nbytes * world,
            "value_with_all_gather": res["repeats"]["with_all_gather"]["value_median"],
            "value_no_all_gather": res["repeats"]["no_all_gather"]["value_median"],
            # decomposition: one chunk's all-gather alone, and what it costs per step if fully exposed
            "gather_only_ms": statistics.median(gather_only) * 1e3,
            "gather_only_ms_min": min(gather_only) * 1e3, "gather_only_ms_max": max(gather_only) * 1e3,
            "gather_only_ms_per_step_if_exposed": statistics.median(gather_only) * 1e3 / T,
            "gather_only_inbound_GBps_per_gpu": traj._nbytes * (D.dist.get_world_size() - 1) / statistics.median(gather_only) / 1e9,
            "exposed_ms_per_step": res["repeats"]["with_all_gather"]["ms_per_step_median"] - res["repeats"]["no_all_gather"]["ms_per_step_median"],
            "rccl": D.rccl_report(),
        }
        # DESIGN.md section 7's prediction evaluated for THIS world size and workload (written in round 4, before RCCL had
        # ever run with more than one rank here): what the gather needs from xGMI and whether a block can hide it
        ws = D.dist.get_world_size()
        per_step = N * (8 + 4 * A + 4)
        inbound_chunk = traj._nbytes * (ws - 1)
        block_ms_no = res["repeats"]["no_all_gather"]["ms_per_step_median"] * steps
        busbw = (210.0, 300.0)  # GB/s: RCCL large-message all-gather on a fully connected 8-GPU xGMI node (assumption)
        pred = [inbound_chunk / (b * 1e9) * 1e3 for b in (busbw[1], busbw[0])] if ws > 1 else [0.0, 0.0]
        res["multi_gpu"]["prediction"] = {
            "source": "DESIGN.md section 7", "inbound_bytes_per_step_per_gpu": per_step * (ws - 1),
            "inbound_GBps_needed_per_gpu": per_step * (ws - 1) / (block_ms_no / steps * 1e-3) / 1e9 if ws > 1 else 0.0,
            "xgmi_inbound_peak_GBps_per_gpu": 7 * 153.0, "assumed_rccl_busbw_GBps": list(busbw),
            "predicted_gather_only_ms": pred, "block_ms_without_gather": block_ms_no,
            "predicted_hidden_behind_the_block": bool(pred[1] <= block_ms_no),
            "gather_hbm_share_of_block": (traj._nbytes * (ws + 1)) / max(Bh * N * steps, 1),
        }
    if with_cpu and rank == 0:
        res["cpu_baseline"] = cpu_baseline(A, W)
    fence()
    del env, traj, actions, roll
    gc.collect()
    torch.cuda.empty_cache()
    return res


def two_stream_leg(args, steps: int):
    """The headline workload as TWO contiguous shards (rank 0 / 1 of 2: the same envs, the evaluation env in the second)
    stepped on two HIP streams.  Envs are independent, so a rollout loop that evaluates its policy per shard (a
    double-buffered sampler, examples/double_buffered_rollout.py) lets one shard's start-up chain and launch boundary
    overlap the other shard's store stream across steps.  Launches go through the C ABI (pre-generated actions, as in the
    headline's kernel-interval loop).  Never part of `value`."""
    import finenvs_amd

    name, N, A, W = CONFIGS[2]
    prices, day_id, _ = make_series(A)
    dev = "cuda:0"
    parts, streams = [], [torch.cuda.Stream(), torch.cuda.Stream()]
    for r in range(2):
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, rank=r, world_size=2,
                                        redraw="device", seed=1234, obs_buffers=2)
        n = env.num_envs
        g = torch.Generator(device=dev).manual_seed(7 + r)
        acts = [(torch.rand((n, A), generator=g, device=dev) * 2 - 1).float() for _ in range(8)]
        env.reset()
        parts.append((env, acts, torch.empty((n,), dtype=torch.float64, device=dev), torch.empty((n,), dtype=torch.int32, device=dev)))
    k2 = max(min(max(steps, 20), 400), 200)  # (a 27 us step: 200 launch pairs ~ 5 ms per run)
    times = []
    for rep in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s_ in streams:
            s_.wait_event(e0)
        for k in range(k2 if rep else 4 * k2):  # the discarded first run also carries the GPU past its post-idle clock transient
            for (env, acts, rew, done), s_ in zip(parts, streams):
                rc = env._step_fn(env._handle_v, acts[k % 8].data_ptr(), env._obs_ring[k % 2].data_ptr(), rew.data_ptr(), done.data_ptr(),
                                  s_.cuda_stream)
        for s_ in streams:
            torch.cuda.current_stream().wait_stream(s_)
        e1.record()
        torch.cuda.synchronize()
        from finenvs_amd import _lib as _fl

        _fl.check(rc)
        if rep:
            times.append(e0.elapsed_time(e1) / k2)
    ms = statistics.median(times)
    Bh = hbm_bytes(W, A, 8)
    del parts
    torch.cuda.empty_cache()
    return {"workload": name, "what": "two contiguous shards (rank 0 / 1 of 2) of the same 65 536 envs on two HIP streams, C-ABI launches, "
                                      "HIP events around the whole loop", "envs": N, "ms_per_step_all_envs": ms,
            "value": N / ms * 1e3, "unit": "env-steps/s", "achieved_GBps": Bh * N / (ms * 1e-3) / 1e9,
            "frac_of_8TBps": Bh * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "launches_per_run": 2 * k2}


REDRAW_CONTRACT = ("eval_redraw='torch' (the class default AND the headline): the reference's own stream -- one torch.randint on the "
                   "global generator per finished evaluation episode, decided by a per-step host read of dones[-1] (TSE:504-513; here a "
                   "coherent host flag the kernel writes, fe_env_step_traj_notify, not a device-to-host copy); the reference-generated "
                   "golden fixtures and its call log (rng_calls.npz) pin this mode.  eval_redraw='device' (timed as `device_redraw`, "
                   "same block protocol): the redraws come from a Philox4x32-10 counter inside the step kernel -- THIS BUILD's contract "
                   "(finenvs_amd/rng.py; the generator is pinned to Random123's known-answer vectors and to the oracle's restatement, "
                   "the day SEQUENCE has no counterpart in the reference), no host synchronisation: what hipGraph capture and the fused "
                   "rollouts need.")


def reference_semantics_leg(args, steps: int, repeats: int = 0):
    """What a drop-in caller of the reference's loop gets (examples/time_series/PPO_LSTM_training_SPY.py:22-30): the CLASS
    DEFAULTS -- fresh observation / reward / done tensors per step (obs_buffers=0), redraw='torch' with the reference's
    per-step host read of the evaluation env's done flag -- on the headline workload (config 2), timed with the headline's
    block protocol (R blocks of exactly `steps` steps between synchronising fences after SETTLE_MS of untimed steps, median
    block).  Never part of `value`."""
    import finenvs_amd

    name, N, A, W = CONFIGS[2]
    prices, day_id, _ = make_series(A)
    torch.manual_seed(1234)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, device_id=0)
    assert env.obs_buffers == 0 and env.redraw == "torch" and env.obs_dtype == torch.float64
    g = torch.Generator(device="cuda:0").manual_seed(7)
    actions = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
    states = env.reset()
    for i in range(20):
        states, _, _, _ = env.step(actions[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8):
        states, _, _, _ = env.step(actions[i % 8])
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 8
    R = auto_repeats(repeats, steps, est)
    for i in range(int(SETTLE_MS * 1e-3 / est)):  # the clock transient after the idle period of construction (run_workload.settle)
        states, _, _, _ = env.step(actions[i % 8])
    blocks = []
    for _ in range(R):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            states, rew, done, _ = env.step(actions[i % 8])
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
    med = statistics.median(blocks)
    del env, states
    torch.cuda.empty_cache()
    return {"workload": name, "what": "class defaults: obs_buffers=0 (fresh tensors per step), redraw='torch' (per-step host read, TSE:510), "
                                      "f64 observations, eager env.step(actions) loop; the headline's block protocol",
            "value": N * steps / med, "unit": "env-steps/s", "ms_per_step": med / steps * 1e3,
            "ms_per_step_min": min(blocks) / steps * 1e3, "ms_per_step_max": max(blocks) / steps * 1e3, "steps": steps,
            "blocks": R}


STRONG_TOTAL_ENVS = 65536  # BASELINE.json's metric: "env-steps/sec at 64k envs, 1/2/4/8 MI355X" read as ONE 64k-env job


def strong_scaling_leg(args, D: Dist, steps: int, warmup: int, world: int = None, rank: int = None):
    """The STRONG-scaling reading of the metric: 65 536 envs IN TOTAL, sharded contiguously over the world (8 GPUs: 8 192 envs
    per GPU ~ 5 us of HBM time per step -- launch-bound, where the fused / graphed forms earn their keep).  Two launch modes:
    `eager` (env.step per step, --redraw's mode, trajectory slots written by the kernel) and `graph_k8` / `graph_k32` (rollout.GraphedRollout,
    8 / 32 steps per hipGraph replay, redraw='device' as capture requires); with N > 1 each with and without the trajectory
    all-gather (one packed chunk per `steps` eager steps resp. per 8-step replay, asynchronous, double-buffered).
    `world` / `rank`: emulate one rank's shard of a larger world on THIS GPU without collectives (the N = 1 run's preview of
    the per-GPU step time at 2 / 4 / 8 GPUs: the measured basis of DESIGN.md section 7's strong-scaling rows)."""
    import finenvs_amd
    from finenvs_amd.rollout import GraphedRollout
    from finenvs_amd.trajectory import TrajectoryBuffer

    emulated = world is not None
    w = world if emulated else D.world
    r = rank if emulated else D.rank
    dev = D.dev
    _, _, A, W = CONFIGS[2]
    prices, day_id, _ = make_series(A)
    gathering = D.multi and not emulated
    out = {"total_envs": STRONG_TOTAL_ENVS, "world": w, "emulated_on_one_gpu": emulated}

    def fence(trajs=()):
        for t in trajs:
            t.drain()
        if D.dist is not None and not emulated:
            if D.backend != "nccl":
                torch.cuda.synchronize()
            D.barrier()
        torch.cuda.synchronize()

    def blocks_of(run_block, n_steps, trajs=(), r_blocks=None):
        run_block()
        fence(trajs)
        t0 = time.perf_counter()
        run_block()
        fence(trajs)
        est = (time.perf_counter() - t0) / n_steps
        est = est if emulated else D.max_over_ranks(est)
        R = r_blocks or auto_repeats(args.repeats, n_steps, est)
        for _ in range(int(min(4000, SETTLE_MS * 1e-3 / max(est, 1e-9)) / n_steps) + 1):  # settle (run_workload.settle)
            run_block()
        ts = []
        for _ in range(R):
            fence(trajs)
            t0 = time.perf_counter()
            run_block()
            fence(trajs)
            dt = time.perf_counter() - t0
            ts.append(dt if emulated else D.max_over_ranks(dt))
        med = statistics.median(ts)
        return {"value": STRONG_TOTAL_ENVS * n_steps / med, "us_per_step": med / n_steps * 1e6, "us_per_step_min": min(ts) / n_steps * 1e6,
                "us_per_step_max": max(ts) / n_steps * 1e6, "blocks": R, "steps_per_block": n_steps}

    # ---- eager
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=STRONG_TOTAL_ENVS, rank=r, world_size=w,
                                    device_id=D.local_rank, redraw=args.redraw, seed=1234, obs_buffers=2)
    n = env.num_envs
    out["envs_per_gpu"] = n
    g = torch.Generator(device=dev).manual_seed(7 + r)
    actions = [(torch.rand((n, A), generator=g, device=dev) * 2 - 1).float() for _ in range(8)]
    traj = TrajectoryBuffer(steps, n, A, device=dev)
    env.reset()
    gather = [False]

    def eager_block():
        for i in range(steps):
            if traj.full():
                if gather[0]:
                    traj.all_gather_async(defer=True)
                else:
                    traj.clear()
            a, rw, d = traj.next_slot()
            env.step(actions[i % 8], rewards_out=rw, dones_out=d, actions_out=a)
            if i == 2:
                traj.issue_deferred()

    for _ in range(max(1, warmup // max(steps, 1))):
        eager_block()
    out["eager"] = {"no_all_gather": blocks_of(eager_block, steps, (traj,))}
    kt = KernelTrain(env, actions)
    out["eager"]["kernel_us"] = statistics.median(kt.run(400) for _ in range(3)) * 1e3
    out["eager"]["launch"] = env.launch_info()
    if gathering:
        gather[0] = True
        traj.clear()
        eager_block()
        out["eager"]["with_all_gather"] = blocks_of(eager_block, steps, (traj,))
        gather[0] = False
        fence((traj,))
        out["eager"]["packed_bytes_per_rank_per_chunk"] = traj._nbytes
    del env, traj, kt
    # ---- hipGraph, K steps per replay (two graphs over two trajectory chunks: chunk i is gathered while graph 1 - i replays).
    # K = 8 is the leg VERDICT round 4 named; K = 32 shows what a longer replay buys once a collective per replay is in the loop
    # (its host start + latency are per replay, the steps per replay amortise them)
    for K in (8, 32):
        steps_g = (steps + 2 * K - 1) // (2 * K) * (2 * K)  # whole replays, both graphs equally often
        env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=STRONG_TOTAL_ENVS, rank=r, world_size=w,
                                        device_id=D.local_rank, redraw="device", seed=1234, obs_buffers=2)
        trajs = [TrajectoryBuffer(K, n, A, device=dev) for _ in range(2)]
        rolls = [GraphedRollout(env, lambda obs, k: actions[k % 8], K, trajectory=t) for t in trajs]
        pending = [None, None]
        gathered = [None, None]

        def graph_block():
            for j in range(steps_g // K):
                i = j & 1
                if pending[i] is not None:
                    pending[i].wait()  # (stream-level for RCCL) chunk i has left before graph i overwrites it
                    pending[i] = None
                rolls[i].run()
                if gather[0]:
                    if gathered[i] is None:
                        gathered[i] = torch.empty((D.dist.get_world_size(), trajs[i]._nbytes), dtype=torch.uint8, device=dev)
                    pending[i] = D.dist.all_gather_into_tensor(gathered[i].view(-1), trajs[i]._packed, async_op=True)

        def graph_fence_extra():
            for i in (0, 1):
                if pending[i] is not None:
                    pending[i].wait()
                    pending[i] = None

        class _Drain:  # fence() drains these like a TrajectoryBuffer
            drain = staticmethod(graph_fence_extra)

        key = f"graph_k{K}"
        out[key] = {"no_all_gather": blocks_of(graph_block, steps_g, (_Drain,))}
        if gathering:
            gather[0] = True
            out[key]["with_all_gather"] = blocks_of(graph_block, steps_g, (_Drain,))
            gather[0] = False
            fence((_Drain,))
            out[key]["packed_bytes_per_rank_per_chunk"] = trajs[0]._nbytes
        del rolls, trajs, env, gathered
    gc.collect()
    torch.cuda.empty_cache()
    return out


def device_guard_check(D: Dist):
    """N > 1 only: the paths a one-GPU box cannot reach.  (i) fe_env_device(env) == LOCAL_RANK on every rank; (ii) a step() issued
    while ANOTHER device is current (the reference's `device_id` argument works without torch.cuda.set_device, TSE:28, 42;
    csrc/fe_env.hip DeviceGuard) launches on the env's device, returns tensors there, leaves the caller's current device
    unchanged, and computes what a step with the env's device current computes.  Returns a pass / fail record (never raises)."""
    import finenvs_amd

    rec = {"local_rank": D.local_rank, "devices_visible": torch.cuda.device_count()}
    try:
        _, _, A, W = CONFIGS[1]
        prices, day_id, _ = make_series(A)
        mk = lambda: finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=1024, evaluate=True,  # noqa: E731
                                               device_id=D.local_rank)
        env, twin = mk(), mk()
        rec["env_device"] = int(env._lib.fe_env_device(env._handle))
        ok = rec["env_device"] == D.local_rank
        a = (torch.rand((1024, A), device=D.dev) * 2 - 1).float()
        o0, r0, d0, _ = twin.step(a)
        if rec["devices_visible"] > 1:
            other = (D.local_rank + 1) % rec["devices_visible"]
            with torch.cuda.device(other):
                before = torch.cuda.current_device()
                o1, r1, d1, _ = env.step(a)
                after = torch.cuda.current_device()
            rec.update(other_device=other, current_device_before=before, current_device_after=after,
                       outputs_on=str(o1.device), current_device_restored=torch.cuda.current_device() == D.local_rank)
            torch.cuda.synchronize(D.dev)
            same = bool(torch.equal(o0, o1) and torch.equal(r0, r1) and torch.equal(d0, d1))
            rec["same_result_as_with_own_device_current"] = same
            ok = ok and before == other and after == other and str(o1.device) == D.dev and same and rec["current_device_restored"]
        else:
            rec["other_device"] = None
            rec["note"] = "one device visible: the cross-device half needs the multi-GPU node"
        rec["pass"] = bool(ok)
    except Exception as exc:  # noqa: BLE001
        rec["pass"] = False
        rec["error"] = f"{type(exc).__name__}: {exc}"
    return rec


def fused_rollout_legs(args):
    """SURVEY 8f.2 legs, reported beside the headline (never part of `value`): K env steps per launch with the policy
    evaluated in the kernel, at the headline's 65 536 envs x 1 asset.  The observation is never written to HBM, so these
    are not HBM-roofline numbers: the MLP / LSTM legs report the f32 MFMA rate of their contractions instead."""
    import finenvs_amd
    from finenvs_amd.rollout import FusedLinearRollout, FusedLSTMRollout, FusedMLPRollout

    _, N, A, _ = CONFIGS[2]
    prices, day_id, _ = make_series(A)
    g = torch.Generator().manual_seed(0)
    legs = []
    for form, W, K in (("linear_table", 64, 32), ("mlp_h64", 64, 32), ("lstm_h128", 4, 8), ("lstm_h1024", 4, 2)):
        try:
            env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device",  # (in-kernel redraws)
                                            seed=1234, obs_buffers=1)
            flop = 0.0
            if form == "linear_table":
                roll = FusedLinearRollout(env, torch.randn((W, 5), dtype=torch.float64, generator=g) * 2, 0.0, form="table")
                policy = "clamp(<window, weights (W, 5)>), log-return part precomputed as an indicator table"
            elif form == "mlp_h64":
                H = 64
                roll = FusedMLPRollout(env, torch.randn((5 * W, H), generator=g) * (8.0 / W ** 0.5), torch.randn(H, generator=g) * 0.3,
                                       torch.randn(H, generator=g) / H ** 0.5, 0.0)
                flop = 2.0 * N * A * (4 * W) * H
                policy = "Linear(5W, 64) -> ELU -> Linear(64, 1), first layer on v_mfma_f32_32x32x2_f32"
            else:
                H = int(form.split("_h")[1])
                torch.manual_seed(0)
                lstm, lin = torch.nn.LSTM(5, H, batch_first=True), torch.nn.Linear(H, 1)
                with torch.no_grad():
                    lstm.weight_ih_l0[:, :4].mul_(6.0 * H ** 0.5)
                roll = FusedLSTMRollout.from_modules(env, lstm, lin)
                flop = 2.0 * N * A * 4 * H * (8 * W + H * (W - 1))
                policy = (f"the reference's actor: LSTM(5, {H}) over the W rows -> Linear({H}, 1) -> tanh, gates on v_mfma_f32_32x32x2_f32"
                          + (", recurrent weights streamed from L2 (the reference example's hidden_dim)" if H > 128 else ", recurrent weights in registers"))
            roll.run(K, record_actions=True)
            times = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                roll.run(K, record_actions=True)
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) / K)
            ms = statistics.median(times)
            leg = {"form": form, "policy": policy, "envs": N, "num_assets": A, "window": W, "steps_per_launch": K,
                   "us_per_step": round(ms * 1e3, 2), "value": round(N / ms * 1e3, 1), "unit": "env-steps/s"}
            if flop:  # MFMA-bound legs: the contraction's FLOPs as executed against the dense f32 MFMA peak (256 CUs x 256 FLOP/cycle x 2.4 GHz)
                leg["mfma_f32_tflops"] = round(flop / ms / 1e9, 1)
                leg["mfma_f32_peak_tflops"] = 157.3
                leg["roofline"] = {"bound": "mfma", "achieved": round(flop / ms / 1e9, 1), "peak": 157.3, "unit": "TFLOP/s",
                                   "frac": round(flop / ms / 1e9 / 157.3, 3), "traffic": None,
                                   "note": "launch time from HIP events around K-step launches (policy + accounting); the f32-input MFMA "
                                           "shares the vector ALUs with the activations (SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/r02g_lstm_summary.md)"}
            legs.append(leg)
            del env, roll
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001
            legs.append({"form": form, "error": f"{type(exc).__name__}: {exc}"})
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--repeats", type=int, default=0, help="timed K-step blocks (0 = auto: ~0.15 s of timed work, 5..40)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs legs")
    ap.add_argument("--redraw", default="torch", choices=["torch", "device"],
                    help="the evaluation env's day redraw: 'torch' = the reference's RNG stream and per-step host read (class default, "
                         "pinned by the reference's fixtures); 'device' = in-kernel Philox (timed as the `device_redraw` leg by default)")
    ap.add_argument("--obs-f32", action="store_true", help="f32 observations (NOT the reference dtype; extra mode)")
    ap.add_argument("--graph", action="store_true", help="replay the 8-action ring as one hipGraph per 8 steps")
    ap.add_argument("--no-audition", action="store_true", help="take the observation ring as allocated (no placement audition)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 child passes")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        pmc_child(args)
        return

    # fd 1 carries ONE JSON line.  Native libraries write to stdout too -- RCCL printf()s a five-line version banner at
    # communicator init -- so fd 1 points at stderr for the life of the process and the real stdout is kept for the result.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if args.graph and args.redraw != "device":
        print("bench.py --graph: hipGraph capture cannot contain the per-step host read of redraw='torch' -> redraw='device'", file=sys.stderr)
        args.redraw = "device"
    D = Dist(args)
    head = run_workload(args.config, args, D, args.steps, args.warmup, args.repeats,
                        with_cpu=(not args.no_cpu and not D.multi))
    if "error" in head:
        sys.exit(f"bench.py: headline workload failed: {head['error']}")
    def measure_traffic(res, config):
        """roofline.traffic measured live (the workload's env is gone by now: the children have the card to themselves)."""
        if D.multi or args.no_pmc or args.graph or "error" in res:
            return
        t, src = live_pmc_traffic(config, args.obs_f32, args.redraw, args.no_audition)
        if t is not None:
            res["roofline"]["traffic"], res["roofline"]["traffic_source"] = t, src
        else:
            res["roofline"]["traffic_source"] = f"{res['roofline']['traffic_source']} (live PMC pass unavailable: {src})"

    measure_traffic(head, args.config)
    extras = []
    if not args.no_extra and not args.graph and os.environ.get("FE_BENCH_NO_EXTRA") != "1":
        # N = 1: configs 3, 4 and the per-GPU shard of config 5 (524 288 envs: the N = 1 anchor of the weak-scaling curve
        # the 8-GPU run continues); N > 1: the config-5 shard
        wanted = ([3, 4, 5] if not D.multi else [5])
        for c in wanted:
            if c == args.config:
                continue
            k = min(args.steps, 20)
            try:
                extras.append(run_workload(c, args, D, k, min(args.warmup, 5), 3, with_cpu=False))
                measure_traffic(extras[-1], c)
            except Exception as exc:  # noqa: BLE001
                extras.append({"workload": CONFIGS[c][0], "config": c, "error": f"{type(exc).__name__}: {exc}"})
                break

    late = {}  # legs measured after the headline and extra_configs: build_line() reads what is there

    def build_line():
        strong, guard = late.get("strong"), late.get("guard")
        fused, refsem, two_streams, devred = (late.get(k) for k in ("fused", "refsem", "two_streams", "devred"))
        return {
            "metric": "env-steps/sec",
            "value": head["value"],
            "unit": "env-steps/s",
            "n_gpus": D.world,
            "steps": args.steps,
            "warmup": head["warmup"],
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.obs_f32 else "f64",
            "data": "synthetic",
            "config": {"workload": head["workload"], "envs_per_gpu": head["envs_per_gpu"],
                       "num_assets": head["num_assets"], "window": head["window"],
                       "obs_buffers": head["obs_buffers"], "obs_ring_audition": head["obs_ring_audition"],
                       "obs_ring_audition_bound": {"extra_candidates": AUDITION_EXTRA, "budget_bytes": AUDITION_BUDGET},
                       "eval_redraw": args.redraw, "redraw_contract": REDRAW_CONTRACT,
                       "launch_mode": head["launch_mode"], "launch": head["launch"],
                       "untimed_steps_before_timing": {"warmup": head["warmup"], "settle": head["settle_steps"],
                                                       "why": f"{SETTLE_MS:g} ms of steps straight before each timed phase: after an idle period the GPU "
                                                              "runs the same kernel 15 - 25 % slower for ~10 ms (clock transient, "
                                                              "profiles/r04_microbench/idle_transient.txt)"},
                       "timed_region": f"STEADY STATE: beyond --warmup {head['warmup']}, {head['settle_steps']} untimed steps ({SETTLE_MS:g} ms of GPU time) "
                                       "run straight before the timed phase, because after an idle period this pool's GPUs run the same kernel "
                                       "15 - 25 % slower for ~10 ms and a rollout runs for minutes (untimed_steps_before_timing).  Then: median of R blocks of "
                                       "exactly `steps` steps, each between (drain + synchronize + barrier + synchronize) fences, MAX over ranks "
                                       "per block; after each block one train of back-to-back C-ABI launches of the same kernel form gives "
                                       "`roofline.kernel_ms` (same ring, same clock regime).  A block of --steps 20 at 64k envs is ~0.6 ms: the "
                                       "idle-GPU start of every block, the closing fence and env.step's Python make `ms_per_step` a few per cent "
                                       "longer than `roofline.kernel_ms`"},
            # the un-auditioned regime beside the headline: same loop, same fences, the ring as the allocator handed it out
            "as_allocated": head["as_allocated"],
            "repeats": head["repeats"],
            "roofline": head["roofline"],
            "cpu_baseline": head.get("cpu_baseline"),
            "multi_gpu": (dict(head["multi_gpu"], strong=strong, device_guard=guard) if head.get("multi_gpu") else None),
            "strong_scaling": strong,
            "device_redraw": devred,
            "extra_configs": [{k: v for k, v in e.items() if k != "cpu_baseline"} for e in extras],
            "fused_rollouts": fused,
            "reference_semantics": refsem,
            "two_streams": two_streams,
        }

    # the strong-scaling reading of the metric (64k envs IN TOTAL over the world) beside the weak one; at N = 1 also the
    # per-GPU shard of a 2 / 4 / 8-GPU world emulated on this GPU (no collectives): the measured basis of DESIGN.md section 7
    strong = guard = None
    want_strong = not args.no_extra and not args.graph and not args.obs_f32 and args.config == 2
    watchdog = None
    if D.multi:
        # The strong-scaling leg and the DeviceGuard check have never run with more than one RCCL rank before the driver's
        # scaling bench.  They must not be able to cost the run its line: if they are not through after WATCHDOG_S, EVERY
        # rank gives up at about the same time -- rank 0 prints the line it has (headline + extra_configs, the unfinished
        # legs marked) and all ranks leave with exit code 0 without waiting for a collective that will not come.
        import threading

        def give_up():
            late.setdefault("strong", {"error": f"not finished after {WATCHDOG_S:g} s (watchdog): leg abandoned"})
            late.setdefault("guard", {"pass": False, "error": f"not finished after {WATCHDOG_S:g} s (watchdog)"})
            if D.rank == 0:
                sys.stderr.flush()
                print(json.dumps(build_line()), file=result_out, flush=True)
            os._exit(0)

        watchdog = threading.Timer(WATCHDOG_S, give_up)
        watchdog.daemon = True
    if want_strong:
        if watchdog is not None:
            watchdog.start()
        try:
            if os.environ.get("FE_BENCH_HANG_STRONG") == "1":  # rehearsal of the watchdog (tests only)
                time.sleep(1e6)
            strong = strong_scaling_leg(args, D, args.steps, args.warmup)
            if not D.multi:
                strong["shard_preview"] = [strong_scaling_leg(args, D, args.steps, args.warmup, world=w_, rank=w_ - 1) for w_ in (2, 4, 8)]
        except Exception as exc:  # noqa: BLE001  (the same code runs on every rank: an exception here is raised on all of them)
            strong = {"error": f"{type(exc).__name__}: {exc}"}
        late["strong"] = strong
    if D.multi:
        if watchdog is not None and not watchdog.is_alive():
            watchdog.start()
        try:
            mine = device_guard_check(D)
            recs = [None] * D.dist.get_world_size()
            D.dist.all_gather_object(recs, mine)
            guard = {"pass": all(bool(r_ and r_.get("pass")) for r_ in recs), "ranks": recs,
                     "what": "fe_env_device(env) == LOCAL_RANK on every rank; a step() issued with ANOTHER device current launches on the "
                             "env's device, leaves the caller's current device unchanged and equals the step of a twin env (TSE:28, 42; DeviceGuard)"}
        except Exception as exc:  # noqa: BLE001
            guard = {"pass": False, "error": f"{type(exc).__name__}: {exc}"}
        late["guard"] = guard
        if watchdog is not None:
            watchdog.cancel()

    fused = refsem = two_streams = devred = None
    if not D.multi and not args.no_extra and not args.graph and not args.obs_f32 and os.environ.get("FE_BENCH_NO_EXTRA") != "1":
        if args.redraw == "torch" and args.config == 2:
            # the build's own redraw contract on the same workload, the same ring policy, the same block protocol
            try:
                dr = run_workload(2, args, D, args.steps, args.warmup, args.repeats, with_cpu=False, redraw="device")
                devred = {k: dr[k] for k in ("workload", "value", "ms_per_step", "steps", "warmup", "settle_steps", "repeats", "as_allocated", "error") if k in dr}
                if "roofline" in dr:
                    devred.update(kernel=dr["roofline"]["kernel"], kernel_ms=dr["roofline"]["kernel_ms"], frac=dr["roofline"]["frac"])
                devred["what"] = "the headline workload with eval_redraw='device' (in-kernel Philox redraws, no host flag), same ring, fences and block protocol as `value`"
            except Exception as exc:  # noqa: BLE001
                devred = {"error": f"{type(exc).__name__}: {exc}"}
        fused = fused_rollout_legs(args)
        try:
            refsem = reference_semantics_leg(args, args.steps, args.repeats)
        except Exception as exc:  # noqa: BLE001
            refsem = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            two_streams = two_stream_leg(args, args.steps)
        except Exception as exc:  # noqa: BLE001
            two_streams = {"error": f"{type(exc).__name__}: {exc}"}
    late.update(fused=fused, refsem=refsem, two_streams=two_streams, devred=devred)

    out = build_line() if D.rank == 0 else None
    if D.dist is not None:
        D.barrier()
        D.dist.destroy_process_group()
    if out is not None:  # the very last thing this process writes: nothing (process-group teardown chatter) follows it
        sys.stderr.flush()
        print(json.dumps(out), file=result_out, flush=True)


if __name__ == "__main__":
    main()
