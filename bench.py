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


_LIVE_PMC_BROKEN = []  # "no profiler can run in this process at all" (already profiled / no rocprofv3): never asked again
PMC_CHILD_TIMEOUT_S = 90.0  # one counter-pass child (import + tables + a 154 GB first touch + 16 launches under the profiler)
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


_CHILD = [None]  # the counter-pass child in flight (a Popen in its own process group), if any


def kill_child() -> None:
    """End the counter-pass child in flight -- the process group this script started, by its id -- and reap it."""
    import signal

    proc = _CHILD[0]
    if proc is None or proc.poll() is not None:
        return
    try:
        os.killpg(proc.pid, signal.SIGKILL)  # (start_new_session=True: the group's id is the child's pid)
    except (ProcessLookupError, PermissionError):
        pass
    try:
        proc.wait(timeout=10)
    except Exception:  # noqa: BLE001
        pass


def _last_words(path: str, limit: int = 110) -> str:
    """The last informative line of a child's stderr (what made it exit), shortened for the bench line."""
    try:
        with open(path, errors="replace") as f:
            lines = [ln.strip() for ln in f if ln.strip()]
    except OSError:
        return ""
    for ln in reversed(lines):
        if any(w in ln for w in ("Error", "error", "Exception", "failed", "Killed", "fault", "abort", "terminate")):
            return ln[:limit]  # (a Python traceback's last line starts with the exception's name and reason)
    return lines[-1][:limit] if lines else ""


def live_pmc_traffic(config: int, f32: bool, redraw: str, no_audition: bool = False):
    """HBM traffic per launch MEASURED IN THIS RUN: two child processes of this very script (`--pmc-child`: the same
    workload, 16 launches) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, as
    MI355X_MICROARCH.md's HBM section prescribes; both counters are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B
    requests at 64 B).  Returns (bytes per launch, description) or (None, cause).  A failure of one config's children is
    that config's alone (its cause = the child's own last stderr line); only "no profiler can run here at all" is remembered."""
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
            # the interpreter itself directly after `--` (no env / shell hop: the profiler's preloaded library has the GPU
            # initialised before the program starts)
            cmd = [exe, "--pmc", kind, "--output-format", "csv", "-d", os.path.join(out, kind), "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--config", str(config), "--redraw", redraw]
            if f32:
                cmd.append("--obs-f32")
            if no_audition:
                cmd.append("--no-audition")
            errp = os.path.join(out, kind + ".err")
            with open(errp, "w") as ef:
                # its own process group, remembered in _CHILD: a timeout here or the line's watchdog ends exactly this group
                proc = subprocess.Popen(cmd, cwd="/tmp", env=pmc_child_env(), stdout=subprocess.DEVNULL, stderr=ef, start_new_session=True)
                _CHILD[0] = proc
                try:
                    rc = proc.wait(timeout=PMC_CHILD_TIMEOUT_S)
                except subprocess.TimeoutExpired:
                    kill_child()
                    return None, f"--pmc {kind} child of config {config} not done after {PMC_CHILD_TIMEOUT_S:g} s: {_last_words(errp, 60)}"
                finally:
                    _CHILD[0] = None
            if rc != 0:
                return None, f"--pmc {kind} child of config {config} exited {rc}: {_last_words(errp)}"
            rows = []
            for f in glob.glob(os.path.join(out, kind, "*", "*_counter_collection.csv")):
                for row in csv.DictReader(open(f)):
                    if is_step_kernel(row["Kernel_Name"]) and row["Counter_Name"] == kind:
                        rows.append(float(row["Counter_Value"]))
            if len(rows) < 8:
                return None, f"{len(rows)} {kind} rows for the step kernel of config {config} (16 expected)"
            rows = rows[len(rows) // 4:]  # drop the warm-up launches
            vals[kind] = (sum(rows) / len(rows), len(rows))
    except Exception as exc:  # noqa: BLE001
        return None, f"{type(exc).__name__}: {exc}"[:140]
    finally:
        shutil.rmtree(out, ignore_errors=True)
    traffic = (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
    return traffic, (f"measured in this run: --pmc FETCH_SIZE / WRITE_SIZE children, {vals['FETCH_SIZE'][1]}/{vals['WRITE_SIZE'][1]} launches")


def wait_for_free_memory(dev, need: int, limit_s: float = 60.0) -> int:
    """A previous owner's buffers (this process's last workload, or another process that has just exited) are given back
    asynchronously: a 150 GB allocation can take seconds to reappear as free memory.  Returns the free bytes seen last."""
    t0 = time.perf_counter()
    free = torch.cuda.mem_get_info(dev)[0]
    while free < need and time.perf_counter() - t0 < limit_s:
        gc.collect()
        torch.cuda.empty_cache()
        time.sleep(0.5)
        free = torch.cuda.mem_get_info(dev)[0]
    return free


def pmc_child(args):
    """`--pmc-child`: the workload's step kernel in the timed loop's form, 16 launches, nothing else (what the rocprofv3 counter
    passes of live_pmc_traffic profile).  It starts while the parent's last workload may still be on its way back to the free
    pool (round 5: the config-4 child died allocating its 154 GB observation), so it waits for the memory like run_workload."""
    import finenvs_amd

    name, N, A, W = CONFIGS[args.config]
    prices, day_id, _ = make_series(A)
    obs_bytes = N * W * 5 * A * (4 if args.obs_f32 else 8)
    obs_buffers = 2 if 2 * obs_bytes < 200e9 else 1
    free = wait_for_free_memory("cuda:0", obs_buffers * obs_bytes + (6 << 30), 45.0)
    print(f"[pmc-child] config {args.config}: {free / 2**30:.1f} GiB free, observation ring {obs_buffers} x {obs_bytes / 2**30:.1f} GiB", file=sys.stderr, flush=True)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw=args.redraw, seed=1234,
                                    obs_buffers=obs_buffers, obs_dtype=torch.float32 if args.obs_f32 else torch.float64)
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
    print("[pmc-child] done", file=sys.stderr, flush=True)


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
            # ProcessGroupNCCL's heartbeat-monitor thread polls the TCPStore; at teardown the store can go first and the thread's
            # exception then std::terminate()s the process (seen once in ~10 rehearsals: exit code -6 after all the work was done).
            # The monitor only serves torch's own hang diagnostics; this script has its watchdog.
            os.environ.setdefault("TORCH_NCCL_ENABLE_MONITORING", "0")
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
        wait_for_free_memory(D.dev, obs_buffers * obs_bytes + (6 << 30))
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
        issue = lambda n: [roll.run() for _ in range(n // 8)]  # noqa: E731
    else:
        env.reset()
        issue = lambda n: [one_step(i) for i in range(n)]  # noqa: E731
    launched = [0]  # step-kernel launches so far (loop + trains): `untimed_steps_before_value` is read off it

    def run_steps(n):
        launched[0] += n
        issue(n)

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
                launched[0] += train[1]
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
    before_value = None  # launches before the first block that counts towards `value`
    reuse_blocks = None  # the as-allocated blocks ARE the headline's when the audition kept the ring (on every rank)
    if not args.no_audition and roll is None:
        if D.multi:
            gather[0] = True
            fence()
            traj.clear()
            run_steps(T)  # one full chunk: the first timed block has something to gather
        before_value = launched[0]
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
                        "ring_changed_by_audition": changed}
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
        if reuse_blocks is None:
            before_value = launched[0]
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
        if reuse_blocks is None:
            before_value = launched[0]
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

    # The practical store ceiling beside the roofline: torch's fill kernel (one aligned 4 KiB unit per workgroup, one store per
    # lane: the fastest way of writing HBM found on this part) over the SAME observation ring, in the same
    # train protocol as `kernel_ms` (back-to-back launches, alternating ring members, one HIP-event pair per train).  Pure stores:
    # it reads nothing and computes nothing, so it bounds what any kernel that has to produce these bytes can reach here.
    fill_ms = None
    try:
        ring = list(env._obs_ring)
        if ring:
            kf = max(4, min(k2, 60))
            fills = []
            for rep in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(kf):
                    ring[i % len(ring)].fill_(0.0)
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    fills.append(e0.elapsed_time(e1) / kf)
            fill_ms = statistics.median(fills)
    except Exception as exc:  # noqa: BLE001
        print(f"bench.py: store-ceiling fill skipped ({type(exc).__name__}: {exc})", file=sys.stderr)

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
        # untimed steps: `settle_steps` run straight before every timed phase (SETTLE_MS of GPU time: the post-idle clock transient,
        # DESIGN.md section 8); `untimed_steps_before_value` = every step-kernel launch (loop + trains) before the first block of `value`
        "settle_steps": settle_steps, "untimed_steps_before_value": before_value,
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
            "kernel_ms_runs": kern if len(kern) <= 8 else [min(kern), statistics.median(kern), max(kern)],
            "kernel_launches_per_run": k2,
            # for tools/summarize_prof.py: how the LAST launches of this instantiation in the run are laid out
            "kernel_train_layout": {"repeats": len(kern), "loop_launches_per_block": steps, "train_launches": k2,
                                    "order": "block, train, block, train, ..."} if roll is None else None,
            # `achieved` counts only bytes that must cross HBM (observation write + state + outputs):
            "hbm_bytes_per_env_step": Bh, "units_per_launch": N,
            "bytes_model": (f"B_hbm = {5 * obs_elem}WA+84A+36 (SURVEY 8d's 72WA+84A+36 minus the 32WA cache-served window reads"
                            + ("" if obs_elem == 8 else " and 20WA of f32 observations") + ")"),
            # what `frac` WOULD read on SURVEY 8(d)'s B (window re-reads counted as if they crossed HBM): above 1 at config 2
            "frac_on_survey_8d_bytes": Bs * N / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            # the window re-read of the SURVEY 8(d) formula comes from L2 / Infinity Cache, reported apart:
            "l2_read_bytes_per_env_step": l2_read, "l2_GBps": l2_read * N / (kern_ms * 1e-3) / 1e9,
            "survey_8d_bytes_per_env_step": Bs,
            # whole-job check: HBM bytes / wall ms_per_step (must stay below the peak as well)
            "achieved_wall": Bh * N / (block / steps) / 1e9,
            # torch's fill over the same ring, same train protocol: observation bytes / launch interval, and this kernel's
            # observation bytes alone over `kernel_ms` for the like-for-like comparison
            "fill_ms": fill_ms, "fill_GBps": (obs_bytes / (fill_ms * 1e-3) / 1e9) if fill_ms else None,
            "obs_write_GBps": obs_bytes / (kern_ms * 1e-3) / 1e9,
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






# ------------------------------------------------------------------------------------------------------------------------
# The contract line.  bench.py's LAST stdout line is one strict JSON object of at most LINE_CAP characters (the driver keeps
# an 8 KB tail of stdout; round 5's 20 KB line was not parsed).  Numbers only: what each number means is DESIGN.md section 8.
# Everything else a run measured goes to stderr as `[bench-detail] {...}` lines (and to --detail PATH as one JSON document).
# ------------------------------------------------------------------------------------------------------------------------
LINE_CAP = 7600


def _sig(x, digits: int = 6):
    """Floats at `digits` significant digits (the line is for reading numbers, the detail record keeps them whole)."""
    if isinstance(x, float) and x == x and x not in (float("inf"), float("-inf")) and x != 0.0:
        return float(f"{x:.{digits}g}")
    return x


def strict(obj, bad=None, path=""):
    """A copy of `obj` that json.dumps(..., allow_nan=False) accepts: non-finite floats become None and their paths are
    collected in `bad` (one NaN in one leg must not cost the run its line)."""
    if isinstance(obj, float):
        if obj != obj or obj in (float("inf"), float("-inf")):
            if bad is not None:
                bad.append(path)
            return None
        return obj
    if isinstance(obj, dict):
        return {str(k): strict(v, bad, f"{path}.{k}" if path else str(k)) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [strict(v, bad, f"{path}[{i}]") for i, v in enumerate(obj)]
    if isinstance(obj, (str, int, bool)) or obj is None:
        return obj
    if hasattr(obj, "item"):  # numpy / torch scalars
        return strict(obj.item(), bad, path)
    return str(obj)


def _short(s, n):
    s = "" if s is None else str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def compact_roofline(r: dict, full: bool) -> dict:
    out = {"bound": r["bound"], "achieved": _sig(r["achieved"]), "peak": r["peak"], "unit": r["unit"], "frac": _sig(r["frac"], 4),
           "traffic": _sig(r.get("traffic")), "traffic_source": _short(r.get("traffic_source"), 80),
           "kernel": r.get("kernel"), "kernel_ms": _sig(r.get("kernel_ms")),
           "hbm_bytes_per_env_step": r.get("hbm_bytes_per_env_step"), "units_per_launch": r.get("units_per_launch")}
    if r.get("traffic_live_error"):
        out["traffic_live_error"] = _short(r["traffic_live_error"], 140)
    if r.get("traffic") and r.get("hbm_bytes_per_env_step"):
        out["traffic_over_algorithmic"] = _sig(r["traffic"] / (r["hbm_bytes_per_env_step"] * r["units_per_launch"]), 4)
    if r.get("fill_GBps"):  # the practical store ceiling on the same ring (torch's fill, same train protocol) and this kernel's share of it
        out["fill_GBps"] = _sig(r["fill_GBps"], 5)
        out["obs_write_over_fill"] = _sig(r["obs_write_GBps"] / r["fill_GBps"], 4)
    if full:
        out.update(bytes_model=r.get("bytes_model"), survey_8d_bytes_per_env_step=r.get("survey_8d_bytes_per_env_step"),
                   frac_on_survey_8d_bytes=_sig(r.get("frac_on_survey_8d_bytes"), 4), achieved_wall=_sig(r.get("achieved_wall")),
                   kernel_launches_per_train=r.get("kernel_launches_per_run"),
                   trains=(r.get("kernel_train_layout") or {}).get("repeats"))
    return out


def compact_multi_gpu(m: dict, strong, guard) -> dict:
    """Numbers only (task 8): what the first SCALE record needs to explain itself."""
    out = {k: _sig(m.get(k)) for k in ("ranks_seen", "collective_backend", "kernel_form_by_rank", "trajectory_slots",
                                       "packed_bytes_per_rank_per_chunk", "gather_only_ms", "gather_only_inbound_GBps_per_gpu",
                                       "exposed_ms_per_step")}
    out["with_all_gather"] = _sig(m.get("value_with_all_gather"))
    out["no_all_gather"] = _sig(m.get("value_no_all_gather"))
    p = m.get("prediction") or {}
    out["predicted_gather_only_ms"] = [_sig(x, 4) for x in p.get("predicted_gather_only_ms", [])]
    rc = m.get("rccl") or {}
    out["rccl"] = _short("; ".join((rc.get("version") or [])[:1] + (rc.get("algorithm_protocol") or [])[:1]), 120) if rc else None
    out["device_guard"] = None if guard is None else {"pass": bool(guard.get("pass")), **({"error": _short(guard["error"], 100)} if guard.get("error") else {})}
    if strong is not None:
        out["strong"] = compact_strong(strong)
    return out


def compact_strong(st: dict) -> dict:
    """The strong-scaling reading (65 536 envs IN TOTAL): env-steps/s per launch mode; at N = 1 the per-GPU step of a 2 / 4 / 8-GPU
    world emulated on this GPU (us per step, one column per world)."""
    if "error" in st:
        return {"error": _short(st["error"], 120)}
    out = {"total_envs": st.get("total_envs"), "world": st.get("world"), "envs_per_gpu": st.get("envs_per_gpu")}
    for mode in ("eager", "graph_k8", "graph_k32"):
        m = st.get(mode)
        if not m:
            continue
        out[mode] = {k: _sig(m[k]["value"]) for k in ("no_all_gather", "with_all_gather") if k in m}
    prev = st.get("shard_preview")
    if prev:
        worlds = [st] + list(prev)
        out["us_per_step_at_world"] = {"worlds": [w.get("world") for w in worlds], "emulated_on_one_gpu": True,
                                       **{mode: [_sig(w[mode]["no_all_gather"]["us_per_step"], 4) if mode in w else None for w in worlds]
                                          for mode in ("eager", "graph_k8", "graph_k32")},
                                       "kernel_us": [_sig(w["eager"].get("kernel_us"), 4) if "eager" in w else None for w in worlds]}
    return out


def compact_line(detail: dict) -> dict:
    """The driver's line from the full record: the contract keys, `roofline`, `cpu_baseline`, `multi_gpu` (N > 1), and one short
    summary per extra workload / leg.  No prose (DESIGN.md section 8 says what every key means)."""
    head = detail["headline"]
    hr = head["roofline"]
    unfinished = list(detail.get("unfinished") or [])
    line = {
        "metric": "env-steps/sec", "value": head["value"], "unit": "env-steps/s", "n_gpus": detail["n_gpus"],
        "steps": detail["steps"], "warmup": head["warmup"], "ms_per_step": head["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": detail["dtype"], "data": "synthetic",
        "config": {"workload": head["workload"], "envs_per_gpu": head["envs_per_gpu"], "num_assets": head["num_assets"],
                   "window": head["window"], "obs_buffers": head["obs_buffers"], "obs_dtype": detail["dtype"],
                   "eval_redraw": detail["eval_redraw"], "launch_mode": head["launch_mode"],
                   "settle_steps": head["settle_steps"], "untimed_steps_before_value": head.get("untimed_steps_before_value"),
                   "timed_blocks": next(iter(head["repeats"].values()))["blocks"] if head.get("repeats") else None,
                   "ms_per_step_min_max": [_sig(next(iter(head["repeats"].values()))[k]) for k in ("ms_per_step_min", "ms_per_step_max")]
                   if head.get("repeats") else None},
        "roofline": compact_roofline(hr, full=True),
        "cpu_baseline": None,
    }
    cb = head.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": _short(cb["sample"], 120), "value_1thread": _sig(cb.get("value_1thread")),
                                "cpu_model": _short(cb.get("cpu_model"), 60),
                                "reference_quoted": {k: (cb.get("reference_quoted") or {}).get(k) for k in ("value", "cores")}}
    aa = head.get("as_allocated")
    if aa:
        line["as_allocated"] = {"value": _sig(aa["value"]), "ms_per_step": _sig(aa["ms_per_step"]), "kernel_ms": _sig(aa["kernel_ms"]),
                                "frac": _sig(aa["frac_of_8TBps_kernel"], 4), "ring_changed_by_audition": aa.get("ring_changed_by_audition")}
    if head.get("multi_gpu"):
        line["multi_gpu"] = compact_multi_gpu(head["multi_gpu"], detail.get("strong_scaling"), detail.get("device_guard"))
    extras = []
    for e in detail.get("extra_configs") or []:
        if "error" in e:
            extras.append({"config": e.get("config"), "workload": e.get("workload"), "error": _short(e["error"], 120)})
            continue
        x = {"config": e["config"], "workload": e["workload"], "value": _sig(e["value"]), "ms_per_step": _sig(e["ms_per_step"]),
             "envs_per_gpu": e["envs_per_gpu"], "obs_buffers": e["obs_buffers"], "roofline": compact_roofline(e["roofline"], full=False)}
        if e.get("as_allocated"):
            x["as_allocated"] = {"value": _sig(e["as_allocated"]["value"]), "frac": _sig(e["as_allocated"]["frac_of_8TBps_kernel"], 4)}
        if e.get("multi_gpu"):
            x["multi_gpu"] = compact_multi_gpu(e["multi_gpu"], None, None)
        extras.append(x)
    line["extra_configs"] = extras
    legs = []
    for name, leg in (detail.get("legs") or {}).items():
        if leg is None:
            continue
        if "error" in leg:
            legs.append({"leg": name, "error": _short(leg["error"], 100)})
            continue
        x = {"leg": name, "value": _sig(leg.get("value")), "ms_per_step": _sig(leg.get("ms_per_step"))}
        if leg.get("frac") is not None:
            x["frac"] = _sig(leg["frac"], 4)
        if leg.get("bound"):
            x["bound"] = leg["bound"]
        legs.append(x)
    line["legs"] = legs
    if detail.get("strong_scaling") is not None and not head.get("multi_gpu"):
        line["strong_scaling"] = compact_strong(detail["strong_scaling"])
    if unfinished:
        line["unfinished"] = unfinished
    return line


def render_line(detail: dict) -> str:
    """compact_line -> one strict JSON line of at most LINE_CAP characters.  Optional sections are dropped (named in `dropped`)
    rather than ever exceeding the cap; the contract keys, `roofline` and `cpu_baseline` always stay."""
    bad = []
    line = strict(compact_line(detail), bad)
    if bad:
        line["nonfinite_set_to_null"] = bad[:12]
    dropped = []
    for victim in (None, "strong_scaling", "legs", "as_allocated", "extra_configs", "multi_gpu"):
        if victim is not None:
            if victim not in line:
                continue
            if victim == "multi_gpu":  # keep its numbers, lose the nested legs
                line["multi_gpu"] = {k: v for k, v in line["multi_gpu"].items() if not isinstance(v, (dict, list))}
            else:
                del line[victim]
            dropped.append(victim)
            line["dropped_for_size"] = dropped
        s = json.dumps(line, allow_nan=False, separators=(", ", ": "))
        if len(s) <= LINE_CAP:
            return s
    return s[:LINE_CAP]  # unreachable with the sections above gone (the core is ~2.5 KB)


_EMIT_DETAIL = [True]  # main() switches it off on ranks other than 0 (one copy of every record, not one per rank)


def emit_detail(tag: str, obj) -> None:
    """Full records go to stderr, one JSON line each (the driver keeps stderr beside stdout); never to stdout."""
    if not _EMIT_DETAIL[0]:
        return
    try:
        print(f"[bench-detail] {json.dumps({tag: strict(obj)}, allow_nan=False)}", file=sys.stderr, flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"[bench-detail] {tag}: not serialisable ({type(exc).__name__}: {exc})", file=sys.stderr, flush=True)


def leg_summary(name: str, leg):
    """value / ms_per_step / frac of one extra leg's detail record (the keys compact_line reads)."""
    if leg is None:
        return None
    if "error" in leg:
        return {"error": leg["error"]}
    if name == "device_redraw":
        return {"value": leg["value"], "ms_per_step": leg["ms_per_step"], "frac": leg.get("frac"), "bound": "hbm"}
    if name == "reference_semantics":
        return {"value": leg["value"], "ms_per_step": leg["ms_per_step"],
                "frac": hbm_bytes(64, 1, 8) * 65536 / (leg["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, "bound": "hbm"}
    if name == "two_streams":
        return {"value": leg["value"], "ms_per_step": leg["ms_per_step_all_envs"], "frac": leg["frac_of_8TBps"], "bound": "hbm"}
    # fused rollouts: MFMA-bound where there is a contraction, otherwise no roofline figure
    r = leg.get("roofline")
    return {"value": leg["value"], "ms_per_step": leg["us_per_step"] * 1e-3, "frac": r["frac"] if r else None, "bound": "mfma" if r else None}


class LineGuard:
    """Owns the one stdout line.  print_once() writes it exactly once; arm(seconds) starts a watchdog that -- if the legs after
    the headline are not through in time (a hung collective, a hung kernel) -- marks what is unfinished, prints the line this rank
    has, and ends the process with exit code 3 WITHOUT waiting for the GPU (no restart, no re-exec).  Ranks other than 0 leave a few
    seconds later, so that the launcher does not tear rank 0 down before its line is out."""

    EXIT_CODE = 3

    def __init__(self, result_out, detail: dict, rank: int, detail_path=None):
        import threading

        self.out, self.detail, self.rank, self.detail_path = result_out, detail, rank, detail_path
        self.lock = threading.Lock()
        self.printed = False
        self.timer = None
        self.current = ["(nothing)"]  # the leg in progress, for the watchdog's record

    def print_once(self):
        with self.lock:
            if self.printed:
                return
            s = None
            for _ in range(3):  # (the watchdog may render while the main thread is still adding a leg's record)
                try:
                    s = render_line(self.detail)
                    break
                except Exception as exc:  # noqa: BLE001
                    print(f"bench.py: rendering the line failed ({type(exc).__name__}: {exc}); retrying", file=sys.stderr, flush=True)
                    time.sleep(0.05)
            if s is None:  # the headline alone
                core = {k: self.detail[k] for k in ("n_gpus", "steps", "dtype", "eval_redraw", "headline")}
                s = render_line(dict(core, unfinished=["(record could not be rendered in full)"]))
            self.printed = True
            if self.rank == 0:
                sys.stderr.flush()
                if self.detail_path:
                    try:
                        with open(self.detail_path, "w") as f:
                            json.dump(strict(self.detail), f, allow_nan=False)
                    except Exception as exc:  # noqa: BLE001
                        print(f"bench.py: --detail {self.detail_path}: {type(exc).__name__}: {exc}", file=sys.stderr)
                print(s, file=self.out, flush=True)

    def arm(self, seconds: float):
        import threading

        def give_up():
            msg = f"watchdog: `{self.current[0]}` not finished after {seconds:g} s; line printed without it, exit code {self.EXIT_CODE}"
            print(f"bench.py rank {self.rank}: {msg}", file=sys.stderr, flush=True)
            self.detail.setdefault("unfinished", []).append(self.current[0])
            self.detail["watchdog"] = msg
            self.print_once()
            kill_child()  # a rocprofv3 counter pass in flight must not outlive this process on the GPU
            if self.rank != 0:
                time.sleep(4.0)
            os._exit(self.EXIT_CODE)

        self.timer = threading.Timer(seconds, give_up)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--repeats", type=int, default=0, help="timed K-step blocks (0 = auto: ~0.15 s of timed work, 5..40)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs and the extra legs")
    ap.add_argument("--redraw", default="torch", choices=["torch", "device"],
                    help="the evaluation env's day redraw: 'torch' = the reference's RNG stream and per-step host read (class default, "
                         "pinned by the reference's fixtures); 'device' = in-kernel Philox (timed as the `device_redraw` leg by default)")
    ap.add_argument("--obs-f32", action="store_true", help="f32 observations (NOT the reference dtype; extra mode)")
    ap.add_argument("--graph", action="store_true", help="replay the 8-action ring as one hipGraph per 8 steps")
    ap.add_argument("--no-audition", action="store_true", help="take the observation ring as allocated (no placement audition)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 child passes")
    ap.add_argument("--detail", default=None, metavar="PATH", help="also write the full record (everything stderr's [bench-detail] lines carry) as one JSON document")
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
    sys.modules.setdefault("bench", sys.modules[__name__])  # tools/bench_legs.py does `from bench import ...`: this module, not a second copy

    if args.graph and args.redraw != "device":
        print("bench.py --graph: hipGraph capture cannot contain the per-step host read of redraw='torch' -> redraw='device'", file=sys.stderr)
        args.redraw = "device"
    D = Dist(args)
    _EMIT_DETAIL[0] = D.rank == 0
    t_start = time.perf_counter()

    def note(msg):
        if D.rank == 0:
            print(f"[bench {time.perf_counter() - t_start:6.1f} s] {msg}", file=sys.stderr, flush=True)

    # ---- 1. the headline (+ cpu_baseline on the plain N = 1 run): nothing else has touched the GPU yet
    head = run_workload(args.config, args, D, args.steps, args.warmup, args.repeats, with_cpu=(not args.no_cpu and not D.multi))
    if "error" in head:
        sys.exit(f"bench.py: headline workload failed: {head['error']}")
    detail = {"n_gpus": D.world, "steps": args.steps, "dtype": "f32" if args.obs_f32 else "f64", "eval_redraw": args.redraw,
              "headline": head, "extra_configs": [], "legs": {}, "strong_scaling": None, "device_guard": None, "unfinished": []}
    guard = LineGuard(result_out, detail, D.rank, args.detail)
    emit_detail("headline", head)
    note(f"headline: {head['value']:.4g} env-steps/s, kernel {head['roofline']['kernel_ms'] * 1e3:.2f} us, frac {head['roofline']['frac']:.3f}")

    # ---- 2. from here on a watchdog owns the line: whatever hangs below costs its own leg, not `value` / roofline / cpu_baseline
    guard.arm(WATCHDOG_S)

    def leg(name):
        guard.current[0] = name
        note(f"{name} ...")

    def measure_traffic(res, config):
        """roofline.traffic measured live (the workload's env is gone by now: the children have the card to themselves)."""
        if D.multi or args.no_pmc or args.graph or "error" in res:
            return
        leg(f"pmc_traffic_config{config}")
        # the workload's tensors died with run_workload's frame, AFTER its own empty_cache(): without this the caching allocator of
        # THIS process still holds them (config 4: 154 GB) and the child cannot allocate its own (round 5's "child exited with 1")
        gc.collect()
        torch.cuda.empty_cache()
        t, src = live_pmc_traffic(config, args.obs_f32, args.redraw, args.no_audition)
        if t is not None:
            res["roofline"]["traffic"], res["roofline"]["traffic_source"] = t, src
        else:  # the committed counter passes stay in `traffic` (named in traffic_source); the live attempt's cause beside them
            res["roofline"]["traffic_live_error"] = src
            note(f"live PMC pass of config {config} failed: {src}")

    measure_traffic(head, args.config)
    want_extra = not args.no_extra and not args.graph and os.environ.get("FE_BENCH_NO_EXTRA") != "1"
    if os.environ.get("FE_BENCH_HANG_LEG") == "1":  # rehearsal of the watchdog (tests only): a leg that never returns
        leg("rehearsal_hang")
        time.sleep(1e6)
    if want_extra:
        # N = 1: configs 3, 4 and the per-GPU shard of config 5 (524 288 envs: the N = 1 anchor of the weak-scaling curve
        # the 8-GPU run continues); N > 1: the config-5 shard
        for c in ([3, 4, 5] if not D.multi else [5]):
            if c == args.config:
                continue
            leg(f"config{c}")
            try:
                e = run_workload(c, args, D, min(args.steps, 20), min(args.warmup, 5), 3, with_cpu=False)
                detail["extra_configs"].append(e)
                measure_traffic(e, c)
                emit_detail(f"config{c}", e)
            except Exception as exc:  # noqa: BLE001
                detail["extra_configs"].append({"workload": CONFIGS[c][0], "config": c, "error": f"{type(exc).__name__}: {exc}"})
                break

    # the strong-scaling reading of the metric (64k envs IN TOTAL over the world) beside the weak one; at N = 1 also the
    # per-GPU shard of a 2 / 4 / 8-GPU world emulated on this GPU (no collectives): the measured basis of DESIGN.md section 7
    if not args.no_extra and not args.graph and not args.obs_f32 and args.config == 2:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import bench_legs as L

        leg("strong_scaling")
        try:
            if os.environ.get("FE_BENCH_HANG_STRONG") == "1":  # rehearsal of the watchdog (tests only)
                time.sleep(1e6)
            strong = L.strong_scaling_leg(args, D, args.steps, args.warmup)
            if not D.multi:
                strong["shard_preview"] = [L.strong_scaling_leg(args, D, args.steps, args.warmup, world=w_, rank=w_ - 1) for w_ in (2, 4, 8)]
        except Exception as exc:  # noqa: BLE001  (a per-rank failure leaves the other ranks in a collective: the watchdog ends that)
            strong = {"error": f"{type(exc).__name__}: {exc}"}
        detail["strong_scaling"] = strong
        emit_detail("strong_scaling", strong)
        if D.multi:
            leg("device_guard")
            try:
                mine = L.device_guard_check(D)
                recs = [None] * D.dist.get_world_size()
                D.dist.all_gather_object(recs, mine)
                dg = {"pass": all(bool(r_ and r_.get("pass")) for r_ in recs), "ranks": recs}
            except Exception as exc:  # noqa: BLE001
                dg = {"pass": False, "error": f"{type(exc).__name__}: {exc}"}
            detail["device_guard"] = dg
            emit_detail("device_guard", dg)
        elif want_extra:
            legs = detail["legs"]

            def run_leg(name, fn):
                leg(name)
                try:
                    rec = fn()
                except Exception as exc:  # noqa: BLE001
                    rec = {"error": f"{type(exc).__name__}: {exc}"}
                emit_detail(name, rec)
                detail.setdefault("legs_detail", {})[name] = rec
                return rec

            if args.redraw == "torch":
                # the build's own redraw contract on the same workload, the same ring policy, the same block protocol
                def devred():
                    dr = run_workload(2, args, D, args.steps, args.warmup, args.repeats, with_cpu=False, redraw="device")
                    if "error" not in dr:
                        dr.update(kernel=dr["roofline"]["kernel"], kernel_ms=dr["roofline"]["kernel_ms"], frac=dr["roofline"]["frac"])
                    return dr

                rec = run_leg("device_redraw", devred)
                legs["device_redraw"] = leg_summary("device_redraw", rec)
            fused = run_leg("fused_rollouts", lambda: {"legs": L.fused_rollout_legs(args)})
            for f_ in fused.get("legs", []):
                legs["fused_" + f_["form"]] = leg_summary("fused", f_)
            legs["reference_semantics"] = leg_summary("reference_semantics", run_leg("reference_semantics", lambda: L.reference_semantics_leg(args, args.steps, args.repeats)))
            legs["two_streams"] = leg_summary("two_streams", run_leg("two_streams", lambda: L.two_stream_leg(args, args.steps)))

    guard.current[0] = "teardown"
    if D.dist is not None:
        D.barrier()  # every rank is through with its measurements ...
        torch.cuda.synchronize()  # ... and this rank's share of that barrier has left the device before it may go
    note("done")
    # The line goes out BEFORE the process group is torn down: whatever the teardown does (it has aborted once in a rehearsal),
    # the record is on stdout.  Nothing can follow it there -- fd 1 has pointed at stderr since main() began.
    guard.print_once()
    if D.dist is not None:
        try:
            D.dist.destroy_process_group()
        except Exception as exc:  # noqa: BLE001
            print(f"bench.py: destroy_process_group: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
    guard.disarm()
    if D.dist is not None:
        # the measurements are done and printed, the process group is shut down: leave without running the interpreter's and the
        # communication library's static destructors against each other
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
