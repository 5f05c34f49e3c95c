"""GPU box: what decides how fast an observation buffer is?  Config-3-sized buffers (20 GB), the render kernel
(reset(): the step's store stream without the accounting) timed into each.

    python tools/placement_study.py separate     # 10 separate torch allocations (what the ring audition sees)
    python tools/placement_study.py slab         # ONE 200 GB allocation carved into 10 windows at 20 GB steps
    python tools/placement_study.py offsets      # one 60 GB allocation, the same 20 GB window shifted by 0 .. 1 GiB
    python tools/placement_study.py contig | hipmalloc   # raw HIP allocations, physically contiguous / plain
    python tools/placement_study.py scan         # sixteenths of a contiguous and of two torch buffers
    python tools/placement_study.py sequence     # 13 x 20 GB in a row, twice; then behind 160 / 80 GB of ballast

Run each also under PYTORCH_HIP_ALLOC_CONF=expandable_segments:True (torch then maps fixed-size physical granules).
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402
from finenvs_amd import _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "separate"
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
name, N, A, W = CONFIGS[cfg]
prices, day_id, _ = make_series(A)
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234)
if os.environ.get("PS_TILE"):  # launch geometry override: PS_TILE=<envs per tile>[,<grid>]
    t, _, gr = os.environ["PS_TILE"].partition(",")
    print("launch:", env.set_launch(int(t), int(gr or 0)))
else:
    print("launch:", env.launch_info())
n = N * W * 5 * A  # f64 elements per observation
st = torch.cuda.current_stream().cuda_stream


def time_into(ptr, reps=5):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(env._lib.fe_env_reset_obs(env._handle, ptr, st))
        e1.record()
        e1.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


print(f"mode {mode}, config {cfg} ({name}), {n * 8 / 1e9:.1f} GB per buffer, allocator conf: "
      f"{os.environ.get('PYTORCH_HIP_ALLOC_CONF') or os.environ.get('PYTORCH_CUDA_ALLOC_CONF') or 'default'}", flush=True)
rows = []
if mode == "separate":
    bufs = [torch.empty((n,), dtype=torch.float64, device="cuda") for _ in range(10)]
    for rnd in range(2):
        for i, b in enumerate(bufs):
            rows.append((rnd, i, b.data_ptr(), time_into(b.data_ptr())))
elif mode == "slab":
    slab = torch.empty((10 * n,), dtype=torch.float64, device="cuda")
    for rnd in range(2):
        for i in range(10):
            p = slab.data_ptr() + i * n * 8
            rows.append((rnd, i, p, time_into(p)))
elif mode == "scan":
    # 1/16-size env (1.26 GB of observation) rendered into each sixteenth of three 20 GB buffers: one physically
    # contiguous, two from torch -- is a slow buffer slow everywhere, or in places?
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    pc = C.c_void_p()
    assert hip.hipExtMallocWithFlags(C.byref(pc), n * 8, 0x4) == 0
    t1 = torch.empty((n,), dtype=torch.float64, device="cuda")
    t2 = torch.empty((n,), dtype=torch.float64, device="cuda")
    small = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N // 16, redraw="device", seed=1234)
    whole = env
    for label, base in (("contiguous", pc.value), ("torch-1", t1.data_ptr()), ("torch-2", t2.data_ptr())):
        env = whole
        full = time_into(base)
        env = small
        parts = [time_into(base + k * (n // 16) * 8) for k in range(16)]
        print(f"{label:11s} whole buffer {full:8.1f} us ({n * 8 / full / 1e6:5.2f} TB/s); sixteenths (TB/s): "
              + " ".join(f"{(n // 16) * 8 / t / 1e6:5.2f}" for t in parts), flush=True)
    sys.exit(0)
elif mode == "sequence":
    # is a buffer's speed a function of WHEN (= where) it was allocated?  13 x 20 GB in a row, twice; then 160 GB of
    # ballast first and six buffers behind it
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]

    def alloc(nbytes):
        p = C.c_void_p()
        rc = hip.hipMalloc(C.byref(p), nbytes)
        return p.value if rc == 0 else None

    for label, ballast, count in (("A: 13 in a row", 0, 13), ("B: again", 0, 13), ("C: 160 GB ballast first", 160e9, 6), ("D: 80 GB ballast first", 80e9, 9)):
        bal = alloc(int(ballast)) if ballast else None
        ptrs = []
        for i in range(count):
            p = alloc(n * 8)
            if p is None:
                break
            ptrs.append(p)
        speeds = [n * 8 / time_into(p) / 1e6 for p in ptrs]
        print(f"{label:26s}: " + " ".join(f"{x:5.2f}" for x in speeds) + "  TB/s", flush=True)
        for p in ptrs:
            hip.hipFree(p)
        if bal:
            hip.hipFree(bal)
    sys.exit(0)
elif mode in ("contig", "hipmalloc", "uncached"):
    # raw HIP allocations: physically contiguous (hipDeviceMallocContiguous) / plain hipMalloc / uncached, outside torch's allocator
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    ptrs = []
    for i in range(10):
        p = C.c_void_p()
        rc = hip.hipMalloc(C.byref(p), n * 8) if mode == "hipmalloc" else hip.hipExtMallocWithFlags(C.byref(p), n * 8, 0x4 if mode == "contig" else 0x3)
        if rc != 0:
            print(f"allocation {i} failed with hipError {rc}", flush=True)
            break
        ptrs.append(p.value)
    for rnd in range(2):
        for i, p in enumerate(ptrs):
            rows.append((rnd, i, p, time_into(p)))
else:
    slab = torch.empty((3 * n,), dtype=torch.float64, device="cuda")
    for rnd in range(2):
        for i, off in enumerate([0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 64 << 20, 256 << 20, 1 << 30]):
            p = slab.data_ptr() + off
            rows.append((rnd, i, p, time_into(p)))
for rnd, i, p, t in rows:
    print(f"round {rnd} buffer {i:2d} at {p:#016x} (mod 1 GiB: {p % (1 << 30):#011x}): {t:9.1f} us  {n * 8 / t / 1e6:6.3f} TB/s", flush=True)
ts = [t for rnd, _, _, t in rows if rnd == 1]
print(f"spread: best {min(ts):.1f}, worst {max(ts):.1f} us ({(max(ts) / min(ts) - 1) * 100:.1f} %)")
