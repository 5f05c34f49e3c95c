"""GPU box: how soon after the launch does fe_env_step_notify's host flag carry the evaluation env's done bit?
Config 3 (a 3.4 ms launch) and config 2 (29 us): host time from the launch call to the flag, against the kernel's duration."""
import ctypes as C
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd  # noqa: E402
from bench import CONFIGS, make_series  # noqa: E402
from finenvs_amd import _lib  # noqa: E402

for cfg in (2, 3):
    name, N, A, W = CONFIGS[cfg]
    prices, day_id, _ = make_series(A)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1, obs_buffers=1)
    lib = env._lib
    flag = C.c_void_p()
    _lib.check(lib.fe_host_flag_create(C.byref(flag)))
    word = C.c_uint64.from_address(flag.value)
    a = (torch.rand((N, A), device="cuda") * 2 - 1).float()
    obs = env._obs_ring[0]
    rew = torch.empty((N,), dtype=torch.float64, device="cuda")
    done = torch.empty((N,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    to_flag, to_end = [], []
    for k in range(1, 41):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.fe_env_step_notify(env._handle, a.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), flag, k, st))
        while word.value >> 1 != k:
            pass
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if k > 10:
            to_flag.append((t1 - t0) * 1e6)
            to_end.append((t2 - t0) * 1e6)
    print(f"config {cfg} ({name}): launch call -> flag visible {statistics.median(to_flag):8.1f} us (min {min(to_flag):.1f}); "
          f"launch call -> kernel finished {statistics.median(to_end):8.1f} us", flush=True)
    torch.cuda.synchronize()
    _lib.check(lib.fe_host_flag_destroy(flag))
    del env
    torch.cuda.empty_cache()
