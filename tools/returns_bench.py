"""GPU box: the discounted-returns scan kernel (fe_traj_returns, buffer.py:80-100) against its HBM roofline:
reads rewards f64 + dones i32 + values f32, writes returns f32 + advantages f32 = 24 B per (t, env)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finenvs_amd import _lib  # noqa: E402
from finenvs_amd.trajectory import TrajectoryBuffer  # noqa: E402

if len(sys.argv) > 1:  # an experiment build by its tag (finenvs_amd.csrc.build.build_variant)
    VARIANT = _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", f"libfinenvs_amd.{sys.argv[1]}.so"))
    print("library:", sys.argv[1])
else:
    VARIANT = None

for T, N in ((16, 65536), (128, 65536), (16, 1048576), (128, 1048576), (512, 1048576)):
    buf = TrajectoryBuffer(T, N, 1, device="cuda:0")
    if VARIANT is not None:
        buf._lib = VARIANT
    buf.rewards.normal_(); buf.dones.zero_(); buf.t = T
    vals = torch.randn((T, N), device="cuda:0"); last = torch.randn((N,), device="cuda:0")
    buf.returns_and_advantages(vals, last)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        buf.returns_and_advantages(vals, last)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"T={T:4d} N={N:8d}: {ms*1e3:9.1f} us  {T*N*24/ms/1e9:7.2f} TB/s of 24 B per element (incl. two torch.empty + arg checks)", flush=True)
