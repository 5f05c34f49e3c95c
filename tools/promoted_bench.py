"""GPU box: step time of the promoted (float64-action) path beside the f32 fast path, configs 2 and 3.

    python tools/promoted_bench.py
"""
import os, sys, torch, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd
from bench import CONFIGS, make_series
for cfg, f32obs in ((2, False), (2, True), (3, False)):
    name, N, A, W = CONFIGS[cfg]
    name += " (f32 observations: the promoted path takes the unpipelined tile loop)" if f32obs else ""
    prices, day_id, _ = make_series(A)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1, obs_buffers=2,
                                    obs_dtype=torch.float32 if f32obs else torch.float64)
    info0 = env.launch_info()
    g = torch.Generator(device="cuda:0").manual_seed(7)
    a32 = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
    a64 = [x.double() for x in a32]
    K = 600 if cfg == 2 else 24
    def run(acts):
        for i in range(K): env.step(acts[i % 8])   # settle
        out = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(K): env.step(acts[i % 8])
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / K * 1e3)
        return statistics.median(out)
    t32 = run(a32)
    t64 = run(a64)          # promotes the env
    t32p = run(a32)         # f32 actions on the promoted env
    print(f"   launch_info before / after the promotion: {info0} / {env.launch_info()}")
    print(f"{name}: f32 actions {t32:.2f} us/step | f64 actions (promoted) {t64:.2f} | f32 actions on the promoted env {t32p:.2f}")
