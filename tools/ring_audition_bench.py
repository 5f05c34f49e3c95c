"""GPU box: step time of a bench config with and without the opt-in observation-ring audition.

    python tools/ring_audition_bench.py <config> <extra candidates>
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(cfg: int, aud: int) -> None:
    import torch

    import finenvs_amd
    from bench import CONFIGS, make_series

    name, N, A, W = CONFIGS[cfg]
    prices, day_id, _ = make_series(A)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234,
                                    obs_buffers=2, obs_audition=aud)
    print("audition:", getattr(env, "obs_audition", None))
    g = torch.Generator(device="cuda").manual_seed(7)
    actions = [(torch.rand((N, A), generator=g, device="cuda") * 2 - 1).float() for _ in range(8)]
    env.reset()
    ts = []
    K = 10 if cfg > 2 else 100
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(K):
            env.step(actions[i % 8])
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / K * 1e3)
    print(f"config {cfg} audition {aud}: {statistics.median(ts):.1f} us/step")


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))
