"""GPU box: sweep tile size / grid overrides for one bench config, event-timed, one process."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd
from bench import CONFIGS, make_series
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
name, N, A, W = CONFIGS[cfg]
prices, day_id, _ = make_series(A)
g = torch.Generator(device="cuda:0").manual_seed(7)
actions = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
obs_bytes = N * W * 5 * A * 8
def run(eb, grid, steps):
    for k, v in (("FE_TILE_ENVS", eb), ("FE_GRID", grid)):
        if v: os.environ[k] = str(v)
        else: os.environ.pop(k, None)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device",
                                    obs_buffers=2 if obs_bytes < 100e9 else 1)
    env.reset()
    for i in range(10): env.step(actions[i % 8])
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize(); evs[0].record()
    for i in range(steps):
        env.step(actions[i % 8]); evs[i + 1].record()
    torch.cuda.synchronize()
    iv = np.asarray([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
    info = env.launch_info()
    print(f"EB={info['tile_envs']:4d} grid={info['grid']:5d} lds={info['lds_bytes']:6d}  mean={iv.mean()*1e3:9.2f} us  med={np.median(iv)*1e3:9.2f}  min={iv.min()*1e3:9.2f}  obs-write={obs_bytes/np.median(iv)/1e9:6.2f} TB/s", flush=True)
    del env
steps = 200 if cfg <= 2 else 20
ebs = [int(x) for x in os.environ['SWEEP_EBS'].split(',')] if os.environ.get('SWEEP_EBS') else ([None, 4, 8, 11, 16, 22, 32, 44, 64, 128] if A == 1 else [None, 1, 2, 4, 8])
for eb in ebs:
    run(eb, None, steps)
if not os.environ.get('SWEEP_EBS'):
    for grid in (512, 768, 1024, 1280, 1536, 2048):
        run(16 if A == 1 else 8, grid, steps)
