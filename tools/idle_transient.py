"""GPU box, under `rocprofv3 --kernel-trace`: does the step kernel slow down for a few milliseconds shortly after the GPU
leaves an idle period, whoever issues the launches?  Phases (each after 1 s of host sleep, separated by a reset launch):
  1. a train of 600 back-to-back C-ABI launches (kernel_interval_ms's way)
  2. 600 env.step() calls with rotating trajectory slots (the timed loop's way)
  3. a train again
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/idle_transient.py [config]
    python tools/idle_transient.py --digest OUT
"""
import csv
import glob
import os
import sys
import time

K = 600
NAMES = ["train (C ABI, back to back)", "loop (env.step, Python-issued, trajectory slots)", "train again"]


def digest(out):
    f = glob.glob(os.path.join(out, "*", "*_kernel_trace.csv"))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    phases, cur = [], None
    for r in rows:
        nm = r["Kernel_Name"]
        if "fe_env_kernel" not in nm:
            continue
        if ", true, 0>" in nm:
            cur = []
            phases.append(cur)
        elif cur is not None:
            cur.append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"])))
    phases = [p for p in phases if len(p) == K][-3:]
    for name, d in zip(NAMES, phases):
        t0 = d[0][1]
        print(name)
        for i in range(0, K, 50):
            ch = d[i:i + 50]
            print(f"   launches {i:3d}-{i + 49:3d}  t = {(ch[0][1] - t0) / 1e6:6.2f} ms   avg {sum(c[0] for c in ch) / len(ch):6.2f} us   min {min(c[0] for c in ch):6.2f}  max {max(c[0] for c in ch):6.2f}")


def main():
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import finenvs_amd
    from bench import CONFIGS, make_series
    from finenvs_amd import _lib
    from finenvs_amd.trajectory import TrajectoryBuffer

    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    name, N, A, W = CONFIGS[cfg]
    prices, day_id, _ = make_series(A)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=2)
    g = torch.Generator(device="cuda:0").manual_seed(7)
    actions = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
    traj = TrajectoryBuffer(16, N, A, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    obs_b = [t.data_ptr() for t in env._obs_ring]
    rew = torch.empty((N,), dtype=torch.float64, device="cuda:0")
    done = torch.empty((N,), dtype=torch.int32, device="cuda:0")
    act = torch.empty((N, A), dtype=torch.float32, device="cuda:0")
    fn, h = env._lib.fe_env_step_traj, env._handle_v

    def train():
        for i in range(K):
            rc = fn(h, actions[i % 8].data_ptr(), obs_b[i % 2], rew.data_ptr(), done.data_ptr(), act.data_ptr(), None, None, stream)
        _lib.check(rc)

    def loop():
        for i in range(K):
            if traj.full():
                traj.clear()
            a, r, d = traj.next_slot()
            env.step(actions[i % 8], rewards_out=r, dones_out=d, actions_out=a)

    for rnd in range(2):
        for fnc in (train, loop, train):
            torch.cuda.synchronize()
            time.sleep(1.0)
            _lib.check(env._lib.fe_env_reset_obs(env._handle, obs_b[0], stream))
            fnc()
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--digest":
        digest(sys.argv[2])
    else:
        main()
