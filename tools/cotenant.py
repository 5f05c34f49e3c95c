"""GPU box, under `rocprofv3 --kernel-trace`: what a co-resident kernel on ANOTHER stream does to the step kernel.  In an N > 1
rollout RCCL's all-gather kernels run on RCCL's stream beside the steps; the step kernel's grid equals the number of
workgroups the chip holds at once, so any CU slot a co-tenant holds leaves some workgroups waiting for others to finish.
Phases of K back-to-back step launches; during each, a squatter of B workgroups x 256 lanes (with `lds` bytes of LDS each)
is started on a second stream every 10 launches and holds its slots for `micros` us.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/cotenant.py [config]
    python tools/cotenant.py --digest OUT
"""
import csv
import ctypes as C
import glob
import os
import sys

PHASES = [(0, 0, 0.0), (8, 0, 200.0), (32, 0, 200.0), (32, 32768, 200.0), (128, 32768, 200.0), (32, 32768, 2000.0), (0, 0, 0.0)]
K = 200
if len(sys.argv) > 1 and sys.argv[-1] in ('3', '5'):  # multi-asset configs: launches of 3.5 / 13 ms, squatters that outlive several of them
    PHASES = [(0, 0, 0.0), (32, 32768, 9000.0), (128, 32768, 9000.0), (256, 65536, 9000.0), (0, 0, 0.0)]
    K = 40


def digest(out):
    f = glob.glob(os.path.join(out, "*", "*_kernel_trace.csv"))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    phases, cur = [], None
    for r in rows:
        nm = r["Kernel_Name"]
        if "fe_env_kernel" not in nm:
            continue
        if ", true, 0>" in nm or ", true, 0, " in nm:  # RESET_ONLY: phase marker
            cur = []
            phases.append(cur)
        elif cur is not None:
            cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    phases = [p for p in phases if len(p) == K][-len(PHASES):]
    for (b, lds, us), d in zip(PHASES, phases):
        d = sorted(d[K // 10:])
        print(f"squatter {b:4d} workgroups x 256, LDS {lds:6d} B, {us:6.0f} us every 10 launches:  step kernel median {d[len(d) // 2]:7.2f} us   "
              f"mean {sum(d) / len(d):7.2f}   p90 {d[int(0.9 * len(d))]:7.2f}   max {d[-1]:7.2f}")


def main():
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import finenvs_amd
    from bench import CONFIGS, make_series
    from finenvs_amd import _lib

    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    name, N, A, W = CONFIGS[cfg]
    Kc = K
    prices, day_id, _ = make_series(A)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=2)
    g = torch.Generator(device="cuda:0").manual_seed(7)
    actions = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
    co = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcotenant.so"))
    co.cotenant_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    side = torch.cuda.Stream()
    stream = torch.cuda.current_stream().cuda_stream
    obs_b = [t.data_ptr() for t in env._obs_ring]
    rew = torch.empty((N,), dtype=torch.float64, device="cuda:0")
    done = torch.empty((N,), dtype=torch.int32, device="cuda:0")
    act = torch.empty((N, A), dtype=torch.float32, device="cuda:0")
    fn, h = env._lib.fe_env_step_traj, env._handle_v
    # settle (clock transient after idle)
    for i in range(800 if cfg == 2 else 8):
        fn(h, actions[i % 8].data_ptr(), obs_b[i % 2], rew.data_ptr(), done.data_ptr(), act.data_ptr(), None, None, stream)
    for rnd in range(2):
        for b, lds, us in PHASES:
            torch.cuda.synchronize()
            _lib.check(env._lib.fe_env_reset_obs(env._handle, obs_b[0], stream))
            for i in range(Kc):
                if b and i % 10 == 5:
                    assert co.cotenant_launch(b, 256, lds, us, side.cuda_stream) == 0
                rc = fn(h, actions[i % 8].data_ptr(), obs_b[i % 2], rew.data_ptr(), done.data_ptr(), act.data_ptr(), None, None, stream)
            _lib.check(rc)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--digest":
        digest(sys.argv[2])
    else:
        main()
