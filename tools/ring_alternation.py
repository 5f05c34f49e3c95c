"""GPU box, under `rocprofv3 --kernel-trace`: is the 28 / 31 us alternation of back-to-back step launches over a ring of two
observation buffers a property of the BUFFERS (placement) or of the ALTERNATION (what the previous launch left in the
memory-side cache)?  Phases of K back-to-back fe_env_step_traj launches, separated by one fe_env_reset_obs launch (another
kernel name: the phase marker in the trace):  A only | B only | A,B alternating | A,B,C round-robin | A,A,B,B | C only.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/ring_alternation.py [config]
    python tools/ring_alternation.py --digest OUT      (prints per phase: average duration by position in the pattern)
"""
import csv
import glob
import os
import sys

PHASES = [("A only", "A"), ("B only", "B"), ("A,B alternating", "AB"), ("A,B,C round-robin", "ABC"), ("A,A,B,B", "AABB"), ("C only", "C"),
          ("A,B alternating (again)", "AB")]
K = 240
if sys.argv[-1] in ('3', '4', '5'):  # multi-asset configs: launches of 3.5 - 25 ms
    K = 24


def digest(out):
    f = glob.glob(os.path.join(out, "*", "*_kernel_trace.csv"))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    phases, cur = [], None
    for r in rows:
        nm = r["Kernel_Name"]
        if "fe_env_kernel" not in nm:
            continue
        if ", true, 0>" in nm:  # RESET_ONLY: the phase marker
            cur = []
            phases.append(cur)
        elif cur is not None:
            cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    phases = [p for p in phases if len(p) == K]
    for (name, pat), d in zip(PHASES, phases[-len(PHASES):]):
        d = d[K // 6:]  # skip the start of the train
        by_pos = [d[i::len(pat)] for i in range(len(pat))]
        print(f"{name:28s} avg {sum(d) / len(d):6.2f} us   by position in the pattern: " +
              "  ".join(f"{pat[i]}={sum(v) / len(v):.2f}" for i, v in enumerate(by_pos)))


def main():
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import finenvs_amd
    from bench import CONFIGS, make_series
    from finenvs_amd import _lib

    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    name, N, A, W = CONFIGS[cfg]
    prices, day_id, _ = make_series(A)
    env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="device", seed=1234, obs_buffers=3)
    g = torch.Generator(device="cuda:0").manual_seed(7)
    actions = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
    stream = torch.cuda.current_stream().cuda_stream
    bufs = dict(zip("ABC", (t.data_ptr() for t in env._obs_ring)))
    rew = torch.empty((N,), dtype=torch.float64, device="cuda:0")
    done = torch.empty((N,), dtype=torch.int32, device="cuda:0")
    act = torch.empty((N, A), dtype=torch.float32, device="cuda:0")
    fn, h = env._lib.fe_env_step_traj, env._handle_v
    for rnd in range(2):  # the second round is the one digested
        for _, pat in PHASES:
            torch.cuda.synchronize()
            _lib.check(env._lib.fe_env_reset_obs(env._handle, bufs["A"], stream))
            for i in range(K):
                rc = fn(h, actions[i % 8].data_ptr(), bufs[pat[i % len(pat)]], rew.data_ptr(), done.data_ptr(), act.data_ptr(), None, None, stream)
            _lib.check(rc)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--digest":
        digest(sys.argv[2])
    else:
        main()
