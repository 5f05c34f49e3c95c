"""GPU box: the FORMs of the single-asset f64 step kernel (0 lean, 1 full, 2 lean + host flag, 3 full + host flag) on ONE env and ONE
observation ring, trains of back-to-back C-ABI launches, interleaved rounds (config 2: 64k envs x W64, f64 observations).

    python tools/form_ab.py [rounds]

Every arm is labelled by the instantiation launch_env (csrc/fe_env.hip) really dispatches it to: the FULL forms are reached only
with trajectory DESCRIPTORS (or statistics / evaluate mode) -- since round 5 the action copy alone stays on the lean forms, so
"step_traj with the action copy" is a lean-form arm of its own (the round-5 version of this tool still printed
those arms as FORM 1 / FORM 3).
"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finenvs_amd
from bench import CONFIGS, make_series

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
name, N, A, W = CONFIGS[2]
prices, day_id, _ = make_series(A)
from finenvs_amd import _lib
variant = os.environ.get("FORM_AB_LIB")  # an experiment build (finenvs_amd.csrc.build.build_variant), by name
native = _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", f"libfinenvs_amd.{variant}.so")) if variant else None
print("library:", variant or "product")
env = finenvs_amd.TimeSeriesEnv(prices=prices, day_id=day_id, num_intervals=W, num_envs=N, redraw="torch", seed=1, obs_buffers=2, _native=native)
g = torch.Generator(device="cuda:0").manual_seed(7)
acts = [(torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1).float() for _ in range(8)]
rew = torch.empty((N,), dtype=torch.float64, device="cuda:0")
done = torch.empty((N,), dtype=torch.int32, device="cuda:0")
acopy = torch.empty((N, A), dtype=torch.float32, device="cuda:0")
L, h, st = env._lib, env._handle_v, torch.cuda.current_stream().cuda_stream
obs = [t.data_ptr() for t in env._obs_ring]
ap = [a.data_ptr() for a in acts]
seq = [0]
# arms 4 / 5 = forms 2 / 3 with the flag word in DEVICE memory (nobody polls it in a train): separates the cost of the
# host-memory store (a PCIe write whose acknowledgement the wave's in-order vmcnt has to wait for) from the reversed tile walk
import ctypes
dev_flag = torch.zeros((1,), dtype=torch.int64, device="cuda:0")
dev_flag_p = ctypes.c_void_p(dev_flag.data_ptr())


src = torch.empty((N,), dtype=torch.int64, device="cuda:0")       # trajectory descriptors: what selects the FULL forms
pos = torch.empty((N, A), dtype=torch.float64, device="cuda:0")

# arm -> (label, entry point, action copy, descriptors, host flag: None / "host" / "device")
ARM_SPEC = {
    "F0":        ("FORM 0  lean                      fe_env_step", "step", False, False, None),
    "F0+acopy":  ("FORM 0  lean + action copy        fe_env_step_traj(actions_out)", "traj", True, False, None),
    "F1":        ("FORM 1  full (descriptors)        fe_env_step_traj(actions_out, obs_src, obs_pos)", "traj", True, True, None),
    "F2":        ("FORM 2  lean + host flag          fe_env_step_notify", "step", False, False, "host"),
    "F2+acopy":  ("FORM 2  lean + flag + action copy fe_env_step_traj_notify(actions_out)   <- bench.py's headline", "traj", True, False, "host"),
    "F3":        ("FORM 3  full + host flag          fe_env_step_traj_notify(actions_out, obs_src, obs_pos)", "traj", True, True, "host"),
    "F2dev":     ("FORM 2  lean, flag word in DEVICE memory", "step", False, False, "device"),
    "F3dev":     ("FORM 3  full, flag word in DEVICE memory", "traj", True, True, "device"),
}


def launch(arm, i):
    _, entry, ac, desc, flagkind = ARM_SPEC[arm]
    a, o, r, d = ap[i % 8], obs[i % 2], rew.data_ptr(), done.data_ptr()
    outs = (acopy.data_ptr() if ac else None, src.data_ptr() if desc else None, pos.data_ptr() if desc else None)
    if flagkind is None:
        return L.fe_env_step(h, a, o, r, d, st) if entry == "step" else L.fe_env_step_traj(h, a, o, r, d, *outs, st)
    seq[0] += 1
    flag = env._flag if flagkind == "host" else dev_flag_p
    if entry == "step":
        return L.fe_env_step_notify(h, a, o, r, d, flag, seq[0], st)
    return L.fe_env_step_traj_notify(h, a, o, r, d, *outs, flag, seq[0], st)


def train(form, k=400):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = 0
    for i in range(k):
        rc |= launch(form, i)
    e1.record()
    torch.cuda.synchronize()
    assert rc == 0
    return e0.elapsed_time(e1) / k * 1e3


ARMS = tuple(ARM_SPEC)
for f in ARMS:
    train(f, 800)  # settle
res = {f: [] for f in ARMS}
for r in range(rounds):
    for f in (ARMS if r % 2 == 0 else ARMS[::-1]):
        res[f].append(train(f))
for f in ARMS:
    v = res[f]
    print(f"{ARM_SPEC[f][0]}: median {statistics.median(v):.2f} us  min {min(v):.2f}  max {max(v):.2f}   {['%.2f' % x for x in v]}")
