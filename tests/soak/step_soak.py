"""GPU box: long random sweep of the step kernel against the oracle, bit for bit, across shapes AND across the ways a
launch can be made -- plain (lean form), with trajectory outputs / episode statistics / evaluate mode (full form),
fe_env_step_notify (notify form, last tile first).  Complements the fixed cases of tests/test_hip_parity.py.

    python tests/soak/step_soak.py [cases] [seed]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import finenvs_amd  # noqa: E402
from finenvs_amd import _lib  # noqa: E402
from finenvs_amd.data import synthetic  # noqa: E402
from finenvs_amd.stats import EpisodeStats  # noqa: E402
from oracle import fe_oracle as fo  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
fo.build()


def bits(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        raise AssertionError(f"{what}: shape / dtype {a.shape} {a.dtype} vs {b.shape} {b.dtype}")
    if a.dtype.kind == "f":  # a NaN's sign / payload is not part of the contract
        an, bn = np.isnan(a), np.isnan(b)
        ok = np.array_equal(an, bn) and np.array_equal(a[~an].view(np.uint8), b[~bn].view(np.uint8))
    else:
        ok = a.tobytes() == b.tobytes()
    if not ok:
        bad = np.argwhere(a != b)[:3]
        raise AssertionError(f"{what}: differs at {bad.tolist()}")


t2n = lambda t: t.detach().cpu().numpy()
done_cases = 0
for case in range(cases):
    A = int(rng.choice([1, 1, 1, 2, 3, 7, 30, 64]))
    W = int(rng.choice([1, 3, 4, 7, 8, 16, 33, 64, 100]))
    N = int(rng.integers(1, 5000 if A == 1 else 400))
    days, bars = int(rng.integers(3, 8)), int(rng.integers(12, 60))
    drop = float(rng.choice([0.0, 0.0, 0.05, 0.2]))
    f32 = bool(rng.integers(0, 2))
    evaluate = bool(rng.integers(0, 4) == 0)
    balance = float(rng.choice([10000, 2000, 600, 150]))
    mode = (str(rng.choice(["plain", "traj", "stats", "notify", "trajnotify", "promoted"])) if not evaluate
            else str(rng.choice(["plain", "traj", "promoted"])))
    prices, day_id, _ = synthetic.synthetic_series(days, A, bars, int(rng.integers(0, 10**6)), drop)
    try:
        P, LR, *_ = fo.tables_from_series(prices, day_id, W)
    except Exception:  # noqa: BLE001  (no day has W bars of history)
        continue
    D = P.shape[0]
    if D < 1 or P.shape[1] <= W:
        continue
    idx = rng.integers(0, D, N)
    kw = dict(num_intervals=W, evaluate=evaluate, starting_balance=balance)
    if rng.integers(0, 2):  # economics that are not exact in f32 (the promoted path's f64 products differ from the f32 ones)
        kw.update(initial_margin_requirement=float(rng.choice([1.4, 1.1, 1.7])), per_share_commission=float(rng.choice([0.035, 0.013])),
                  maintenance_margin_requirement=float(rng.choice([0.25, 0.3])))
    ref = fo.OracleEnv(P, LR, env_indices=idx, obs_f32=f32, redraw_mode=1, seed=case, **kw)
    env = finenvs_amd.TimeSeriesEnv(tables=(P, LR), env_indices=idx, redraw="device", seed=case,
                                    obs_dtype=torch.float32 if f32 else torch.float64, **kw)
    ref.redraw_counter[0] = 1
    lib = env._lib
    stats = EpisodeStats(env) if mode == "stats" else None
    ref_stats = fo.EpisodeStatsOracle(N, env._eval_env) if mode == "stats" else None
    promote_from = int(rng.integers(0, 6)) if mode == "promoted" else None  # first step with float64 actions
    flag = None
    if mode in ("notify", "trajnotify"):
        flag = C.c_void_p()
        _lib.check(lib.fe_host_flag_create(C.byref(flag)))
        word = C.c_uint64.from_address(flag.value)
    g = torch.Generator().manual_seed(case)
    steps = int(min(2.5 * bars, 90))
    dev = env.device
    for t in range(steps):
        a = (torch.rand((N, A), generator=g) * 2 - 1).float()
        if t % 5 == 2:
            a = torch.sign(a)
        if mode == "promoted" and t >= promote_from and (t - promote_from) % 7 < 4:
            # float64 actions (the reference promotes its share tensors from here on; f32 steps in between stay promoted)
            a = (torch.rand((N, A), generator=g, dtype=torch.float64) * 2 - 1) if t % 5 != 2 else a.double()
        o_r, r_r, d_r, i_r = ref.step(a.numpy())
        ad = a.to(dev)
        if mode == "traj":
            src = torch.empty((N,), dtype=torch.int64, device=dev)
            pos = torch.empty((N, A), dtype=torch.float64, device=dev)
            aout = torch.empty((N, A), dtype=torch.float32, device=dev)
            o, r, d, i = env.step(ad, descriptors_out=(src, pos), actions_out=aout)
            bits(t2n(aout), a.numpy(), f"case {case} step {t} action copy")
            bits(t2n(env.render(src, pos)), o_r, f"case {case} step {t} rendered descriptors")
        elif mode == "trajnotify":
            o = torch.empty((N, W, 5 * A), dtype=env.obs_dtype, device=dev)
            r = torch.empty((N,), dtype=torch.float64, device=dev)
            d = torch.empty((N,), dtype=torch.int32, device=dev)
            src = torch.empty((N,), dtype=torch.int64, device=dev)
            pos = torch.empty((N, A), dtype=torch.float64, device=dev)
            aout = torch.empty((N, A), dtype=torch.float32, device=dev)
            _lib.check(lib.fe_env_step_traj_notify(env._handle, ad.data_ptr(), o.data_ptr(), r.data_ptr(), d.data_ptr(), aout.data_ptr(),
                                                   src.data_ptr(), pos.data_ptr(), flag, t + 1, torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            assert word.value == ((t + 1) << 1 | int(d_r[-1])), f"case {case} step {t}: flag {word.value:#x}"
            bits(t2n(aout), a.numpy(), f"case {case} step {t} action copy")
            bits(t2n(env.render(src, pos)), o_r, f"case {case} step {t} rendered descriptors")
            i = {}
        elif mode == "notify":
            o = torch.empty((N, W, 5 * A), dtype=env.obs_dtype, device=dev)
            r = torch.empty((N,), dtype=torch.float64, device=dev)
            d = torch.empty((N,), dtype=torch.int32, device=dev)
            _lib.check(lib.fe_env_step_notify(env._handle, ad.data_ptr(), o.data_ptr(), r.data_ptr(), d.data_ptr(), flag, t + 1,
                                              torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            assert word.value == ((t + 1) << 1 | int(d_r[-1])), f"case {case} step {t}: flag {word.value:#x}"
            i = {}
        else:
            o, r, d, i = env.step(ad)
        what = f"case {case} ({N}x{A}xW{W} {'f32' if f32 else 'f64'} {mode} {'eval' if evaluate else 'train'}) step {t}"
        bits(t2n(o), o_r, what + " obs")
        bits(t2n(r), r_r, what + " rewards")
        bits(t2n(d), d_r, what + " dones")
        bits(t2n(env.cash), ref.cash, what + " cash")
        bits(t2n(env.margin), ref.margin, what + " margin")
        bits(t2n(env.long_shares).astype(np.float32), ref.long, what + " long")  # (float64 on a promoted env, as the reference; same counts)
        bits(t2n(env.short_shares).astype(np.float32), ref.short, what + " short")
        bits(t2n(env._spot0), ref.spot0, what + " spot0")
        bits(t2n(env.env_indices), ref.env_idx, what + " day indices")
        if ref_stats is not None:
            ref_stats.step(r_r, d_r)
            bits(t2n(stats.running_returns), ref_stats.running, what + " running returns")
        if evaluate and mode == "plain":
            assert ("returns" in i) == ("returns" in i_r), what
            if "returns" in i:
                bits(t2n(i["returns"]), i_r["returns"], what + " returns")
    if stats is not None:
        got, want = stats.read(), ref_stats.read()
        for k in ("num_training_episodes", "mean_training_return", "std_dev_training_return", "evaluation_return"):
            same = got[k] == want[k] or (isinstance(got[k], float) and got[k] != got[k] and want[k] != want[k])
            assert same, f"case {case}: statistics {k}: {got[k]!r} vs the oracle's {want[k]!r}"
        stats.close()
    if flag is not None:
        torch.cuda.synchronize()
        _lib.check(lib.fe_host_flag_destroy(flag))
    done_cases += 1
    if done_cases % 20 == 0:
        print(f"{done_cases} cases ok (last: {N} envs x {A} assets x W{W}, {mode})", flush=True)
print(f"step soak: {done_cases} random cases, every step bit for bit against the oracle")
