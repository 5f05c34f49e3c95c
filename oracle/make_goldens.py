#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

TEST INFRASTRUCTURE ONLY -- runs where /root/reference exists (never on the
GPU box, never from the product).  The reference's Python is imported from a
scratch copy under /tmp (it writes a bounds-cache JSON next to its CSVs,
TSE:122-124, and __pycache__), with a six-line stand-in for the one class of
the absent ``gym`` package it touches (``spaces.Box``, TSE:224-234).  Nothing
of the reference is written into this repository: only arrays (inputs we
generated ourselves + the outputs the reference computed from them).

    python oracle/make_goldens.py            # rewrites tests/golden/*.npz

Inputs are the build's own seeded synthetic CSVs (finenvs_amd.data.synthetic),
plus one real-data case: the reference's smallest test fixture
(finenvs/data/OIH/dummy.csv, 1050 rows), whose market-hours rows are stored in
the fixture as a plain (T, 4) array.
"""
from __future__ import annotations

import glob
import json
import os
import shutil
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from finenvs_amd.data import synthetic  # noqa: E402

REF = "/root/reference"
WORK = "/tmp/fe_oracle_work"
GOLD = os.path.join(REPO, "tests", "golden")


def setup_reference():
    if os.path.isdir(WORK):
        shutil.rmtree(WORK)
    os.makedirs(WORK)
    shutil.copytree(
        os.path.join(REF, "finenvs"),
        os.path.join(WORK, "refcopy", "finenvs"),
        ignore=shutil.ignore_patterns("isaac_gym_envs", "__pycache__", "data"),
    )
    os.makedirs(os.path.join(WORK, "data"))
    # stand-in for gym.spaces.Box: the env only stores these objects
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")

    class Box:  # noqa: D401
        def __init__(self, low, high, dtype=None):
            self.low, self.high, self.dtype, self.shape = low, high, dtype, np.shape(low)

    spaces.Box = Box
    gym.spaces = spaces
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = spaces
    sys.path.insert(0, os.path.join(WORK, "refcopy"))


def data_dir(name: str) -> str:
    # the path must contain the substring "data" to be taken verbatim (TSE:47-51)
    return os.path.join(WORK, "data", name)


def make_env(name: str, W: int, **kw):
    from finenvs.environments.time_series_env import TimeSeriesEnv

    d = data_dir(name)
    for f in glob.glob(os.path.join(d, "*_bounds_cache.json")):
        os.remove(f)  # the cache ignores num_intervals (TSE:104-106)
    env = TimeSeriesEnv(d, "dummy", num_intervals=W, device_id=-1, **kw)
    with open(os.path.join(d, "dummy_bounds_cache.json")) as f:
        cache = json.load(f)
    return env, cache


def scale_env(env, N: int):
    """Appendix B of SURVEY.md: replicate to N envs, env n -> day n mod D."""
    D = env.price_environments.shape[0]
    W = env.num_intervals
    S = env.starting_balance
    env.env_indices = torch.arange(N) % D
    env.num_envs = N
    env.env_pointers = torch.zeros((N,), dtype=torch.int64)
    env.env_spots = torch.arange(0, W).repeat(N, 1)
    env.cash = S * torch.ones((N, 1))
    env.long_shares = torch.zeros((N, 1))
    env.short_shares = torch.zeros((N, 1))
    env.margin = torch.zeros((N, 1))
    if env.evaluate:
        env.reset_evaluation_metrics()


def write_case_csv(name, num_days, bars, seed=1234, drop=0.0, premarket=2, asset=0, num_assets=1,
                   spikes=None):
    prices, day_id, minute = synthetic.synthetic_series(num_days, num_assets, bars, seed, drop)
    if spikes:
        rng = np.random.default_rng(seed + 99)
        rows = rng.choice(prices.shape[0], size=spikes, replace=False)
        for r in rows:
            f = rng.uniform(1.3, 2.5)
            for a in range(num_assets):
                prices[r, 4 * a + 1] = np.round(prices[r, 4 * a + 1] * f, 4)  # High
                if rng.random() < 0.5:
                    prices[r, 4 * a + 3] = np.round(prices[r, 4 * a + 3] * f, 4)  # Close
    for a in range(num_assets):
        nm = name if num_assets == 1 else f"{name}_a{a}"
        synthetic.write_csv(os.path.join(data_dir(nm), "dummy.csv"), prices, day_id, minute, a, premarket)
    return prices, day_id, minute


def tables_case(fname, name, W, prices, day_id, minute):
    env, cache = make_env(name, W, evaluate=True)
    np.savez_compressed(
        os.path.join(GOLD, fname),
        W=np.int64(W),
        series_prices=prices,
        series_day_id=day_id,
        series_minute=minute,
        ref_dataset=env.dataset.numpy(),
        ref_log_return_dataset=env.log_return_dataset.numpy(),
        ref_start_indices=np.asarray(cache["start_indices"], dtype=np.int64),
        ref_stop_indices=np.asarray(cache["stop_indices"], dtype=np.int64),
        ref_max_length=np.int64(cache["max_length"]),
        ref_price_environments=env.price_environments.numpy(),
        ref_log_return_environments=env.log_return_environments.numpy(),
    )
    print(fname, tuple(env.price_environments.shape))
    return env


def state_of(env):
    return dict(
        cash=env.cash.squeeze(1).numpy().copy(),
        margin=env.margin.squeeze(1).double().numpy().copy(),
        long=env.long_shares.squeeze(1).numpy().copy(),
        short=env.short_shares.squeeze(1).numpy().copy(),
        spot0=env.env_spots[:, 0].numpy().copy(),
        env_idx=env.env_indices.numpy().copy(),
    )


def rollout(env, T, action_seed=7, action_kind="uniform", stop_on_returns=False, full_obs=True):
    N = env.num_envs
    g = torch.Generator().manual_seed(action_seed)
    rec = {k: [] for k in ("actions", "rewards", "dones", "cash", "margin", "long", "short", "spot0",
                           "env_idx", "obs")}
    obs0 = env.reset()
    init = state_of(env)
    returns = None
    for t in range(T):
        if action_kind == "uniform":
            a = torch.rand((N, 1), generator=g) * 2 - 1
        elif action_kind == "bangbang":
            a = torch.where(torch.rand((N, 1), generator=g) < 0.5, -1.0, 1.0) * torch.ones((N, 1))
            hold = torch.rand((N, 1), generator=g) < 0.6
            a = torch.where(hold, torch.zeros_like(a), a)
        elif action_kind == "edge":
            a = torch.rand((N, 1), generator=g) * 2 - 1
            specials = [float("nan"), float("inf"), float("-inf"), 7.5, -123.0, -0.0, 1e-30, 0.0909090909]
            if t % 9 == 4:
                for k, sp in enumerate(specials):
                    a[(t + 3 * k) % N, 0] = sp
        else:
            raise ValueError(action_kind)
        a = a.float()
        obs, rew, done, info = env.step(a)
        st = state_of(env)
        rec["actions"].append(a.squeeze(1).numpy().copy())
        rec["rewards"].append(rew.numpy().copy())
        rec["dones"].append(done.numpy().copy())
        for k in ("cash", "margin", "long", "short", "spot0", "env_idx"):
            rec[k].append(st[k])
        rec["obs"].append(obs.numpy().copy() if full_obs else obs[:, -1, :].numpy().copy())
        if "returns" in info:
            returns = info["returns"].numpy().copy()
            if stop_on_returns:
                break
    out = {k: np.stack(v) for k, v in rec.items()}
    out["obs_reset"] = obs0.numpy().copy()
    for k, v in init.items():
        out["init_" + k] = v
    if returns is not None:
        out["returns"] = returns
    return out


def save_rollout(fname, env, roll, extra=None):
    meta = dict(
        W=np.int64(env.num_intervals),
        N=np.int64(env.num_envs),
        evaluate=np.int64(bool(env.evaluate)),
        max_shares=np.int64(env.max_shares),
        starting_balance=np.float64(env.starting_balance),
        commission=np.float64(env.per_share_commission),
        imr=np.float64(env.initial_margin_requirement),
        mmr=np.float64(env.maintenance_margin_requirement),
        prices=env.price_environments.numpy(),
        logret=env.log_return_environments.numpy(),
    )
    meta.update(roll)
    if extra:
        meta.update(extra)
    np.savez_compressed(os.path.join(GOLD, fname), **meta)
    nd = int(roll["dones"].sum())
    print(f"{fname}: steps={roll['dones'].shape[0]} N={env.num_envs} dones={nd}")


def main():
    os.makedirs(GOLD, exist_ok=True)
    setup_reference()

    # ---------------- tables (a18-a20) ----------------
    p, d, m = write_case_csv("SYN_full", 5, 40)
    tables_case("tables_full.npz", "SYN_full", 8, p, d, m)
    p, d, m = write_case_csv("SYN_ragged", 7, 40, seed=77, drop=0.10)
    tables_case("tables_ragged.npz", "SYN_ragged", 8, p, d, m)
    p, d, m = write_case_csv("SYN_skip2", 6, 40, seed=5)
    tables_case("tables_skip2.npz", "SYN_skip2", 50, p, d, m)
    # real data: the reference's smallest fixture; store its market-hours rows as arrays
    os.makedirs(data_dir("OIH"))
    shutil.copy(os.path.join(REF, "finenvs", "data", "OIH", "dummy.csv"), os.path.join(data_dir("OIH"), "dummy.csv"))
    env, cache = make_env("OIH", 32, evaluate=True)
    df = env.dataframe
    dates = df["Date"].values
    uniq = {s: i for i, s in enumerate(dict.fromkeys(dates))}
    day_id = np.asarray([uniq[s] for s in dates], dtype=np.int64)
    minute = np.asarray([int(t[:2]) * 60 + int(t[3:5]) for t in df["Time"].values], dtype=np.int64)
    np.savez_compressed(
        os.path.join(GOLD, "tables_oih.npz"),
        W=np.int64(32),
        series_prices=env.dataset.numpy(),
        series_day_id=day_id,
        series_minute=minute,
        ref_dataset=env.dataset.numpy(),
        ref_log_return_dataset=env.log_return_dataset.numpy(),
        ref_start_indices=np.asarray(cache["start_indices"], dtype=np.int64),
        ref_stop_indices=np.asarray(cache["stop_indices"], dtype=np.int64),
        ref_max_length=np.int64(cache["max_length"]),
        ref_price_environments=env.price_environments.numpy(),
        ref_log_return_environments=env.log_return_environments.numpy(),
    )
    print("tables_oih.npz", tuple(env.price_environments.shape))

    # ---------------- rollouts (a2-a17) ----------------
    write_case_csv("SYN_roll", 7, 40, seed=1234)
    # native training mode: N = D + 1, eval env redraws from torch's global generator
    torch.manual_seed(123)
    env, _ = make_env("SYN_roll", 8)
    roll = rollout(env, 150)
    save_rollout("rollout_train_native.npz", env, roll, {"torch_seed": np.int64(123)})

    # replicated to N = 64 (training mode; last env redraws)
    torch.manual_seed(321)
    env, _ = make_env("SYN_roll", 8)
    scale_env(env, 64)
    roll = rollout(env, 150)
    save_rollout("rollout_train_n64.npz", env, roll, {"torch_seed": np.int64(321)})

    # evaluate mode, native N = D, until info["returns"] appears
    env, _ = make_env("SYN_roll", 8, evaluate=True)
    roll = rollout(env, 400, stop_on_returns=True)
    assert "returns" in roll
    save_rollout("rollout_eval.npz", env, roll)

    # evaluate mode on ragged days: envs terminate at different steps, so rewards of
    # terminated envs get zeroed while others still run (TSE:526-528)
    write_case_csv("SYN_rag_roll", 8, 40, seed=42, drop=0.15)
    env, _ = make_env("SYN_rag_roll", 8, evaluate=True)
    roll = rollout(env, 400, stop_on_returns=True)
    assert "returns" in roll
    save_rollout("rollout_eval_ragged.npz", env, roll)

    # stress: small balances + price spikes -> illegal long/short, margin call, bankruptcy
    write_case_csv("SYN_stress", 7, 40, seed=9, spikes=40)
    for bal in (60, 150, 400, 1500):
        env, _ = make_env("SYN_stress", 8, starting_balance=bal, evaluate=True)
        scale_env(env, 48)
        roll = rollout(env, 120, action_seed=1000 + bal, action_kind="bangbang" if bal in (60, 1500) else "uniform")
        save_rollout(f"rollout_stress_{bal}.npz", env, roll)
    # non-default economics
    env, _ = make_env("SYN_stress", 8, starting_balance=2500.0, max_shares=9, per_share_commission=0.035,
                      initial_margin_requirement=1.4, maintenance_margin_requirement=0.3, evaluate=True)
    scale_env(env, 48)
    roll = rollout(env, 120, action_seed=5)
    save_rollout("rollout_econ.npz", env, roll)

    # real data: OIH, ragged NaN-padded days
    torch.manual_seed(2024)
    env, _ = make_env("OIH", 32)
    roll = rollout(env, 700, full_obs=False)
    save_rollout("rollout_oih.npz", env, roll, {"torch_seed": np.int64(2024)})

    # edge-case actions: NaN, +-inf, out-of-range, -0.0 (a diverged policy must not crash or desync parity)
    env, _ = make_env("SYN_roll", 8, evaluate=True, starting_balance=3000)
    scale_env(env, 24)
    roll = rollout(env, 90, action_seed=77, action_kind="edge")
    save_rollout("rollout_edge_actions.npz", env, roll)

    # ---------------- multi-asset sleeve contract: A reference envs side by side ----------------
    A = 3
    write_case_csv("SYN_multi", 6, 40, seed=31, num_assets=A)
    envs = []
    for a in range(A):
        e, _ = make_env(f"SYN_multi_a{a}", 8, evaluate=True)
        scale_env(e, 20)
        envs.append(e)
    g = torch.Generator().manual_seed(11)
    N = 20
    acts, rews, dones, obss = [], [], [], []
    st = {k: [] for k in ("cash", "margin", "long", "short", "spot0")}
    obs0 = torch.cat([e.reset() for e in envs], dim=2)
    for t in range(70):
        a = (torch.rand((N, A), generator=g) * 2 - 1).float()
        o_l, r_l, d_l = [], [], []
        for k, e in enumerate(envs):
            # evaluate mode keeps the RNG redraw out of it; clearing the metrics before each
            # step keeps TSE:526-528 from zeroing rewards of already-terminated envs
            e.reset_evaluation_metrics()
            o, r, dn, _ = e.step(a[:, k : k + 1].contiguous())
            o_l.append(o); r_l.append(r); d_l.append(dn)
        assert all(torch.equal(d_l[0], x) for x in d_l), "sleeves must finish together in this fixture"
        r = r_l[0]
        for x in r_l[1:]:
            r = r + x  # sequential sum in asset order, f64
        acts.append(a.numpy().copy()); rews.append(r.numpy().copy()); dones.append(d_l[0].numpy().copy())
        obss.append(torch.cat(o_l, dim=2).numpy().copy())
        for k2 in st:
            st[k2].append(np.stack([state_of(e)[k2] for e in envs], axis=1))
    np.savez_compressed(
        os.path.join(GOLD, "rollout_sleeves3.npz"),
        W=np.int64(8), N=np.int64(N), A=np.int64(A), evaluate=np.int64(1),
        max_shares=np.int64(5), starting_balance=np.float64(10000), commission=np.float64(0.01),
        imr=np.float64(1.5), mmr=np.float64(0.25),
        prices=np.concatenate([e.price_environments.numpy() for e in envs], axis=2),
        logret=np.concatenate([e.log_return_environments.numpy() for e in envs], axis=2),
        actions=np.stack(acts), rewards=np.stack(rews), dones=np.stack(dones), obs=np.stack(obss),
        obs_reset=obs0.numpy(), init_env_idx=(np.arange(N) % envs[0].price_environments.shape[0]),
        **{k2: np.stack(v) for k2, v in st.items()},
    )
    print("rollout_sleeves3.npz dones=", int(np.stack(dones).sum()))

    # ---------------- rounding probes (a3) ----------------
    env, _ = make_env("SYN_roll", 8, evaluate=True)
    probes = np.asarray(
        [0.5 / 5.5, -0.5 / 5.5, 1.5 / 5.5, 2.5 / 5.5, -2.5 / 5.5, 3.5 / 5.5, 4.5 / 5.5, 5.4999 / 5.5,
         1.0, -1.0, 0.99999994, -0.2, 0.0, -0.0, 0.09090909, 0.0909091, 0.27272728, 0.45454547,
         0.6363636, 0.8181818, 1e-9, -1e-9, 0.9090909, 0.90909094, 1.5, -3.0],
        dtype=np.float32,
    )
    sc = env.get_share_changes_from_actions(torch.from_numpy(probes).unsqueeze(1)).squeeze(1).numpy()
    env9, _ = make_env("SYN_roll", 8, evaluate=True, max_shares=9)
    sc9 = env9.get_share_changes_from_actions(torch.from_numpy(probes).unsqueeze(1)).squeeze(1).numpy()
    np.savez_compressed(os.path.join(GOLD, "rounding.npz"), actions=probes, share_changes_ms5=sc, share_changes_ms9=sc9)
    print("rounding.npz", sc.tolist())

    # ---------------- f1: PPO buffer discounted returns (buffer.py:80-100) ----------------
    from finenvs.agents.PPO.buffer import Buffer

    import finenvs.agents.PPO.buffer as bufmod
    bufmod.set_device = lambda device_id: "cpu"
    T, N = 24, 10
    g = torch.Generator().manual_seed(3)
    buf = Buffer(gamma=0.99, device_id=-1)
    rew = torch.randn((T, N), generator=g, dtype=torch.float64)
    done = (torch.rand((T, N), generator=g) < 0.15).int()
    val = torch.randn((T, N, 1), generator=g)
    last = torch.randn((N, 1), generator=g)
    for t in range(T):
        buf.store(torch.zeros((N, 2, 5), dtype=torch.float64), torch.zeros((N, 1)), rew[t], done[t],
                  torch.zeros((N, 1)), val[t])
    buf.compute_returns_and_advantages(last)
    np.savez_compressed(
        os.path.join(GOLD, "ppo_returns.npz"),
        gamma=np.float64(0.99), rewards=rew.numpy(), dones=done.numpy(), values=val.squeeze(-1).numpy(),
        last_values=last.squeeze(-1).numpy(),
        returns=buf.container["returns"].squeeze(-1).numpy().T.copy(),       # (T, N)
        advantages=buf.container["advantages"].squeeze(-1).numpy().T.copy(),  # (T, N)
    )
    print("ppo_returns.npz", buf.container["returns"].dtype, buf.container["advantages"].dtype)

    # the reference tree must be untouched
    dirty = [p for p in glob.glob(os.path.join(REF, "finenvs", "data", "*", "*.json"))]
    assert not dirty, dirty
    tot = sum(os.path.getsize(f) for f in glob.glob(os.path.join(GOLD, "*.npz")))
    print(f"golden fixtures: {tot/1e6:.2f} MB")


if __name__ == "__main__":
    main()
