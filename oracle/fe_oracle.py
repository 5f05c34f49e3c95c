"""ctypes/numpy front-end of the C parity oracle (oracle/fe_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by finenvs_amd.  It mirrors the state
layout of the reference's TimeSeriesEnv (TSE:245-269) in numpy arrays so that
a test can step the oracle and the HIP env side by side on the same inputs.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import Dict, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfe_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fe_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libfe_oracle.so"])
    return _LIB_PATH


class FoConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int64), ("D", C.c_int64), ("L", C.c_int64),
        ("W", C.c_int32), ("A", C.c_int32),
        ("max_shares", C.c_int32), ("evaluate", C.c_int32),
        ("starting_balance", C.c_double), ("commission", C.c_double),
        ("init_margin", C.c_double), ("maint_margin", C.c_double),
        ("obs_is_f32", C.c_int32), ("redraw_mode", C.c_int32),
        ("seed", C.c_uint64), ("eval_env", C.c_int64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        # FE_ORACLE_LIB: an instrumented build of the same source (tests/soak/sanitize_cpu.sh)
        _lib = C.CDLL(os.environ.get("FE_ORACLE_LIB") or build())
        _lib.fo_redraw_day.restype = C.c_int64
        _lib.fo_redraw_day.argtypes = [C.c_uint64, C.c_uint64, C.c_int64]
        _lib.fo_philox_u32.restype = C.c_uint32
        _lib.fo_philox_u32.argtypes = [C.c_uint64, C.c_uint64]
        _lib.fo_bounds.restype = C.c_int64
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def build_logret(prices: np.ndarray) -> np.ndarray:
    prices = np.ascontiguousarray(prices, dtype=np.float64)
    T, c4 = prices.shape
    out = np.empty_like(prices)
    lib().fo_build_logret(_p(prices), _p(out), C.c_int64(T), C.c_int32(c4 // 4))
    return out


def bounds(day_id: np.ndarray, W: int) -> Tuple[np.ndarray, np.ndarray, int]:
    day_id = np.ascontiguousarray(day_id, dtype=np.int64)
    T = day_id.shape[0]
    starts = np.empty(T, dtype=np.int64)
    stops = np.empty(T, dtype=np.int64)
    maxlen = C.c_int64(0)
    D = lib().fo_bounds(_p(day_id), C.c_int64(T), C.c_int32(W), _p(starts), _p(stops), C.byref(maxlen))
    return starts[:D].copy(), stops[:D].copy(), int(maxlen.value)


def build_tables(series: np.ndarray, starts: np.ndarray, stops: np.ndarray, L: int) -> np.ndarray:
    series = np.ascontiguousarray(series, dtype=np.float64)
    T, c4 = series.shape
    D = len(starts)
    out = np.empty((D, L, c4), dtype=np.float64)
    lib().fo_build_tables(_p(series), C.c_int64(T), C.c_int32(c4 // 4), _p(np.ascontiguousarray(starts, dtype=np.int64)),
                          _p(np.ascontiguousarray(stops, dtype=np.int64)), C.c_int64(D), C.c_int64(L), _p(out))
    return out


def tables_from_series(prices: np.ndarray, day_id: np.ndarray, W: int):
    """Market-hours-filtered series -> (price tables, log-return tables, starts, stops, L)."""
    starts, stops, L = bounds(day_id, W)
    lr = build_logret(prices)
    return build_tables(prices, starts, stops, L), build_tables(lr, starts, stops, L), starts, stops, L


def discounted_returns(rew: np.ndarray, done: np.ndarray, last_values: np.ndarray, gamma: float) -> np.ndarray:
    rew = np.ascontiguousarray(rew, dtype=np.float64)
    done = np.ascontiguousarray(done, dtype=np.int32)
    last_values = np.ascontiguousarray(last_values, dtype=np.float32)
    T, N = rew.shape
    out = np.empty((T, N), dtype=np.float32)
    lib().fo_discounted_returns(_p(rew), _p(done), _p(last_values), C.c_int64(T), C.c_int64(N), C.c_double(gamma), _p(out))
    return out


def f32_iadd_f64(base: np.ndarray, delta: np.ndarray, subtract: bool = False) -> np.ndarray:
    base = np.ascontiguousarray(base, dtype=np.float32)
    delta = np.ascontiguousarray(delta, dtype=np.float64)
    out = np.empty_like(base)
    lib().fo_f32_iadd_f64(_p(base), _p(delta), C.c_int64(base.shape[0]), C.c_int32(int(subtract)), _p(out))
    return out


class EpisodeStatsOracle:
    """PPO_agent.py:120-132 / 146-163 on numpy state (per-env count / sum / sum of squares instead of a list, added up
    in the fixed order of fe_env_stats_reduce: the build's contract, restated by fo_stats_reduce)."""

    def __init__(self, num_envs: int, eval_env: int):
        self.N, self.eval_env = int(num_envs), int(eval_env)
        self.running = np.zeros(self.N, dtype=np.float32)
        self.acc = np.zeros((self.N, 3), dtype=np.float64)
        self.eval = np.zeros(2, dtype=np.float32)

    def step(self, rewards: np.ndarray, dones: np.ndarray) -> None:
        rewards = np.ascontiguousarray(rewards, dtype=np.float64)
        dones = np.ascontiguousarray(dones, dtype=np.int32)
        lib().fo_episode_stats_step(C.c_int64(self.N), C.c_int64(self.eval_env), _p(rewards), _p(dones),
                                    _p(self.running), _p(self.acc), _p(self.eval))

    def read(self, reset: bool = True):
        sums = np.zeros(3, dtype=np.float64)
        lib().fo_stats_reduce(_p(self.acc), C.c_int64(self.N), _p(sums))
        n, s, ss = (float(x) for x in sums)
        mean = s / n if n > 0 else float("nan")
        var = (ss - n * mean * mean) / (n - 1) if n > 1 else float("nan")
        out = {"num_training_episodes": int(n), "mean_training_return": mean,
               "std_dev_training_return": math.sqrt(max(var, 0.0)) if n > 1 else float("nan"),  # (sqrt is correctly rounded; x ** 0.5 is not)
               "evaluation_return": float(self.eval[0]) if self.eval[1] > 0 else None}
        if reset:
            self.acc[:] = 0
            self.eval[:] = 0
        return out


def policy_linear(obs: np.ndarray, weights: np.ndarray, bias: float) -> np.ndarray:
    """Actions (N, A) f32 of the in-kernel linear policy on a materialised observation (N, W, 5A)."""
    obs = np.ascontiguousarray(obs, dtype=np.float64)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    N, W, c5 = obs.shape
    A = c5 // 5
    out = np.empty((N, A), dtype=np.float32)
    lib().fo_policy_linear(_p(obs), _p(weights), C.c_double(bias), C.c_int64(N), C.c_int32(W), C.c_int32(A), _p(out))
    return out


def mlp_pack(W1: np.ndarray, W: int):
    """(5W, H) first-layer weights (rows 5j+c, the flattened observation order) -> (w1t (H, 4W), wpos (H,)) as the
    kernel takes them: log-return rows transposed, position rows summed over j in f32, j ascending."""
    W1 = np.asarray(W1, dtype=np.float32).reshape(W, 5, -1)
    w1t = np.ascontiguousarray(W1[:, :4, :].reshape(4 * W, -1).T)
    wpos = np.zeros(W1.shape[2], dtype=np.float32)
    for j in range(W):
        wpos = (wpos + W1[j, 4, :]).astype(np.float32)
    return w1t, wpos


def policy_mlp(obs: np.ndarray, w1t: np.ndarray, wpos: np.ndarray, b1: np.ndarray, w2: np.ndarray, b2: float,
               act: int = 0, return_pre: bool = False):
    """Actions (N, A) f32 of the in-kernel MLP policy on a materialised observation (N, W, 5A)."""
    obs = np.ascontiguousarray(obs, dtype=np.float64)
    N, W, c5 = obs.shape
    A = c5 // 5
    w1t = np.ascontiguousarray(w1t, dtype=np.float32)
    H = w1t.shape[0]
    assert w1t.shape == (H, 4 * W)
    wpos, b1, w2 = (np.ascontiguousarray(x, dtype=np.float32) for x in (wpos, b1, w2))
    out = np.empty((N, A), dtype=np.float32)
    pre = np.empty((N, A, H), dtype=np.float32) if return_pre else None
    lib().fo_policy_mlp(_p(obs), _p(w1t), _p(wpos), _p(b1), _p(w2), C.c_float(b2), C.c_int32(H), C.c_int32(act),
                        C.c_int64(N), C.c_int32(W), C.c_int32(A), _p(out), _p(pre))
    return (out, pre) if return_pre else out


def lstm_row_order(H: int) -> np.ndarray:
    """torch gate-row index (gate * H + unit) of every packed row R = 32*mt + 8*b + 4*half + gate, whose hidden
    unit is 8*mt + 4*half + b (the order the kernel's accumulator registers hold the gates in)."""
    R = np.arange(4 * H)
    mt, rho = R // 32, R % 32
    gate, half, b = rho % 4, (rho % 8) // 4, rho // 8
    return gate * H + 8 * mt + 4 * half + b


def lstm_pack(W_ih: np.ndarray, W_hh: np.ndarray, b_ih: np.ndarray, b_hh: np.ndarray):
    """nn.LSTM(5, H) parameters (weight_ih_l0 (4H, 5), weight_hh_l0 (4H, H), bias_ih_l0, bias_hh_l0; gate order
    i, f, g, o) -> (whh (4H, H), wx (4H, 8)) with rows in packed order and wx = [w_ih[0..3], w_ih[4], b_ih + b_hh
    (one f32 add), 0, 0]."""
    W_ih, W_hh = np.asarray(W_ih, dtype=np.float32), np.asarray(W_hh, dtype=np.float32)
    H = W_hh.shape[1]
    assert W_ih.shape == (4 * H, 5) and W_hh.shape == (4 * H, H)
    order = lstm_row_order(H)
    bias = (np.asarray(b_ih, dtype=np.float32) + np.asarray(b_hh, dtype=np.float32)).astype(np.float32)
    wx = np.zeros((4 * H, 8), dtype=np.float32)
    wx[:, :5] = W_ih[order]
    wx[:, 5] = bias[order]
    return np.ascontiguousarray(W_hh[order]), wx


def policy_lstm(obs: np.ndarray, whh: np.ndarray, wx: np.ndarray, wout: np.ndarray, bout: float, out_act: int = 0,
                return_h: bool = False):
    """Outputs (N, A) f32 of the in-kernel LSTM head on a materialised observation (N, W, 5A); out_act 0 tanh, 1 clamp to
    [-1, 1], 2 none (a critic's value)."""
    obs = np.ascontiguousarray(obs, dtype=np.float64)
    N, W, c5 = obs.shape
    A = c5 // 5
    whh, wx, wout = (np.ascontiguousarray(x, dtype=np.float32) for x in (whh, wx, wout))
    H = whh.shape[1]
    assert whh.shape == (4 * H, H) and wx.shape == (4 * H, 8) and wout.shape == (H,)
    out = np.empty((N, A), dtype=np.float32)
    h = np.empty((N, A, H), dtype=np.float32) if return_h else None
    lib().fo_policy_lstm(_p(obs), _p(whh), _p(wx), _p(wout), C.c_float(bout), C.c_int32(H), C.c_int32(out_act),
                         C.c_int64(N), C.c_int32(W), C.c_int32(A), _p(out), _p(h))
    return (out, h) if return_h else out


def lstm_activations(x: np.ndarray):
    """(sigmoid, tanh) of the LSTM head's exact-operation activation forms, elementwise on f32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    sig, tnh = np.empty_like(x), np.empty_like(x)
    lib().fo_lstm_activations(_p(x), _p(sig), _p(tnh), C.c_int64(x.size))
    return sig, tnh


def policy_table(LR: np.ndarray, weights: np.ndarray, W: int):
    """(table (D, L, A), wsum) of the table-form linear policy."""
    LR = np.ascontiguousarray(LR, dtype=np.float64)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    D, L, c4 = LR.shape
    A = c4 // 4
    table = np.empty((D, L, A), dtype=np.float64)
    wsum = np.empty(1, dtype=np.float64)
    lib().fo_policy_table(_p(LR), _p(weights), C.c_int64(D), C.c_int64(L), C.c_int32(W), C.c_int32(A), _p(table), _p(wsum))
    return table, float(wsum[0])


def policy_table_actions(table: np.ndarray, wsum: float, bias: float, obs_row: np.ndarray, obs_pos: np.ndarray) -> np.ndarray:
    obs_row = np.ascontiguousarray(obs_row, dtype=np.int64)
    obs_pos = np.ascontiguousarray(obs_pos, dtype=np.float64)
    N, A = obs_pos.shape
    out = np.empty((N, A), dtype=np.float32)
    lib().fo_policy_table_actions(_p(np.ascontiguousarray(table)), C.c_double(wsum), C.c_double(bias), _p(obs_row), _p(obs_pos),
                                  C.c_int64(N), C.c_int32(A), _p(out))
    return out


class OracleEnv:
    """numpy-state mirror of TimeSeriesEnv driven by fo_step / fo_reset_obs."""

    def __init__(
        self,
        prices: np.ndarray,
        logret: np.ndarray,
        num_intervals: int,
        num_envs: Optional[int] = None,
        max_shares: int = 5,
        starting_balance: float = 10000,
        per_share_commission: float = 0.01,
        initial_margin_requirement: float = 1.5,
        maintenance_margin_requirement: float = 0.25,
        evaluate: bool = False,
        env_indices: Optional[np.ndarray] = None,
        obs_f32: bool = False,
        redraw_mode: int = 0,
        seed: int = 0,
        eval_env: Optional[int] = None,
        nthreads: int = 1,
        auto_emit: bool = True,
    ):
        self.auto_emit = auto_emit  # False: leave the evaluate-mode metrics alone (fused K-step launches)
        self.P = np.ascontiguousarray(prices, dtype=np.float64)
        self.LR = np.ascontiguousarray(logret, dtype=np.float64)
        D, L, c4 = self.P.shape
        A = c4 // 4
        if env_indices is None:
            N = num_envs if num_envs is not None else D
            env_indices = np.arange(N, dtype=np.int64) % D
        self.env_idx = np.ascontiguousarray(env_indices, dtype=np.int64).copy()
        N = self.env_idx.shape[0]
        if eval_env is None:
            eval_env = -1 if evaluate else N - 1
        self.cfg = FoConfig(N, D, L, num_intervals, A, max_shares, int(evaluate), float(starting_balance),
                            float(per_share_commission), float(initial_margin_requirement),
                            float(maintenance_margin_requirement), int(obs_f32), redraw_mode, seed, eval_env)
        self.N, self.D, self.L, self.W, self.A = N, D, L, num_intervals, A
        self.nthreads = nthreads
        self.spot0 = np.zeros(N, dtype=np.int64)
        self.cash = np.full((N, A), np.float32(starting_balance), dtype=np.float32)
        self.long = np.zeros((N, A), dtype=np.float32)
        self.short = np.zeros((N, A), dtype=np.float32)
        self.margin = np.zeros((N, A), dtype=np.float64)
        self.terminated = np.zeros(N, dtype=np.uint8)
        self.episode_returns = np.zeros(N, dtype=np.float32)
        self.n_terminated = np.zeros(1, dtype=np.int64)
        self.redraw_counter = np.zeros(1, dtype=np.uint64)
        self.obs_dtype = np.float32 if obs_f32 else np.float64
        self.obs = np.empty((N, self.W, 5 * A), dtype=self.obs_dtype)
        self.rew = np.empty(N, dtype=np.float64)
        self.done = np.empty(N, dtype=np.int32)

    def reset(self) -> np.ndarray:
        lib().fo_reset_obs(C.byref(self.cfg), _p(self.P), _p(self.LR), _p(self.env_idx), _p(self.spot0),
                           _p(self.long), _p(self.short), _p(self.obs))
        return self.obs

    def step(self, actions: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray, Dict]:
        """float64 actions take the reference's dtype promotion (TSE:298-302, 353-374): the share tensors are f64 from that
        step on (``self.shares_f64``, sticky), which changes the precision of the commission products -- see fo_step_ex."""
        actions = np.asarray(actions)
        act_f64 = actions.dtype == np.float64
        if act_f64:
            self.shares_f64 = True
        a = np.ascontiguousarray(actions, dtype=np.float64 if act_f64 else np.float32).reshape(self.N, self.A)
        rc = lib().fo_step_ex(C.byref(self.cfg), _p(self.P), _p(self.LR), _p(self.env_idx), _p(self.spot0),
                              _p(self.cash), _p(self.long), _p(self.short), _p(self.margin), _p(self.terminated),
                              _p(self.episode_returns), _p(self.n_terminated), _p(self.redraw_counter), _p(a),
                              C.c_int(int(act_f64)), C.c_int(int(getattr(self, "shares_f64", False))),
                              _p(self.obs), _p(self.rew), _p(self.done), C.c_int(self.nthreads))
        if rc != 0:
            raise RuntimeError(f"fo_step failed: {rc}")
        info: Dict = {}
        if self.auto_emit and self.cfg.evaluate and int(self.n_terminated[0]) == self.N:
            info = {"returns": self.episode_returns.copy()}
            self.terminated[:] = 0
            self.episode_returns[:] = 0
            self.n_terminated[0] = 0
        return self.obs, self.rew, self.done, info

    def set_day(self, env: int, day: int) -> None:
        self.env_idx[env] = day
