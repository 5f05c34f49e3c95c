"""Deterministic synthetic minute-bar series (the build's own generator).

The reference ships only three small CSV fixtures (finenvs/data/README.md:7-9
describes the row format ``Date,Time,Open,High,Low,Close,Volume``); BASELINE's
64k..4M env configurations need a series generator of our own.  The recipe is
the one fixed in SURVEY.md section 8(d):

* business days from 2020-01-02, ``bars_per_day`` one-minute bars from 09:30,
* per asset ``a``: geometric random walk, ``p0 = 100*(1 + a/10)``, per-bar
  sigma 5e-4, ``O_t = C_{t-1}*exp(s*z1)``, ``C_t = O_t*exp(s*z2)``,
  ``H = max(O,C)*exp(|s*z3|/2)``, ``L = min(O,C)*exp(-|s*z4|/2)``,
* prices rounded to 4 decimals as in the reference CSVs,
* ``numpy.random.default_rng(seed + a)`` with seed 1234.

Nothing here is on the hot path: it runs once, on the host.
"""
from __future__ import annotations

import datetime as _dt
import os
from typing import List, Optional, Tuple

import numpy as np

MARKET_OPEN_MINUTE = 9 * 60 + 30  # 09:30, first bar kept by the market-hours filter
MARKET_LAST_MINUTE = 15 * 60 + 59  # 15:59, last bar kept (TSE:90-91)


def business_days(num_days: int, start: _dt.date = _dt.date(2020, 1, 2)) -> List[_dt.date]:
    days: List[_dt.date] = []
    d = start
    while len(days) < num_days:
        if d.weekday() < 5:
            days.append(d)
        d += _dt.timedelta(days=1)
    return days


def gbm_ohlc(
    num_bars: int, asset: int = 0, seed: int = 1234, sigma: float = 5e-4
) -> np.ndarray:
    """(num_bars, 4) float64 O,H,L,C for one asset, rounded to 4 decimals."""
    rng = np.random.default_rng(seed + asset)
    z = rng.standard_normal((num_bars, 4))
    p0 = 100.0 * (1.0 + asset / 10.0)
    # log-price walk: open gap then intrabar move, cumulatively
    steps = sigma * z[:, :2]
    log_close = np.log(p0) + np.cumsum(steps.sum(axis=1))
    log_open = log_close - steps[:, 1]
    o = np.exp(log_open)
    c = np.exp(log_close)
    h = np.maximum(o, c) * np.exp(np.abs(sigma * z[:, 2]) / 2.0)
    l = np.minimum(o, c) * np.exp(-np.abs(sigma * z[:, 3]) / 2.0)
    out = np.stack([o, h, l, c], axis=1)
    return np.round(out, 4)


def synthetic_series(
    num_days: int,
    num_assets: int = 1,
    bars_per_day: int = 390,
    seed: int = 1234,
    drop_prob: float = 0.0,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """A market-hours-only series on a shared calendar.

    Returns ``(prices (T, 4*A) f64, day_id (T,) i64, minute (T,) i64)`` where
    ``minute`` is minutes since midnight.  With ``drop_prob > 0`` each bar is
    dropped independently (the "no transactions in this interval" case of the
    reference data format), giving ragged days.
    """
    total = num_days * bars_per_day
    cols = [gbm_ohlc(total, a, seed) for a in range(num_assets)]
    prices = np.concatenate(cols, axis=1)
    day_id = np.repeat(np.arange(num_days, dtype=np.int64), bars_per_day)
    minute = np.tile(MARKET_OPEN_MINUTE + np.arange(bars_per_day, dtype=np.int64), num_days)
    if drop_prob > 0.0:
        keep = np.random.default_rng(seed + 7919).random(total) >= drop_prob
        prices, day_id, minute = prices[keep], day_id[keep], minute[keep]
    return np.ascontiguousarray(prices), day_id, minute


def write_csv(
    path: str,
    prices: np.ndarray,
    day_id: np.ndarray,
    minute: np.ndarray,
    asset: int = 0,
    premarket_rows: int = 0,
) -> None:
    """Write one asset of a series in the reference's CSV row format.

    ``premarket_rows`` extra 04:00.. rows per day are emitted ahead of the
    session so the market-hours filter has something to remove.
    """
    days = business_days(int(day_id.max()) + 1)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    p = prices[:, 4 * asset : 4 * asset + 4]
    with open(path, "w", encoding="utf-8") as f:
        prev_day = -1
        for i in range(p.shape[0]):
            d = int(day_id[i])
            if d != prev_day and premarket_rows:
                for k in range(premarket_rows):
                    mm = 4 * 60 + k
                    f.write(
                        "%s,%02d:%02d:00,%.4f,%.4f,%.4f,%.4f,%d\n"
                        % (days[d].isoformat(), mm // 60, mm % 60, p[i, 0], p[i, 0], p[i, 0], p[i, 0], 100)
                    )
            prev_day = d
            m = int(minute[i])
            f.write(
                "%s,%02d:%02d:00,%.4f,%.4f,%.4f,%.4f,%d\n"
                % (days[d].isoformat(), m // 60, m % 60, p[i, 0], p[i, 1], p[i, 2], p[i, 3], 1000 + i % 977)
            )
