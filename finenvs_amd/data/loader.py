"""CSV -> market-hours series -> per-day episode bounds (host side, runs once).

Behavioural restatement of the reference's init path (TSE = finenvs/environments/
time_series_env.py), re-designed:

* file lookup by dataset key with the reference's error behaviour (TSE:47-73),
* CSV rows ``Date,Time,Open,High,Low,Close,Volume`` (TSE:80-88), Volume dropped (TSE:170),
  parsed by the native reader csrc/fe_csv.cpp (mmap, one pass; pandas only in the tests),
* market-hours filter 09:30 <= time <= 15:59 inclusive (TSE:90-91),
* per date, in order of first appearance: rows ``[first - W, last]``; dates whose
  backtracked start would be negative are skipped; L = longest episode (TSE:127-152).

Differences by design: bounds are computed in O(T) with numpy instead of the
reference's O(days x rows) mask scan (TSE:141-148), and there is NO bounds cache
file -- the reference's JSON cache is keyed by dataset key only and silently goes
stale when num_intervals changes (TSE:104-106).  The log-return transform and the
NaN-padded (D, L, 4A) tables are built on the GPU (csrc/fe_env.hip).
"""
from __future__ import annotations

import os
from glob import glob
from typing import List, Sequence, Tuple

import numpy as np

POSSIBLE_KEYS = ["dummy", "train", "valid", "test"]
OPEN_SECONDS = (9 * 60 + 30) * 60
LAST_SECONDS = (15 * 60 + 59) * 60


def get_data_dir_name(instrument_name: str) -> str:
    """TSE:47-51: a name containing "data" is a path; otherwise <package>/data/<name>
    (or $FINENVS_DATA_DIR/<name> when that variable is set)."""
    if "data" not in instrument_name:
        root = os.environ.get("FINENVS_DATA_DIR") or os.path.dirname(os.path.realpath(__file__))
        return os.path.join(root, instrument_name)
    return instrument_name


def determine_file_key(key_attempt: str) -> str:
    """TSE:53-58."""
    for possible_key in POSSIBLE_KEYS:
        if possible_key in key_attempt:
            return possible_key
    raise Exception("dataset_key expected to be one of: " + str(POSSIBLE_KEYS))


def find_file_by_key(data_dir_name: str, key: str) -> str:
    """TSE:60-73: exactly one ``*key*.csv`` in the directory, else Exception."""
    filenames = glob(os.path.join(data_dir_name, "*" + key + "*.csv"))
    if len(filenames) == 0:
        raise Exception(f"No file was found in {data_dir_name} with key ({key})")
    if len(filenames) > 1:
        raise Exception(f"More than one file was found in {data_dir_name} with key ({key})")
    return filenames[0]


def _read_native(path: str, market_hours_only: bool = True):
    """One pass over an mmap of the file in C++ (csrc/fe_csv.cpp): prices, day ids, date keys, seconds."""
    import ctypes as C

    from .. import _lib

    lib = _lib.load()
    bpath = os.fsencode(path)
    cap = lib.fe_csv_count_lines(bpath)
    if cap < 0:
        _lib.check(int(cap), lib)
    prices = np.empty((max(cap, 1), 4), dtype=np.float64)
    day_id = np.empty(max(cap, 1), dtype=np.int64)
    key = np.empty(max(cap, 1), dtype=np.int64)
    sec = np.empty(max(cap, 1), dtype=np.int64)
    rows = lib.fe_csv_read(bpath, cap, int(market_hours_only), prices.ctypes.data_as(C.c_void_p),
                           day_id.ctypes.data_as(C.c_void_p), key.ctypes.data_as(C.c_void_p),
                           sec.ctypes.data_as(C.c_void_p))
    if rows < 0:
        _lib.check(int(rows), lib)
    return prices[:rows].copy(), day_id[:rows].copy(), key[:rows].copy(), sec[:rows].copy()


def read_csv_series(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """One instrument: (prices (T,4) f64 O,H,L,C; day_id (T,) i64; second-of-day (T,) i64),
    already restricted to market hours.  day_id numbers dates by first appearance."""
    prices, day_id, _, sec = _read_native(path)
    return prices, day_id, sec


def read_csv_series_pandas(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """The same through pandas, the reference's own parser (TSE:80-91).  Kept as the
    cross-check of the native reader in tests; the product path does not use it."""
    import pandas as pd

    df = pd.read_csv(path, names=["Date", "Time", "Open", "High", "Low", "Close", "Volume"], dtype={"Date": str, "Time": str})
    t = df["Time"].str.strip().str.split(":", expand=True).fillna("0").astype(np.int64)
    sec = t[0].values * 3600 + t[1].values * 60 + (t[2].values if t.shape[1] > 2 else 0)
    keep = (sec >= OPEN_SECONDS) & (sec <= LAST_SECONDS)
    dates = df["Date"].values[keep]
    _, first_pos, inv = np.unique(dates, return_index=True, return_inverse=True)
    order = np.argsort(np.argsort(first_pos))  # rank of each unique date by first appearance
    day_id = order[inv].astype(np.int64)
    prices = np.ascontiguousarray(df[["Open", "High", "Low", "Close"]].values[keep], dtype=np.float64)
    return prices, day_id, sec[keep].astype(np.int64)


def read_csv_portfolio(paths: Sequence[str]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """A instruments on a shared calendar: rows present in every file (inner join on
    date + time, in the first file's order), columns 4a..4a+3 = asset a."""
    if len(paths) == 1:
        return read_csv_series(paths[0])
    parts = [_read_native(p) for p in paths]
    # join key: (date key < 2^45, second of day < 2^17) -- no int64 wrap; the date key is the calendar date
    # itself where the text is one (fe_csv.cpp date_join_key), so files spelling dates differently still join
    jk = [key * 131072 + sec for _, _, key, sec in parts]
    for path, k in zip(paths, jk):
        if np.unique(k).size != k.size:
            raise Exception(f"{path}: duplicate (date, time) rows -- cannot join it on a shared calendar")
    common = jk[0]
    for k in jk[1:]:
        common = common[np.isin(common, k)]
    if common.size == 0:
        raise Exception("portfolio join is empty: the files " + ", ".join(paths) + " share no (date, time) row")
    cols = []
    for (prices, _, _, _), k in zip(parts, jk):
        order = np.argsort(k, kind="stable")
        pos = order[np.searchsorted(k[order], common)]
        assert np.array_equal(k[pos], common)  # every matched row carries exactly the joined (date, time)
        cols.append(prices[pos])
    sel = np.isin(jk[0], common)
    day0, sec0 = parts[0][1][sel], parts[0][3][sel]
    _, first_pos, inv = np.unique(day0, return_index=True, return_inverse=True)
    rank = np.argsort(np.argsort(first_pos))
    return np.ascontiguousarray(np.concatenate(cols, axis=1)), rank[inv].astype(np.int64), sec0


def episode_bounds(day_id: np.ndarray, num_intervals: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """(starts, stops, max_length) per TSE:127-152, O(T).

    For each date in order of first appearance: first/last row carrying that date;
    start = first - num_intervals; dates with start < 0 are skipped.
    """
    day_id = np.asarray(day_id, dtype=np.int64)
    T = day_id.shape[0]
    if T == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), 0
    uniq, first = np.unique(day_id, return_index=True)
    last = T - 1 - np.unique(day_id[::-1], return_index=True)[1]
    order = np.argsort(first, kind="stable")
    first, last = first[order], last[order]
    start = first - int(num_intervals)
    ok = start >= 0
    starts = start[ok].astype(np.int64)
    stops = last[ok].astype(np.int64)
    max_length = int((stops - starts + 1).max()) if starts.size else 0
    return starts, stops, max_length


def padding_rows(starts: np.ndarray, stops: np.ndarray, max_length: int) -> List[int]:
    """Rows of NaN padding per day (what the reference draws torch.rand for, TSE:205-210)."""
    return [int(max_length - (b - a + 1)) for a, b in zip(starts, stops)]
