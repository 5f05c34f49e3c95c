"""CPU: throughput of the native CSV reader (csrc/fe_csv.cpp) against the reference's pandas path
(read_csv + Datetime index + between_time, TSE:80-91) on a synthetic minute-bar file."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finenvs_amd.data import loader, synthetic  # noqa: E402

days = int(sys.argv[1]) if len(sys.argv) > 1 else 2500  # ~10 years of 390-bar sessions
prices, day_id, minute = synthetic.synthetic_series(days, 1, 390, 1)
with tempfile.TemporaryDirectory() as t:
    p = os.path.join(t, "data", "BIG", "dummy.csv")
    synthetic.write_csv(p, prices, day_id, minute, 0, premarket_rows=60)
    size = os.path.getsize(p)
    t0 = time.perf_counter(); a = loader.read_csv_series(p); t1 = time.perf_counter()
    import pandas as pd
    t2 = time.perf_counter()
    df = pd.read_csv(p, names=["Date", "Time", "Open", "High", "Low", "Close", "Volume"])
    df["Datetime"] = pd.to_datetime(df["Date"] + " " + df["Time"])
    df = df.set_index("Datetime").between_time("9:30", "15:59")
    t3 = time.perf_counter()
    assert np.array_equal(a[0], df[["Open", "High", "Low", "Close"]].values)
    print(f"{size/1e6:.1f} MB, {len(df)} market-hours rows of {days * 450} lines: native {t1-t0:.3f} s = {size/1e6/(t1-t0):.0f} MB/s; "
          f"pandas read_csv + to_datetime + between_time (TSE:80-91) {t3-t2:.3f} s = {size/1e6/(t3-t2):.0f} MB/s; x{(t3-t2)/(t1-t0):.1f}")
