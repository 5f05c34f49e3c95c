"""CPU tests of the native CSV reader (csrc/fe_csv.cpp) against pandas, the parser the reference
uses (TSE:80-91), and against the reference-generated table fixtures."""
import os

import numpy as np
import pytest

from finenvs_amd.data import loader, synthetic
from tests.helpers import assert_bits, load_golden


def _write(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("\n".join(rows) + "\n")


def test_native_equals_pandas_on_generated_files(tmp_path):
    for seed, drop in ((1, 0.0), (2, 0.2)):
        prices, day_id, minute = synthetic.synthetic_series(6, 1, 120, seed, drop)
        p = str(tmp_path / f"data{seed}" / "dummy.csv")
        synthetic.write_csv(p, prices, day_id, minute, 0, premarket_rows=3)
        a = loader.read_csv_series(p)
        b = loader.read_csv_series_pandas(p)
        for x, y, what in zip(a, b, ("prices", "day_id", "sec")):
            assert_bits(x, y, what)


def test_number_parsing_matches_pandas_bit_for_bit(tmp_path):
    """Decimal strings of the kind price files hold (<= 15 significant digits): from_chars ==
    pandas' default ('high') converter."""
    rng = np.random.default_rng(0)
    rows = []
    for i in range(20000):
        dec = int(rng.integers(0, 9))
        vals = [f"{rng.uniform(0.001, 99999):.{dec}f}" for _ in range(4)]
        if i % 97 == 0:
            vals[0] = ["1e3", "2.5E-3", "+7.25", "000123.4500", ".5", "5.", "123456789012345", "0.000001"][i // 97 % 8]
        rows.append(f"2020-01-0{1 + i % 9},10:{i % 60:02d}:00," + ",".join(vals) + ",100")
    p = str(tmp_path / "data" / "dummy.csv")
    _write(p, rows)
    a = loader.read_csv_series(p)
    b = loader.read_csv_series_pandas(p)
    assert_bits(a[0], b[0], "prices")
    assert_bits(a[1], b[1], "day ids")


def test_formats_filter_and_order_of_first_appearance(tmp_path):
    rows = [
        "01/05/1998,09:29,1,1,1,1,5",        # before the open: dropped
        "01/05/1998,09:30,2,2,2,2,5",
        "01/02/1998,15:59,3,3,3,3,5",        # an earlier date appearing later: gets the next id
        "01/02/1998,16:00,4,4,4,4,5",        # after the last kept bar: dropped
        "01/05/1998,15:59:00,5,5,5,5,5",
        "01/05/1998,15:59:30,6,6,6,6,5",     # 15:59:30 > 15:59:00: dropped, as between_time does
        "",
        "01/06/1998,12:00,7.5,8,7,7.25,5\r",  # CRLF
    ]
    p = str(tmp_path / "data" / "dummy.csv")
    _write(p, rows)
    prices, day, sec = loader.read_csv_series(p)
    assert prices[:, 0].tolist() == [2, 3, 5, 7.5]
    assert day.tolist() == [0, 1, 0, 2]
    assert sec.tolist() == [34200, 57540, 57540, 43200]
    b = loader.read_csv_series_pandas(p)
    assert_bits(prices, b[0]); assert_bits(day, b[1]); assert_bits(sec, b[2])


def test_errors_are_loud(tmp_path):
    from finenvs_amd._lib import FinEnvsNativeError

    with pytest.raises(FinEnvsNativeError, match="cannot open"):
        loader.read_csv_series(str(tmp_path / "nope.csv"))
    p = str(tmp_path / "data" / "dummy.csv")
    _write(p, ["2020-01-02,10:00,1,2,x,4,5"])
    with pytest.raises(FinEnvsNativeError, match="bad number"):
        loader.read_csv_series(p)
    _write(p, ["2020-01-02,10:00,1,2"])
    with pytest.raises(FinEnvsNativeError, match="fields"):
        loader.read_csv_series(p)
    _write(p, ["2020-01-02,1000,1,2,3,4,5"])
    with pytest.raises(FinEnvsNativeError, match="bad time"):
        loader.read_csv_series(p)
    open(p, "w").close()
    prices, day, sec = loader.read_csv_series(p)
    assert prices.shape == (0, 4)


def test_native_reader_reproduces_reference_frames(tmp_path):
    g = load_golden("tables_ragged.npz")
    prices, day_id, minute = synthetic.synthetic_series(7, 1, 40, 77, 0.10)
    p = str(tmp_path / "data" / "dummy.csv")
    synthetic.write_csv(p, prices, day_id, minute, 0, premarket_rows=2)
    got, d, _ = loader.read_csv_series(p)
    assert_bits(got, g["ref_dataset"], "the frame the reference built from the same CSV")
    s, e, L = loader.episode_bounds(d, int(g["W"]))
    assert_bits(s, g["ref_start_indices"]); assert_bits(e, g["ref_stop_indices"])
