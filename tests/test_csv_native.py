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


def test_multi_piece_parse_equals_pandas_and_reports_the_first_error_in_file_order(tmp_path):
    """Files beyond 8 MiB are cut at line boundaries and parsed by several threads (fe_csv.cpp); the stitched result, the
    day numbering across the cuts, the absolute line number of the FIRST malformed line and the capacity check must be
    what a single pass gives."""
    import ctypes as C

    from finenvs_amd import _lib
    from finenvs_amd._lib import FinEnvsNativeError

    prices, day_id, minute = synthetic.synthetic_series(700, 1, 390, 3)   # ~18.7 MB with the pre-market rows: 4 pieces
    p = str(tmp_path / "data" / "BIG" / "dummy.csv")
    synthetic.write_csv(p, prices, day_id, minute, 0, premarket_rows=60)
    assert os.path.getsize(p) > 16 * (1 << 20)
    a = loader.read_csv_series(p)
    b = loader.read_csv_series_pandas(p)
    assert a[0].shape[0] == 700 * 390
    assert_bits(a[0], b[0]); assert_bits(a[1], b[1]); assert_bits(a[2], b[2])
    assert a[1].tolist() == np.repeat(np.arange(700), 390).tolist()  # day ids run on across the cuts

    lines = open(p).read().split("\n")
    n_lines = len(lines) - 1
    lib = _lib.load()
    # two malformed lines, one in the last piece and one in the third: the earlier one is reported, with its absolute number
    bad_late, bad_early = n_lines - 50, int(n_lines * 0.6)
    broken = list(lines)
    broken[bad_late] = "2020-01-02,10:00,1,2"
    while not ("09:30" <= broken[bad_early].split(",")[1][:5] <= "15:59"):  # (a pre-market row is dropped before its numbers are read)
        bad_early += 1
    f = broken[bad_early].split(",")
    f[2] = "x" + f[2]  # the Open field
    broken[bad_early] = ",".join(f)
    q = str(tmp_path / "data" / "BIG" / "broken.csv")
    open(q, "w").write("\n".join(broken))
    with pytest.raises(FinEnvsNativeError, match=rf"line {bad_early + 1}: bad number in column 3"):
        loader.read_csv_series(q)
    broken[bad_early] = lines[bad_early]
    open(q, "w").write("\n".join(broken))
    with pytest.raises(FinEnvsNativeError, match=rf"line {bad_late + 1} has 4 fields"):
        loader.read_csv_series(q)
    # capacity: exactly enough is fine, one row less is refused (whichever piece the overflowing row falls into)
    rows = 700 * 390
    for cap, ok in ((rows, True), (rows - 1, False), (rows // 2, False)):
        pr = np.empty((max(cap, 1), 4)); d = np.empty(max(cap, 1), np.int64); k = np.empty(max(cap, 1), np.int64); sc = np.empty(max(cap, 1), np.int64)
        rc = lib.fe_csv_read(os.fsencode(p), cap, 1, pr.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                             k.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p))
        assert (rc == rows) if ok else (rc < 0 and b"capacity" in lib.fe_last_error())


def test_native_reader_reproduces_reference_frames(tmp_path):
    g = load_golden("tables_ragged.npz")
    prices, day_id, minute = synthetic.synthetic_series(7, 1, 40, 77, 0.10)
    p = str(tmp_path / "data" / "dummy.csv")
    synthetic.write_csv(p, prices, day_id, minute, 0, premarket_rows=2)
    got, d, _ = loader.read_csv_series(p)
    assert_bits(got, g["ref_dataset"], "the frame the reference built from the same CSV")
    s, e, L = loader.episode_bounds(d, int(g["W"]))
    assert_bits(s, g["ref_start_indices"]); assert_bits(e, g["ref_stop_indices"])


REF_DATA = "/root/reference/finenvs/data"


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("inst,fixture", [("OIH", "tables_oih.npz"), ("IBM", "tables_ibm.npz"), ("SPY", "tables_spy.npz")])
def test_native_reader_on_the_references_own_files(inst, fixture):
    """fe_csv_read on the three real files the reference's unit test builds (tests/unit/test_time_series_env.py:
    10-14) -- MM/DD/YYYY + HH:MM (IBM, OIH), YYYY-MM-DD + HH:MM:SS with 04:00 pre-market rows (SPY) -- equals, bit
    for bit, the frame the reference's pandas path kept (stored in the fixture by oracle/make_goldens.py), and the
    episode bounds at the fixture's window equal the reference's bounds cache."""
    g = load_golden(fixture)
    prices, day_id, sec = loader.read_csv_series(os.path.join(REF_DATA, inst, "dummy.csv"))
    assert_bits(prices, g["ref_dataset"], "market-hours OHLC rows")
    assert_bits(day_id, g["series_day_id"], "day ids by first appearance")
    assert_bits(sec, g["series_second"], "second of day")
    starts, stops, L = loader.episode_bounds(day_id, int(g["W"]))
    assert_bits(starts, g["ref_start_indices"]); assert_bits(stops, g["ref_stop_indices"])
    assert L == int(g["ref_max_length"])


def test_portfolio_join_across_date_spellings_and_its_errors(tmp_path):
    a = ["2022-04-01,09:30:00,1,1,1,1,5", "2022-04-01,09:31:00,2,2,2,2,5", "2022-04-04,09:30:00,3,3,3,3,5"]
    b = ["04/01/2022,09:30,10,10,10,10,5", "04/01/2022,09:32,20,20,20,20,5", "04/04/2022,09:30,30,30,30,30,5"]
    pa, pb = str(tmp_path / "data" / "A" / "dummy.csv"), str(tmp_path / "data" / "B" / "dummy.csv")
    _write(pa, a); _write(pb, b)
    prices, day, sec = loader.read_csv_portfolio([pa, pb])
    assert prices[:, 0].tolist() == [1, 3] and prices[:, 4].tolist() == [10, 30]   # rows present in both files
    assert day.tolist() == [0, 1] and sec.tolist() == [34200, 34200]
    _write(pb, ["05/01/2022,09:30,10,10,10,10,5"])
    with pytest.raises(Exception, match="join is empty"):
        loader.read_csv_portfolio([pa, pb])
    _write(pb, b + [b[0]])
    with pytest.raises(Exception, match="duplicate"):
        loader.read_csv_portfolio([pa, pb])
    # opaque (non-calendar) date texts still work as keys: equal text joins
    _write(pa, ["dayA,09:30,1,1,1,1,5", "dayB,09:30,2,2,2,2,5"]); _write(pb, ["dayB,09:30,7,7,7,7,5"])
    prices, day, _ = loader.read_csv_portfolio([pa, pb])
    assert prices.tolist() == [[2, 2, 2, 2, 7, 7, 7, 7]]
