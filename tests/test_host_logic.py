"""CPU tests of the host-side logic: CSV loader and bounds vs the reference-generated
fixtures, shard arithmetic, the host mirror of the device RNG, the synthetic generator."""
import ctypes as C
import os

import numpy as np
import pytest

from finenvs_amd.data import loader, synthetic
from finenvs_amd.environments.time_series_env import shard_range
from finenvs_amd.rng import philox_u32, redraw_day
from oracle import fe_oracle as fo
from tests.helpers import assert_bits, load_golden

CSV_CASES = [  # fixture, days, bars, seed, drop  (must mirror oracle/make_goldens.py)
    ("tables_full.npz", 5, 40, 1234, 0.0),
    ("tables_ragged.npz", 7, 40, 77, 0.10),
    ("tables_skip2.npz", 6, 40, 5, 0.0),
]


@pytest.mark.parametrize("name,days,bars,seed,drop", CSV_CASES)
def test_csv_loader_reproduces_reference_frames(tmp_path, name, days, bars, seed, drop):
    g = load_golden(name)
    prices, day_id, minute = synthetic.synthetic_series(days, 1, bars, seed, drop)
    assert_bits(prices, g["series_prices"], "generator is deterministic")
    d = tmp_path / "data" / "SYN"
    synthetic.write_csv(str(d / "dummy.csv"), prices, day_id, minute, 0, premarket_rows=2)
    path = loader.find_file_by_key(loader.get_data_dir_name(str(d)), loader.determine_file_key("dummy"))
    p2, d2, sec = loader.read_csv_series(path)
    assert_bits(p2, g["ref_dataset"], "market-hours rows (premarket rows filtered out)")
    assert sec.min() >= 9 * 3600 + 30 * 60 and sec.max() <= 15 * 3600 + 59 * 60
    starts, stops, L = loader.episode_bounds(d2, int(g["W"]))
    assert_bits(starts, g["ref_start_indices"], "starts")
    assert_bits(stops, g["ref_stop_indices"], "stops")
    assert L == int(g["ref_max_length"])


def test_loader_errors_match_reference_behaviour(tmp_path):
    with pytest.raises(Exception, match="dataset_key expected"):
        loader.determine_file_key("nope")
    assert loader.determine_file_key("cross_validation") == "valid"
    d = tmp_path / "data" / "X"
    d.mkdir(parents=True)
    with pytest.raises(Exception, match="No file was found"):
        loader.find_file_by_key(str(d), "train")
    (d / "a_train.csv").write_text("")
    (d / "b_train.csv").write_text("")
    with pytest.raises(Exception, match="More than one file"):
        loader.find_file_by_key(str(d), "train")
    assert "data" in loader.get_data_dir_name("SPY") and loader.get_data_dir_name("SPY").endswith("SPY")
    assert loader.get_data_dir_name("/x/data/SPY") == "/x/data/SPY"


def test_portfolio_loader_inner_joins_calendars(tmp_path):
    prices, day_id, minute = synthetic.synthetic_series(3, 2, 30, 11)
    a = tmp_path / "data" / "A" / "dummy.csv"
    b = tmp_path / "data" / "B" / "dummy.csv"
    synthetic.write_csv(str(a), prices, day_id, minute, 0)
    keep = np.ones(len(day_id), bool)
    keep[[5, 40, 41]] = False  # asset B misses three bars
    synthetic.write_csv(str(b), prices[keep], day_id[keep], minute[keep], 1)
    p, d, s = loader.read_csv_portfolio([str(a), str(b)])
    assert p.shape == (keep.sum(), 8)
    assert_bits(p, prices[keep], "joined series")
    assert_bits(d, day_id[keep], "day ids")


def test_bounds_handle_unsorted_and_empty():
    s, e, L = loader.episode_bounds(np.zeros(0, np.int64), 4)
    assert len(s) == 0 and L == 0
    day = np.array([3, 3, 3, 1, 1, 1, 1, 2, 2])
    s, e, L = loader.episode_bounds(day, 2)
    assert s.tolist() == [1, 5] and e.tolist() == [6, 8] and L == 6
    s2, e2, L2 = fo.bounds(day, 2)
    assert s.tolist() == s2.tolist() and e.tolist() == e2.tolist() and L == L2
    assert loader.padding_rows(s, e, L) == [0, 2]


def test_shard_range_partitions_exactly():
    for n, g in [(10, 1), (10, 3), (65536, 8), (7, 8), (1_000_003, 8)]:
        spans = [shard_range(n, r, g) for r in range(g)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(g - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_host_philox_mirrors_oracle():
    lib = fo.lib()
    for seed, ctr in [(0, 0), (99, 1), (2**63 + 5, 2**40 + 3)]:
        assert philox_u32(seed, ctr) == lib.fo_philox_u32(C.c_uint64(seed), C.c_uint64(ctr))
        assert redraw_day(seed, ctr, 64) == lib.fo_redraw_day(C.c_uint64(seed), C.c_uint64(ctr), C.c_int64(64))


def test_fastdiv_formula_used_by_the_kernel():
    """Granlund-Montgomery constants of csrc/fe_env.hip:make_fastdiv, checked in Python."""
    rng = np.random.default_rng(0)

    def make(d):
        l = 0
        while (1 << l) < d:
            l += 1
        m = ((1 << 32) * ((1 << l) - d)) // d + 1
        return m & 0xFFFFFFFF, min(l, 1), max(l - 1, 0)

    for d in [1, 2, 3, 5, 7, 30, 64, 150, 160, 975, 9600, 29250, 2**24, 2**31 - 1, 2**31 + 11]:
        m, s1, s2 = make(d)
        ns = np.concatenate([rng.integers(0, 2**32, 2000, dtype=np.uint64), np.array([0, 1, d - 1, d, d + 1, 2**32 - 1], dtype=np.uint64)])
        for n in ns.tolist():
            t = (m * n) >> 32
            q = (t + (((n - t) & 0xFFFFFFFF) >> s1)) >> s2
            assert q == n // d, (d, n)


def test_synthetic_generator_spec():
    p, day, minute = synthetic.synthetic_series(3, 2, 390)
    assert p.shape == (1170, 8) and minute.min() == 570 and minute.max() == 959
    assert np.all(p[:, 1] >= np.maximum(p[:, 0], p[:, 3]) - 1e-4) and np.all(p[:, 2] <= np.minimum(p[:, 0], p[:, 3]) + 1e-4)
    assert abs(p[0, 0] - 100) < 1 and abs(p[0, 4] - 110) < 1
    assert np.array_equal(np.round(p, 4), p)


def test_bench_helpers_and_cli_parse_without_a_gpu():
    """bench.py is what the driver runs unattended: its byte formulas, the repeat heuristic and the argument parser
    must at least import and behave on the CPU (SURVEY 8d figures)."""
    import importlib.util
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # B = 72*W*A + 84*A + 36; the HBM part drops the 32*W*A window re-read
    assert bench.survey_bytes(64, 1) == 4728 and bench.survey_bytes(64, 30) == 140796 and bench.survey_bytes(128, 30) == 279036
    assert bench.hbm_bytes(64, 1, 8) == 2680 == bench.survey_bytes(64, 1) - 32 * 64
    assert bench.hbm_bytes(128, 30, 8) == 156156 and bench.hbm_bytes(64, 1, 4) == 1400
    assert bench.auto_repeats(7, 20, 3e-5) == 7 and 5 <= bench.auto_repeats(0, 20, 3.4e-5) <= 40
    assert bench.auto_repeats(0, 20, 25e-3) == 5 and bench.auto_repeats(0, 20, 1e-6) == 40
    assert set(bench.CONFIGS) == {1, 2, 3, 4, 5} and bench.CONFIGS[2][1:] == (65536, 1, 64)
    # roofline.kernel: the instantiation names rocprofv3 prints (VEC follows fe_env_create: 16-byte packs unless W*5*A is odd)
    assert bench.step_kernel_name(64, 1, False, 1) == "fe_env_kernel<double, 2, true, false, 1>"
    assert bench.step_kernel_name(128, 30, False, 1) == "fe_env_kernel<double, 2, false, false, 1>"
    assert bench.step_kernel_name(64, 1, True, 3) == "fe_env_kernel<float, 4, true, false, 3>"
    assert bench.step_kernel_name(7, 3, False, 0) == "fe_env_kernel<double, 1, false, false, 0>"   # 105 elements: odd
    assert bench.step_kernel_name(7, 2, True, 0) == "fe_env_kernel<float, 2, false, false, 0>"     # 70 elements: pairs only
    assert bench.is_step_kernel("void (anonymous namespace)::fe_env_kernel<double, 2, true, false, 1>((anonymous namespace)::Params)")
    assert not bench.is_step_kernel("void (anonymous namespace)::fe_env_kernel<double, 2, true, true, 0>((anonymous namespace)::Params)")  # reset()
    # the audition and the settle phase are bounded
    assert bench.AUDITION_EXTRA <= 10 and bench.AUDITION_BUDGET <= 64 << 30 and 5.0 <= bench.SETTLE_MS <= 50.0
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


def test_bench_never_nests_a_profiler(monkeypatch):
    """bench.py's live counter passes start `rocprofv3 --pmc` children.  Under a profiler (rocprofv3 preloads a tool
    library and sets ROCP_TOOL_LIBRARIES) the child would inherit it and exec after GPU initialisation -- the case that
    takes a box of this pool down.  So: detect it and skip, and scrub the children's environment."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    clean = {"PATH": "/usr/bin", "LD_PRELOAD": "/usr/lib/libjemalloc.so", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    assert not bench.under_profiler(clean)
    for dirty in ({"ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/librocprofiler-sdk-tool.so"},
                  {"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so:/usr/lib/libjemalloc.so"},
                  {"HSA_TOOLS_LIB": "libfoo.so"}):
        assert bench.under_profiler(dict(clean, **dirty)), dirty
    env = bench.pmc_child_env(dict(clean, ROCP_TOOL_LIBRARIES="x", ROCPROF_OUTPUT_PATH="y", ROCPROFILER_LIBRARY_CTOR="1",
                                   LD_PRELOAD="/opt/rocm/lib/librocprofiler-sdk-tool.so:/usr/lib/libjemalloc.so"))
    assert env["LD_PRELOAD"] == "/usr/lib/libjemalloc.so" and env["TMPDIR"] == "/tmp"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # what multi-process GPU work needs stays
    assert not [k for k in env if k.startswith(("ROCP_", "ROCPROF"))]
    assert "LD_PRELOAD" not in bench.pmc_child_env({"LD_PRELOAD": "librocprofiler-sdk-tool.so"})
    # under a profiler the live pass refuses before it looks for rocprofv3, and remembers it for later workloads
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    t, why = bench.live_pmc_traffic(2, False, "device")
    assert t is None and "profiler" in why
    monkeypatch.delenv("ROCP_TOOL_LIBRARIES")
    assert bench.live_pmc_traffic(2, False, "device") == (None, why)


def test_trajectory_state_descriptor_protocol_on_host_tensors():
    """TrajectoryBuffer(states=True) without a GPU (host_rehearsal): packed layout, begin / state_slot order, the
    bootstrap row carried into the next chunk, capacity padding, errors."""
    import torch

    from finenvs_amd.trajectory import TrajectoryBuffer

    T, N, A, CAP = 4, 5, 2, 7
    buf = TrajectoryBuffer(T, N, A, device="cpu", host_rehearsal=True, capacity=CAP, states=True)
    # [rewards f64 (T) | obs_pos f64 (T+1, A) | obs_src i64 (T+1) | actions f32 (T, A) | dones i32 (T)] over CAP envs
    assert buf._nbytes == CAP * (8 * T + 8 * A * (T + 1) + 8 * (T + 1) + 4 * A * T + 4 * T)
    assert buf.obs_src.shape == (T + 1, N) and buf.obs_pos.shape == (T + 1, N, A)
    with pytest.raises(RuntimeError, match="begin"):
        buf.next_slot()
        buf.state_slot()
    buf.clear()
    desc = lambda k: (torch.arange(N) * 10 + k, torch.full((N, A), float(k), dtype=torch.float64))  # noqa: E731
    buf.begin(desc(0))
    with pytest.raises(RuntimeError, match="follows next_slot"):
        buf.state_slot()
    for t in range(T):
        a, r, d = buf.next_slot()
        a.fill_(t); r.fill_(t); d.zero_()
        src, pos = buf.state_slot()
        s2, p2 = desc(t + 1)
        src.copy_(s2); pos.copy_(p2)
    assert buf.full()
    for k in range(T + 1):
        assert torch.equal(buf.obs_src[k], desc(k)[0]) and torch.equal(buf.obs_pos[k], desc(k)[1])
    with pytest.raises(RuntimeError, match="start of a chunk"):
        buf.begin(desc(0))
    buf.clear()  # row T -> row 0
    assert torch.equal(buf.obs_src[0], desc(T)[0]) and torch.equal(buf.obs_pos[0], desc(T)[1])
    buf.mark_filled(2)
    assert len(buf) == 2
    with pytest.raises(ValueError):
        buf.mark_filled(T + 1)
    plain = TrajectoryBuffer(T, N, A, device="cpu", host_rehearsal=True)
    for fn in (lambda: plain.begin(desc(0)), plain.state_slot, lambda: plain.states(None, 0), lambda: plain.minibatch_states(None, torch.zeros(1))):
        with pytest.raises(RuntimeError, match="states=True"):
            fn()
    assert plain._nbytes == N * (8 * T + 4 * A * T + 4 * T)  # the compact layout is unchanged without states


def test_lstm_weight_packing_host_side_matches_the_oracle_packing():
    """finenvs_amd.rollout.lstm_row_order (torch, product) and oracle.fe_oracle.lstm_row_order / lstm_pack (numpy, test
    side) are written independently: same row permutation, same packed Wx (bias = one f32 add), and the fragment-major
    layout of the streaming kernels is a pure re-arrangement of the packed rows."""
    import numpy as np
    import torch

    from finenvs_amd.rollout import lstm_row_order
    from oracle import fe_oracle as fo

    rng = np.random.default_rng(0)
    for H in (32, 128, 256):
        order = lstm_row_order(H).numpy()
        assert np.array_equal(order, fo.lstm_row_order(H))
        w_ih = rng.normal(size=(4 * H, 5)).astype(np.float32)
        w_hh = rng.normal(size=(4 * H, H)).astype(np.float32)
        b_ih, b_hh = rng.normal(size=4 * H).astype(np.float32), rng.normal(size=4 * H).astype(np.float32)
        whh, wx = fo.lstm_pack(w_ih, w_hh, b_ih, b_hh)
        # what FusedLSTMRollout.set_weights builds (without touching a GPU)
        t_hh, t_ih = torch.from_numpy(w_hh), torch.from_numpy(w_ih)
        bias = torch.from_numpy(b_ih) + torch.from_numpy(b_hh)
        t_wx = torch.zeros((4 * H, 8), dtype=torch.float32)
        t_wx[:, :5] = t_ih[lstm_row_order(H)]
        t_wx[:, 5] = bias[lstm_row_order(H)]
        assert np.array_equal(t_wx.numpy(), wx) and np.array_equal(t_hh[lstm_row_order(H)].numpy(), whh)
        frag = t_hh[lstm_row_order(H)].reshape(4 * H // 32, 32, H // 8, 2, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)
        # element ((mt * NG + g) * 64 + lane) * 4 + c  ==  packed[32 mt + (lane & 31)][8 g + 4 (lane >> 5) + c]
        NG = H // 8
        for mt, g, lane, c in ((0, 0, 0, 0), (1, 2, 37, 3), (4 * H // 32 - 1, NG - 1, 63, 1), (3, 1, 31, 2)):
            assert frag[((mt * NG + g) * 64 + lane) * 4 + c] == whh[32 * mt + (lane & 31), 8 * g + 4 * (lane >> 5) + c]


def test_host_flag_poll_backs_off_and_rereads_after_the_timeout_synchronise():
    """poll_host_word (the host half of fe_env_step_notify; replaces the reference's per-step dones[-1].item(), TSE:510):
    a word that arrives late is returned, not reported as an error -- even when it only arrives while the timeout
    handler synchronises the stream --, a word that never arrives raises HostFlagTimeout after ONE
    call of the handler, and past the tight-spin phase the poll yields the CPU instead of burning it."""
    from finenvs_amd.environments.time_series_env import HostFlagTimeout, poll_host_word

    class Clock:  # a fake monotonic clock advanced by the fake sleep: no real waiting in the test
        def __init__(self):
            self.t, self.sleeps = 0.0, []

        def now(self):
            return self.t

        def sleep(self, dt):
            self.sleeps.append(dt)
            self.t += max(dt, 1e-3)

    seq = 7
    # 1. already there: no polling at all
    reads = []
    assert poll_host_word(lambda: reads.append(1) or (seq << 1 | 1), lambda v: v >> 1 == seq, 1.0) == (seq << 1 | 1)
    assert len(reads) == 1
    # 2. arrives during the back-off phase
    c, n = Clock(), [0]

    def late():
        n[0] += 1
        return (seq << 1) if n[0] > 50 else ((seq - 1) << 1 | 1)

    assert poll_host_word(late, lambda v: v >> 1 == seq, 10.0, spin=10, clock=c.now, sleep=c.sleep) == seq << 1
    assert c.sleeps and c.sleeps[0] == 0.0 and c.sleeps[-1] == 50e-6  # yields first, sleeps from 20 ms on
    # 3. arrives only because the timeout handler drained the stream: returned, no error
    c, word, calls = Clock(), [0], []

    def drain():
        calls.append(1)
        word[0] = seq << 1 | 1

    assert poll_host_word(lambda: word[0], lambda v: v >> 1 == seq, 0.5, on_timeout=drain, spin=3, clock=c.now, sleep=c.sleep) & 1
    assert calls == [1]
    # 4. never arrives: one handler call, then the error (with the last word in the text)
    c, calls = Clock(), []
    with pytest.raises(HostFlagTimeout, match="0x4"):
        poll_host_word(lambda: 4, lambda v: v >> 1 == seq, 0.5, on_timeout=lambda: calls.append(1), spin=3, clock=c.now, sleep=c.sleep)
    assert calls == [1]


def test_bench_legs_module_imports_against_the_running_bench_module():
    """tools/bench_legs.py is imported by bench.py only after the headline is in hand, inside try / except blocks under the watchdog:
    a NameError or a missing re-export there would not fail the run, it would silently turn every extra leg into an `error` entry.
    So: import it here the way bench.py does (`bench` registered in sys.modules first) and check the entry points exist."""
    import importlib.util
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    saved = sys.modules.get("bench")
    sys.modules["bench"] = bench
    try:
        spec.loader.exec_module(bench)
        spec2 = importlib.util.spec_from_file_location("bench_legs_under_test", os.path.join(root, "tools", "bench_legs.py"))
        legs = importlib.util.module_from_spec(spec2)
        spec2.loader.exec_module(legs)
    finally:
        if saved is None:
            sys.modules.pop("bench", None)
        else:
            sys.modules["bench"] = saved
    for name in ("two_stream_leg", "reference_semantics_leg", "strong_scaling_leg", "device_guard_check", "fused_rollout_legs"):
        assert callable(getattr(legs, name)), name
    assert legs.CONFIGS is bench.CONFIGS and legs.KernelTrain is bench.KernelTrain  # the running module's objects, not a second copy
    # every name the legs use from bench exists there (a rename in bench.py must not leave the legs behind)
    import ast

    tree = ast.parse(open(os.path.join(root, "tools", "bench_legs.py")).read())
    imported = [a.name for node in ast.walk(tree) if isinstance(node, ast.ImportFrom) and node.module == "bench" for a in node.names]
    assert imported and all(hasattr(bench, n) for n in imported), imported
