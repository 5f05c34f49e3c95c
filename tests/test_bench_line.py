"""CPU: the contract line bench.py prints is small, strict and complete whatever the legs measured.

Round 5's line was 20.2 KB and the driver did not parse it (BENCH_r05.json: parsed = null).  bench.compact_line / render_line are
pure functions of the run's full record: fed the committed round-5 record (profiles/r05_bench_driverlike.json, mapped to the
record's layout) and hostile variants of it (NaN legs, kilobyte error strings, an 8-rank multi_gpu object) they must give ONE
strict JSON line of at most 8 000 characters that still carries the contract keys, `roofline` and `cpu_baseline`."""
import copy
import importlib.util
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _strict_loads(s):
    def refuse(tok):
        raise ValueError(f"non-strict JSON token {tok}")

    return json.loads(s, parse_constant=refuse)


def _record_from_round5():
    """profiles/r05_bench_driverlike.json (the fat round-5 line) in the layout of bench.py's full record."""
    old = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_driverlike.json")).read().strip().splitlines()[-1])
    head = {"workload": old["config"]["workload"], "config": 2, "value": old["value"], "ms_per_step": old["ms_per_step"], "steps": old["steps"],
            "warmup": old["warmup"], "envs_per_gpu": old["config"]["envs_per_gpu"], "num_assets": 1, "window": 64,
            "obs_buffers": old["config"]["obs_buffers"], "as_allocated": old["as_allocated"], "launch_mode": old["config"]["launch_mode"],
            "settle_steps": 648, "untimed_steps_before_value": 1200, "repeats": old["repeats"], "roofline": dict(old["roofline"]),
            "cpu_baseline": old["cpu_baseline"]}
    head["roofline"].update(bytes_model="B_hbm = 40WA+84A+36 (SURVEY 8d's 72WA+84A+36 minus the 32WA cache-served window reads)",
                            frac_on_survey_8d_bytes=1.36)
    legs = {"device_redraw": {"value": old["device_redraw"]["value"], "ms_per_step": old["device_redraw"]["ms_per_step"], "frac": 0.77, "bound": "hbm"},
            "reference_semantics": {"value": old["reference_semantics"]["value"], "ms_per_step": old["reference_semantics"]["ms_per_step"], "frac": 0.7, "bound": "hbm"},
            "two_streams": {"value": old["two_streams"]["value"], "ms_per_step": old["two_streams"]["ms_per_step_all_envs"], "frac": 0.85, "bound": "hbm"}}
    for f in old["fused_rollouts"]:
        legs["fused_" + f["form"]] = {"value": f["value"], "ms_per_step": f["us_per_step"] * 1e-3, "frac": (f.get("roofline") or {}).get("frac"),
                                      "bound": "mfma" if f.get("roofline") else None}
    return {"n_gpus": 1, "steps": old["steps"], "dtype": "f64", "eval_redraw": "torch", "headline": head, "extra_configs": old["extra_configs"],
            "legs": legs, "strong_scaling": old["strong_scaling"], "device_guard": None, "unfinished": []}


def _check_line(s, n_gpus=1):
    assert "\n" not in s and len(s) <= 8000, len(s)
    d = _strict_loads(s)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms", "hbm_bytes_per_env_step",
              "survey_8d_bytes_per_env_step", "units_per_launch", "bytes_model", "frac_on_survey_8d_bytes"):
        assert k in r, k
    assert len(r["traffic_source"]) <= 80 and "B_hbm" in r["bytes_model"]
    for k in ("workload", "envs_per_gpu", "num_assets", "window", "obs_buffers", "eval_redraw", "launch_mode", "settle_steps"):
        assert k in d["config"], k
    assert isinstance(d["config"]["settle_steps"], int)
    # no prose: nothing in the line is a paragraph
    def longest(o):
        if isinstance(o, str):
            return len(o)
        if isinstance(o, dict):
            return max([longest(v) for v in o.values()] + [0])
        if isinstance(o, list):
            return max([longest(v) for v in o] + [0])
        return 0
    assert longest(d) <= 140
    return d


def test_round5_record_compacts_to_a_small_strict_line():
    b = _bench()
    rec = _record_from_round5()
    s = b.render_line(rec)
    d = _check_line(s)
    assert len(s) < 6000, len(s)  # (round 5 printed 20.2 KB for this very record)
    assert d["value"] == rec["headline"]["value"] and d["ms_per_step"] == rec["headline"]["ms_per_step"]  # the headline is not rounded
    assert d["config"]["workload"] == "64k envs, 1 asset, window=64" and d["dtype"] == "f64" and d["config"]["eval_redraw"] == "torch"
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and len(c["sample"]) <= 120 and c["reference_quoted"]["value"] == 40497
    assert [e["config"] for e in d["extra_configs"]] == [3, 4, 5]
    for e in d["extra_configs"]:
        assert 0.3 < e["roofline"]["frac"] < 1.0 and e["roofline"]["kernel_ms"] > 0 and "traffic_source" in e["roofline"]
    assert {x["leg"] for x in d["legs"]} >= {"device_redraw", "reference_semantics", "two_streams", "fused_mlp_h64", "fused_lstm_h128"}
    st = d["strong_scaling"]["us_per_step_at_world"]
    assert st["worlds"] == [1, 2, 4, 8] and len(st["eager"]) == len(st["graph_k8"]) == len(st["kernel_us"]) == 4
    assert "dropped_for_size" not in d and "unfinished" not in d


def test_non_finite_numbers_and_huge_strings_cannot_break_the_line():
    b = _bench()
    rec = _record_from_round5()
    rec["legs"]["two_streams"]["value"] = float("nan")
    rec["legs"]["fused_mlp_h64"]["frac"] = float("inf")
    rec["extra_configs"][1] = {"config": 4, "workload": "1M envs", "error": "HIP out of memory " * 400}
    rec["legs"]["reference_semantics"] = {"error": "x" * 5000}
    rec["headline"]["roofline"]["traffic_source"] = "y" * 900
    rec["headline"]["roofline"]["traffic_live_error"] = "z" * 900
    rec["unfinished"] = ["config5"]
    d = _check_line(b.render_line(rec))
    assert "legs[" in " ".join(d["nonfinite_set_to_null"]) and d["unfinished"] == ["config5"]
    assert any(x["leg"] == "two_streams" and x["value"] is None for x in d["legs"])
    assert len(d["extra_configs"][1]["error"]) <= 120 and len(d["roofline"]["traffic_live_error"]) <= 140


def test_an_8_rank_record_with_every_leg_stays_under_the_cap_and_sections_drop_in_order():
    b = _bench()
    rec = _record_from_round5()
    rec["n_gpus"] = 8
    mg = {"ranks_seen": 8, "collective_backend": "nccl", "kernel_form_by_rank": [0] * 7 + [2], "trajectory_slots": 20,
          "packed_bytes_per_rank_per_chunk": 20 * 65536 * 16, "value_with_all_gather": 1.7e10, "value_no_all_gather": 1.75e10,
          "gather_only_ms": 0.41, "gather_only_inbound_GBps_per_gpu": 357.0, "exposed_ms_per_step": 0.0004,
          "prediction": {"predicted_gather_only_ms": [0.49, 0.70]},
          "rccl": {"version": ["RCCL version 2.26.6+hip7.0 HEAD:abcdef"], "algorithm_protocol": ["AllGather: 20971520 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..31}"] * 5,
                   "topology": ["t" * 200] * 10}}
    rec["headline"]["multi_gpu"] = mg
    rec["headline"]["cpu_baseline"] = None
    st = copy.deepcopy(rec["strong_scaling"])
    st.pop("shard_preview")
    for mode in ("eager", "graph_k8", "graph_k32"):
        st[mode]["with_all_gather"] = dict(st[mode]["no_all_gather"])
    st["world"] = 8
    rec["strong_scaling"] = st
    rec["device_guard"] = {"pass": True, "ranks": [{"local_rank": i, "env_device": i, "pass": True} for i in range(8)]}
    c5 = copy.deepcopy(rec["extra_configs"][2])
    c5["multi_gpu"] = dict(mg, trajectory_slots=20)
    rec["extra_configs"] = [c5]
    rec["legs"] = {}
    s = b.render_line(rec)
    d = _check_line(s, n_gpus=8)
    m = d["multi_gpu"]
    for k in ("ranks_seen", "collective_backend", "with_all_gather", "no_all_gather", "gather_only_ms", "device_guard", "strong", "kernel_form_by_rank"):
        assert k in m, k
    assert m["device_guard"] == {"pass": True} and m["ranks_seen"] == 8 and d["cpu_baseline"] is None
    assert set(m["strong"]) >= {"eager", "graph_k8", "graph_k32"} and set(m["strong"]["eager"]) == {"no_all_gather", "with_all_gather"}
    assert d["extra_configs"][0]["multi_gpu"]["ranks_seen"] == 8
    # a record too big for the cap loses optional sections, never the contract
    big = copy.deepcopy(rec)
    big["legs"] = {f"leg{i:03d}": {"value": 1.0 * i, "ms_per_step": 0.1, "frac": 0.5, "bound": "hbm"} for i in range(200)}
    d2 = _check_line(b.render_line(big), n_gpus=8)
    assert "legs" in d2["dropped_for_size"] and "legs" not in d2 and "multi_gpu" in d2


def test_line_guard_prints_exactly_once_and_the_watchdog_keeps_the_headline(tmp_path):
    """LineGuard in a child process (no GPU, no torch work): the main thread hangs in a leg, the watchdog prints the line with the leg
    marked unfinished and ends the process with exit code 3; a normal run prints once although print_once() is called twice."""
    prog = r'''
import importlib.util, json, os, sys, time
sys.path.insert(0, %r)
spec = importlib.util.spec_from_file_location("b", os.path.join(%r, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
sys.path.insert(0, os.path.join(%r, "tests"))
import test_bench_line as T
rec = T._record_from_round5()
g = b.LineGuard(sys.stdout, rec, 0, None)
if sys.argv[1] == "hang":
    g.arm(0.5); g.current[0] = "config4"; time.sleep(60)
g.print_once(); g.print_once()
''' % (ROOT, ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", prog, "hang"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3, (out.returncode, out.stderr[-800:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = _check_line(lines[0])
    assert d["unfinished"] == ["config4"] and d["value"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    assert "watchdog" in out.stderr
    out = subprocess.run([sys.executable, "-c", prog, "ok"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-800:]
    assert len([ln for ln in out.stdout.splitlines() if ln.strip()]) == 1


def test_strict_names_what_it_nulled():
    b = _bench()
    bad = []
    out = b.strict({"a": float("nan"), "b": [1.0, float("-inf")], "c": {"d": 2}}, bad)
    assert out == {"a": None, "b": [1.0, None], "c": {"d": 2}} and bad == ["a", "b[1]"]
    json.dumps(out, allow_nan=False)


def test_the_counter_pass_child_in_flight_is_ended_by_its_own_process_group():
    """The watchdog (and a counter pass's own timeout) must not leave a rocprofv3 child behind on the GPU: bench.kill_child() ends the
    process group it started -- by that group's id, nothing else -- and reaps it."""
    b = _bench()
    proc = subprocess.Popen([sys.executable, "-c", "import subprocess, sys, time; subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(120)']); time.sleep(120)"],
                            start_new_session=True)
    time.sleep(0.5)
    b._CHILD[0] = proc
    t0 = time.perf_counter()
    b.kill_child()
    assert proc.poll() is not None and time.perf_counter() - t0 < 10
    b._CHILD[0] = None
    b.kill_child()  # nothing in flight: a no-op


def test_a_failed_counter_pass_is_described_by_the_childs_own_words(tmp_path):
    """`traffic_live_error` carries the child's last informative stderr line (round 5 threw the stderr away and reported "exited with
    1"): a Python traceback's last line starts with the exception's name and reason; chatter after it does not hide it."""
    b = _bench()
    p = tmp_path / "child.err"
    p.write_text("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n[pmc-child] config 4: 61.2 GiB free, observation ring 1 x 143.1 GiB\n"
                 "Traceback (most recent call last):\n  File \"bench.py\", line 1, in <module>\n"
                 "torch.OutOfMemoryError: HIP out of memory. Tried to allocate 143.05 GiB. GPU 0 has a total capacity of 287.98 GiB of which 60.83 GiB is free. " + "x" * 400 + "\n"
                 "some teardown chatter\n")
    got = b._last_words(str(p))
    assert got.startswith("torch.OutOfMemoryError: HIP out of memory. Tried to allocate 143.05 GiB") and len(got) <= 110
    p.write_text("only chatter\nmore chatter\n")
    assert b._last_words(str(p)) == "more chatter"
    assert b._last_words(str(tmp_path / "absent.err")) == ""
