"""GPU: bench.py honours the driver's contract -- ONE strict JSON line on stdout (nothing else) of at most 8 000 characters, the
required keys, the roofline / cpu_baseline objects, a timed region of exactly `steps` steps -- and the full record of the same run
(`--detail PATH`; the same objects go to stderr as `[bench-detail]` lines) holds what the line summarises."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE_MAX = 8000  # the driver's visible tail of stdout (a 20 KB line was not parsed)


def _strict_loads(s):
    def refuse(tok):
        raise ValueError(f"non-strict JSON token {tok}")

    return json.loads(s, parse_constant=refuse)


def _one_json_line(out, rc=0):
    if out.returncode != rc:  # keep the whole stderr of a run that ended in an unexpected way (gpurun merges gpurun_out/ back)
        dump = os.path.join(REPO, "gpurun_out")
        if os.path.isdir(dump):
            with open(os.path.join(dump, f"bench_contract_failure_rc{out.returncode}.err"), "w") as f:
                f.write(out.stderr)
                f.write("\n---- stdout ----\n" + out.stdout)
    assert out.returncode == rc, (out.returncode, out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {out.stdout[:500]}"
    assert len(lines[0]) <= LINE_MAX, len(lines[0])
    d = _strict_loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    return d


def _run(args, env=None, timeout=900, detail=None, launcher=()):
    cmd = [sys.executable, *launcher, os.path.join(REPO, "bench.py"), *args]
    if detail is not None:
        cmd += ["--detail", str(detail)]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})), cwd=REPO)


def test_bench_prints_one_json_line_with_the_contract_fields(tmp_path):
    out = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extra", "--no-pmc"], {"FE_CPU_THREADS": "4"}, 600, tmp_path / "d.json")
    d = _one_json_line(out)
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["unit"] == "env-steps/s" and d["config"]["workload"] == "64k envs, 1 asset, window=64"
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    # achieved = algorithmic HBM bytes per launch / the kernel's HIP-event launch interval (the line carries 6 significant digits)
    assert abs(r["achieved"] - r["hbm_bytes_per_env_step"] * r["units_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-4
    assert r["hbm_bytes_per_env_step"] == 2680 and r["units_per_launch"] == 65536 and r["survey_8d_bytes_per_env_step"] == 4728
    assert r["bytes_model"].startswith("B_hbm = 40WA+84A+36") and abs(r["frac_on_survey_8d_bytes"] - 4728 / 2680 * r["frac"]) < 2e-3 and r["frac_on_survey_8d_bytes"] > 1.0
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.01  # the kernel cannot take longer than the wall step (trains alternate with the blocks)
    assert len(r["traffic_source"]) <= 80
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "env-steps/s" and "sample" in c
    # the headline runs the reference-pinned mode: redraw='torch', whose step is a host-flag form of the kernel (the lean one:
    # rewards / dones / action copy into trajectory slots need no more)
    assert d["config"]["eval_redraw"] == "torch" and d["config"]["launch_mode"].startswith("eager")
    assert r["kernel"] == "fe_env_kernel<double, 2, true, false, 2>"
    # the untimed steps are machine-readable: settle steps straight before each timed phase, and everything launched before `value`
    assert isinstance(d["config"]["settle_steps"], int) and d["config"]["settle_steps"] > 100
    assert d["config"]["untimed_steps_before_value"] >= d["config"]["settle_steps"] + d["warmup"]
    aa = d["as_allocated"]
    assert aa["value"] > 0 and aa["kernel_ms"] > 0
    # the full record of the same run
    full = json.load(open(tmp_path / "d.json"))
    h = full["headline"]
    assert h["value"] == d["value"] and h["roofline"]["timed_loop_kernel"] == r["kernel"]  # what loop AND trains launch
    lay = h["roofline"]["kernel_train_layout"]
    assert lay["repeats"] >= 3 and lay["loop_launches_per_block"] == 20 and lay["train_launches"] >= 40
    assert h["as_allocated"]["blocks"] >= 3 and h["obs_ring_audition"]["candidates"] <= 12
    assert "[bench-detail] {\"headline\"" in out.stderr


def test_default_driver_command_every_leg_kernel_below_step_and_both_scaling_readings(tmp_path):
    """The driver's N = 1 command in full (`--steps 20 --warmup 5`, live counter passes included): the line is small and strict;
    for the headline AND every extra_configs leg the kernel interval is at most the wall step (1 % slack) and `roofline.traffic`
    was MEASURED IN THIS RUN (the config-4 / 5 children used to die); the build's own redraw contract is a leg
    timed with the same protocol; the strong-scaling reading (64k envs in total) sits beside the weak one, with the per-GPU shard
    of a 2 / 4 / 8-GPU world measured on this GPU."""
    out = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu"], {"FE_CPU_THREADS": "4"}, 1100, tmp_path / "d.json")
    d = _one_json_line(out)
    assert d["config"]["eval_redraw"] == "torch" and d["scaling"] == "weak" and "unfinished" not in d and "dropped_for_size" not in d
    assert [e["config"] for e in d["extra_configs"]] == [3, 4, 5]
    for leg in [d] + d["extra_configs"]:
        assert "error" not in leg, leg.get("error")
        r = leg["roofline"]
        assert r["kernel_ms"] <= leg["ms_per_step"] * 1.01, (leg.get("workload", "headline"), r["kernel_ms"], leg["ms_per_step"])
        assert 0.3 < r["frac"] < 1.0 and r["kernel"].endswith(", 2>")
        # measured in this run -- or, where the box refuses the counter passes, the committed passes named beside the child's own cause
        # ("... or names a specific cause"); never a silent fallback
        assert r["traffic_source"].startswith("measured in this run") or (r["traffic_source"].startswith("profiles/hbm_traffic.json") and
                                                                          len(r.get("traffic_live_error", "")) > 10), (
            leg.get("workload", "headline"), r["traffic_source"], r.get("traffic_live_error"))
        assert 0.9 < r["traffic_over_algorithmic"] < 1.1
    legs = {x["leg"]: x for x in d["legs"]}
    assert set(legs) == {"device_redraw", "reference_semantics", "two_streams", "fused_linear_table", "fused_mlp_h64", "fused_lstm_h128", "fused_lstm_h1024"}
    assert all("error" not in x and x["value"] > 0 for x in legs.values())
    assert legs["device_redraw"]["value"] > 0.9 * d["value"]  # the two modes run the same arithmetic: within 10 % of each other either way
    st = d["strong_scaling"]
    assert st["total_envs"] == 65536 and st["world"] == 1 and st["envs_per_gpu"] == 65536
    for mode in ("eager", "graph_k8", "graph_k32"):
        assert st[mode]["no_all_gather"] > 0 and "with_all_gather" not in st[mode]
    pw = st["us_per_step_at_world"]
    assert pw["worlds"] == [1, 2, 4, 8]
    assert pw["graph_k8"][0] > pw["graph_k8"][1] > pw["graph_k8"][2] > pw["graph_k8"][3]  # fewer envs per GPU: a shorter step
    full = json.load(open(tmp_path / "d.json"))
    dr = full["legs_detail"]["device_redraw"]
    assert dr["steps"] == 20 and dr["kernel"].endswith(", 0>") and dr["kernel_ms"] <= dr["ms_per_step"] * 1.01
    rs = full["legs_detail"]["reference_semantics"]
    assert rs["steps"] == 20 and rs["blocks"] >= 5 and rs["value"] > 0
    assert [p["envs_per_gpu"] for p in full["strong_scaling"]["shard_preview"]] == [32768, 16384, 8192]
    for p in full["strong_scaling"]["shard_preview"]:
        assert p["emulated_on_one_gpu"] and p["eager"]["kernel_us"] < p["eager"]["no_all_gather"]["us_per_step"] * 1.01
    for e in full["extra_configs"]:
        assert e["roofline"]["kernel"] == e["roofline"]["timed_loop_kernel"]


def test_a_hung_leg_at_one_gpu_still_leaves_value_roofline_and_cpu_baseline():
    """at N = 1 the headline is measured first but printed last.  A leg that never returns (rehearsed with a
    sleep and a 5 s watchdog) must cost its own entry only: the line carries value / roofline / cpu_baseline, names the leg in
    `unfinished`, and the process ends with the watchdog's exit code (a hang is not a success)."""
    out = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-pmc"], {"FE_CPU_THREADS": "4", "FE_BENCH_WATCHDOG_S": "5", "FE_BENCH_HANG_LEG": "1"}, 600)
    d = _one_json_line(out, rc=3)
    assert d["value"] > 0 and 0.3 < d["roofline"]["frac"] < 1.0 and d["cpu_baseline"]["value"] > 0
    assert d["unfinished"] == ["rehearsal_hang"] and d["extra_configs"] == [] and d["legs"] == []
    assert "watchdog" in out.stderr


def _check_strong_and_guard(m, ranks, envs_per_gpu, full):
    st = m["strong"]
    assert st["total_envs"] == 65536 and st["world"] == ranks and st["envs_per_gpu"] == envs_per_gpu
    for mode, slots in (("eager", 20), ("graph_k8", 8), ("graph_k32", 32)):
        assert st[mode]["no_all_gather"] > 0 and st[mode]["with_all_gather"] > 0
        assert full["strong_scaling"][mode]["packed_bytes_per_rank_per_chunk"] == slots * envs_per_gpu * 16
    assert not full["strong_scaling"]["emulated_on_one_gpu"]
    assert m["device_guard"] == {"pass": True}
    g = full["device_guard"]
    assert g["pass"] is True and len(g["ranks"]) == ranks and all(r["env_device"] == r["local_rank"] for r in g["ranks"])


def _check_multi_gpu_object(m, ranks):
    """The keys the first SCALE record needs to explain itself, numbers only."""
    for key in ("ranks_seen", "collective_backend", "kernel_form_by_rank", "trajectory_slots", "packed_bytes_per_rank_per_chunk",
                "with_all_gather", "no_all_gather", "gather_only_ms", "gather_only_inbound_GBps_per_gpu", "exposed_ms_per_step"):
        assert key in m, key
    assert m["ranks_seen"] == ranks and len(m["kernel_form_by_rank"]) == ranks
    assert m["with_all_gather"] > 0 and m["no_all_gather"] > 0 and m["gather_only_ms"] > 0


def test_bench_n_gt_1_code_path_over_rccl_with_one_rank(tmp_path):
    """bench.py's N > 1 path -- RCCL process group, asynchronous trajectory all-gather legs, gather-only leg, the config-5
    rank shard -- with the one rank a one-GPU box allows (FE_BENCH_FORCE_DIST=1: a rehearsal knob the driver never sets).
    The first execution with more than one RCCL rank happens on the driver's 8-GPU node; everything but the transport is
    exercised here."""
    out = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-pmc"],
               {"FE_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29537"}, 900, tmp_path / "d.json")
    d = _one_json_line(out)
    full = json.load(open(tmp_path / "d.json"))
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["cpu_baseline"] is None  # (the CPU leg belongs to the plain N = 1 line)
    assert d["multi_gpu"]["collective_backend"] == "nccl"
    _check_multi_gpu_object(d["multi_gpu"], 1)
    _check_strong_and_guard(d["multi_gpu"], 1, 65536, full)
    assert d["multi_gpu"]["kernel_form_by_rank"] == [2]
    assert d["multi_gpu"]["rccl"] is not None and full["headline"]["multi_gpu"]["rccl"] is not None
    # value is the with-all-gather leg
    assert abs(d["value"] - d["multi_gpu"]["with_all_gather"]) / d["value"] < 1e-4
    assert set(full["headline"]["repeats"]) == {"with_all_gather", "no_all_gather"}
    hm = full["headline"]["multi_gpu"]
    assert hm["gathered_bytes_per_rank_per_chunk"] == hm["packed_bytes_per_rank_per_chunk"]
    # N > 1 runs carry exactly one extra leg: the per-GPU shard of config 5 (4M envs over 8 GPUs), error-free
    assert len(d["extra_configs"]) == 1
    c5 = d["extra_configs"][0]
    assert "error" not in c5, c5.get("error")
    assert c5["config"] == 5 and c5["envs_per_gpu"] == 524288
    _check_multi_gpu_object(c5["multi_gpu"], 1)
    assert c5["multi_gpu"]["packed_bytes_per_rank_per_chunk"] == c5["multi_gpu"]["trajectory_slots"] * 524288 * (8 + 4 * 30 + 4)
    assert 0.3 < c5["roofline"]["frac"] < 1.0


def test_watchdog_prints_the_line_if_the_new_multi_gpu_legs_hang():
    """The strong-scaling leg and the DeviceGuard check run with more than one RCCL rank for the first time on the driver's node.
    If they hang there (a collective that never completes), every rank gives up after WATCHDOG_S, rank 0 still prints the
    line with the headline, and every rank leaves with exit code 3 -- a hang is reported as a failure, with its record
     -- rehearsed here with a leg that sleeps forever and a 5 s watchdog (RCCL world 1)."""
    out = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-pmc"],
               {"FE_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29545", "FE_BENCH_NO_EXTRA": "1",
                "FE_BENCH_WATCHDOG_S": "5", "FE_BENCH_HANG_STRONG": "1"}, 600)
    d = _one_json_line(out, rc=3)
    assert d["value"] > 0 and d["multi_gpu"]["with_all_gather"] > 0
    assert d["unfinished"] == ["strong_scaling"] and d["multi_gpu"]["device_guard"] is None and "strong" not in d["multi_gpu"]


def test_bench_two_ranks_through_torch_distributed_run(tmp_path):
    """The driver's launch line for N = 2 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 ... bench.py --gpus 2 ...`) on a one-GPU box: both ranks on device 0 over gloo (FE_BENCH_SINGLE_DEVICE /
    FE_BENCH_BACKEND: rehearsal knobs the driver never sets).  The launcher is started from a process that has not touched
    the GPU.  Checks the rank plumbing (RANK / LOCAL_RANK / WORLD_SIZE from the env), the sharded env (the evaluation env on
    the last rank), the collective legs with two real ranks, max-over-ranks timing, that only rank 0 prints, and that the N > 1
    line obeys the size cap and carries the multi_gpu keys.  The config-5 leg is skipped here: two ranks' shards (2 x 154 GB)
    do not fit one card."""
    out = _run(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-pmc"],
               {"FE_BENCH_SINGLE_DEVICE": "1", "FE_BENCH_BACKEND": "gloo", "FE_BENCH_NO_EXTRA": "1"}, 900, tmp_path / "d.json",
               launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541"))
    d = _one_json_line(out)
    full = json.load(open(tmp_path / "d.json"))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    _check_multi_gpu_object(d["multi_gpu"], 2)
    _check_strong_and_guard(d["multi_gpu"], 2, 32768, full)  # the strong-scaling leg: 64k envs in total over the two ranks
    assert d["multi_gpu"]["kernel_form_by_rank"] == [0, 2]  # only the last rank owns the evaluation env and polls the host flag
    assert d["multi_gpu"]["collective_backend"] == "gloo"
    # weak scaling: every rank owns the config's full env count; value counts all ranks' envs
    assert d["config"]["envs_per_gpu"] == 65536
    assert abs(d["value"] - 2 * 65536 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["extra_configs"] == []
