"""GPU: bench.py honours the driver's contract -- ONE JSON line on stdout (nothing else), the required keys, the
roofline / cpu_baseline objects, a timed region of exactly `steps` steps -- in its short form (no extra legs)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    env = dict(os.environ, FE_CPU_THREADS="4")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extra",
                          "--no-pmc"], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}"
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["unit"] == "env-steps/s" and "workload" in d["config"] and "64k envs" in d["config"]["workload"]
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    # achieved = algorithmic HBM bytes per launch / the kernel's HIP-event launch interval
    assert abs(r["achieved"] - r["hbm_bytes_per_env_step"] * r["units_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
    assert r["hbm_bytes_per_env_step"] == 2680 and r["units_per_launch"] == 65536
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.01  # the kernel cannot take longer than the wall step (trains alternate with the blocks)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "env-steps/s" and "sample" in c
    assert "redraw_contract" in d["config"]
    # the headline runs the reference-pinned mode: redraw='torch', whose step is a host-flag form of the kernel (the lean one:
    # rewards / dones / action copy into trajectory slots need no more)
    assert d["config"]["eval_redraw"] == "torch"
    assert r["kernel"] == r["timed_loop_kernel"] == "fe_env_kernel<double, 2, true, false, 2>"  # what loop AND trains launch
    assert d["config"]["timed_region"].startswith("STEADY STATE")
    lay = r["kernel_train_layout"]
    assert lay["repeats"] >= 3 and lay["loop_launches_per_block"] == 20 and lay["train_launches"] >= 40
    # the un-auditioned regime is in the line too: same loop and fences on the ring as allocated
    aa = d["as_allocated"]
    assert aa["value"] > 0 and aa["kernel_ms"] > 0 and aa["blocks"] >= 3
    assert abs(aa["value"] - 65536 / (aa["ms_per_step"] * 1e-3)) / aa["value"] < 1e-6
    assert d["config"]["obs_ring_audition"]["candidates"] <= 2 + d["config"]["obs_ring_audition_bound"]["extra_candidates"]


def test_default_driver_command_every_leg_kernel_below_step_and_both_scaling_readings():
    """The driver's N = 1 command in full (`--steps 20 --warmup 5`): for the headline AND every extra_configs leg the kernel
    interval is at most the wall step (1 % slack; VERDICT round 4 #4: config 3 once read 3.505 > 3.485 ms); the build's own redraw
    contract is a leg timed with the same protocol; the strong-scaling reading (64k envs in total) is in the line beside the
    weak one, with the per-GPU shard of a 2 / 4 / 8-GPU world measured on this GPU."""
    env = dict(os.environ, FE_CPU_THREADS="4")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-pmc", "--no-cpu"],
                         capture_output=True, text=True, timeout=1100, env=env, cwd=REPO)
    d = _one_json_line(out)
    assert d["config"]["eval_redraw"] == "torch" and d["scaling"] == "weak"
    legs = [d] + d["extra_configs"]
    assert [e["config"] for e in d["extra_configs"]] == [3, 4, 5]
    for leg in legs:
        assert "error" not in leg, leg.get("error")
        r = leg["roofline"]
        assert r["kernel_ms"] <= leg["ms_per_step"] * 1.01, (leg.get("workload", "headline"), r["kernel_ms"], leg["ms_per_step"])
        assert 0.3 < r["frac"] < 1.0 and r["kernel"] == r["timed_loop_kernel"] and r["kernel"].endswith(", 2>")
    dr = d["device_redraw"]
    assert "error" not in dr and dr["steps"] == 20 and dr["kernel"].endswith(", 0>") and dr["kernel_ms"] <= dr["ms_per_step"] * 1.01
    assert dr["value"] > 0.9 * d["value"]  # the two modes run the same arithmetic: within 10 % of each other either way
    rs = d["reference_semantics"]
    assert rs["steps"] == 20 and rs["blocks"] >= 5 and rs["value"] > 0
    st = d["strong_scaling"]
    assert "error" not in st and st["total_envs"] == 65536 and st["world"] == 1 and st["envs_per_gpu"] == 65536
    for mode in ("eager", "graph_k8", "graph_k32"):
        assert st[mode]["no_all_gather"]["value"] > 0 and "with_all_gather" not in st[mode]
    assert [p["world"] for p in st["shard_preview"]] == [2, 4, 8]
    assert [p["envs_per_gpu"] for p in st["shard_preview"]] == [32768, 16384, 8192]
    for p in st["shard_preview"]:
        assert p["emulated_on_one_gpu"] and p["eager"]["kernel_us"] < p["eager"]["no_all_gather"]["us_per_step"] * 1.01
        # fewer envs per GPU: a step gets shorter (how much shorter is the strong-scaling curve)
        assert p["graph_k8"]["no_all_gather"]["us_per_step"] < st["graph_k8"]["no_all_gather"]["us_per_step"]


def _check_strong_and_guard(m, ranks, envs_per_gpu):
    st = m["strong"]
    assert st["total_envs"] == 65536 and st["world"] == ranks and st["envs_per_gpu"] == envs_per_gpu and not st["emulated_on_one_gpu"]
    for mode, slots in (("eager", 20), ("graph_k8", 8), ("graph_k32", 32)):
        assert st[mode]["no_all_gather"]["value"] > 0 and st[mode]["with_all_gather"]["value"] > 0
        assert st[mode]["packed_bytes_per_rank_per_chunk"] == slots * envs_per_gpu * 16
    g = m["device_guard"]
    assert g["pass"] is True and len(g["ranks"]) == ranks
    assert all(r["env_device"] == r["local_rank"] for r in g["ranks"])


def _one_json_line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {out.stdout[:500]}"
    return json.loads(lines[0])


def _check_multi_gpu_object(m, ranks):
    for key in ("ranks_seen", "collective_backend", "trajectory_slots", "packed_bytes_per_rank_per_chunk", "value_with_all_gather",
                "value_no_all_gather", "gather_only_ms", "gather_only_ms_per_step_if_exposed", "exposed_ms_per_step"):
        assert key in m, key
    assert m["ranks_seen"] == ranks
    assert m["value_with_all_gather"] > 0 and m["value_no_all_gather"] > 0 and m["gather_only_ms"] > 0
    assert m["gathered_bytes_per_rank_per_chunk"] == ranks * m["packed_bytes_per_rank_per_chunk"]


def test_bench_n_gt_1_code_path_over_rccl_with_one_rank():
    """bench.py's N > 1 path -- RCCL process group, asynchronous trajectory all-gather legs, gather-only leg, the config-5
    rank shard -- with the one rank a one-GPU box allows (FE_BENCH_FORCE_DIST=1: a rehearsal knob the driver never sets).
    The first execution with more than one RCCL rank happens on the driver's 8-GPU node; everything but the transport is
    exercised here."""
    env = dict(os.environ, FE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29537")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-pmc"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    d = _one_json_line(out)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["cpu_baseline"] is None  # (the CPU leg belongs to the plain N = 1 line)
    assert d["multi_gpu"]["collective_backend"] == "nccl"
    _check_multi_gpu_object(d["multi_gpu"], 1)
    _check_strong_and_guard(d["multi_gpu"], 1, 65536)
    assert d["multi_gpu"]["kernel_form_by_rank"] == [2]
    assert "rccl" in d["multi_gpu"] and d["multi_gpu"]["rccl"] is not None
    # value is the with-all-gather leg
    assert abs(d["value"] - d["multi_gpu"]["value_with_all_gather"]) / d["value"] < 1e-9
    assert set(d["repeats"]) == {"with_all_gather", "no_all_gather"}
    # N > 1 runs carry exactly one extra leg: the per-GPU shard of config 5 (4M envs over 8 GPUs), error-free
    assert len(d["extra_configs"]) == 1
    c5 = d["extra_configs"][0]
    assert "error" not in c5, c5.get("error")
    assert c5["config"] == 5 and c5["envs_per_gpu"] == 524288 and c5["num_assets"] == 30 and c5["window"] == 128
    _check_multi_gpu_object(c5["multi_gpu"], 1)
    assert c5["multi_gpu"]["packed_bytes_per_rank_per_chunk"] == c5["multi_gpu"]["trajectory_slots"] * 524288 * (8 + 4 * 30 + 4)
    assert 0.3 < c5["roofline"]["frac"] < 1.0


def test_watchdog_prints_the_line_if_the_new_multi_gpu_legs_hang():
    """The strong-scaling leg and the DeviceGuard check run with more than one RCCL rank for the first time on the driver's node.
    If they hang there (a collective that never completes), every rank gives up after WATCHDOG_S and rank 0 still prints the
    line with the headline -- rehearsed here with a leg that sleeps forever and a 5 s watchdog (RCCL world 1)."""
    env = dict(os.environ, FE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29545", FE_BENCH_NO_EXTRA="1",
               FE_BENCH_WATCHDOG_S="5", FE_BENCH_HANG_STRONG="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-pmc"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    d = _one_json_line(out)
    assert d["value"] > 0 and d["multi_gpu"]["value_with_all_gather"] > 0
    assert "watchdog" in d["multi_gpu"]["strong"]["error"] and d["multi_gpu"]["device_guard"]["pass"] is False


def test_bench_two_ranks_through_torch_distributed_run():
    """The driver's launch line for N = 2 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 ... bench.py --gpus 2 ...`) on a one-GPU box: both ranks on device 0 over gloo (FE_BENCH_SINGLE_DEVICE /
    FE_BENCH_BACKEND: rehearsal knobs the driver never sets).  The launcher is started from a process that has not touched
    the GPU.  Checks the rank plumbing (RANK / LOCAL_RANK / WORLD_SIZE from the env), the sharded env (the evaluation env on
    the last rank), the collective legs with two real ranks, max-over-ranks timing, and that only rank 0 prints.  The config-5
    leg is skipped here: two ranks' shards (2 x 154 GB) do not fit one card."""
    env = dict(os.environ, FE_BENCH_SINGLE_DEVICE="1", FE_BENCH_BACKEND="gloo", FE_BENCH_NO_EXTRA="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-pmc"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    d = _one_json_line(out)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    _check_multi_gpu_object(d["multi_gpu"], 2)
    _check_strong_and_guard(d["multi_gpu"], 2, 32768)  # the strong-scaling leg: 64k envs in total over the two ranks
    assert d["multi_gpu"]["kernel_form_by_rank"] == [0, 2]  # only the last rank owns the evaluation env and polls the host flag
    assert d["multi_gpu"]["collective_backend"] == "gloo"
    # weak scaling: every rank owns the config's full env count; value counts all ranks' envs
    assert d["config"]["envs_per_gpu"] == 65536
    assert abs(d["value"] - 2 * 65536 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["extra_configs"] == []
