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
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.05  # the kernel cannot take longer than the wall step (5 % timing slack)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "env-steps/s" and "sample" in c
    assert "redraw_contract" in d["config"]
    assert r["kernel"] == "fe_env_kernel<double, 2, true, false, 1>"  # the instantiation the timed loop launches
    # the un-auditioned regime is in the line too: same loop and fences on the ring as allocated
    aa = d["as_allocated"]
    assert aa["value"] > 0 and aa["kernel_ms"] > 0 and aa["blocks"] >= 3
    assert abs(aa["value"] - 65536 / (aa["ms_per_step"] * 1e-3)) / aa["value"] < 1e-6
    assert d["config"]["obs_ring_audition"]["candidates"] <= 2 + d["config"]["obs_ring_audition_bound"]["extra_candidates"]
