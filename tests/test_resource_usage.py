"""CPU (cross-compile only): no scratch memory in the step / reset / render kernels.

hipcc reports registers and scratch per kernel (-Rpass-analysis=kernel-resource-usage).  A streaming kernel that
spills keeps part of its software pipeline in memory behind the very store stream it is trying to feed; round 2's
f32 single-asset step kernels did (40 - 212 bytes per lane) while a comment claimed otherwise.  The table of this
build is committed as profiles/r05_resource_usage.txt (tools/resource_usage.py --out ...).
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def table():
    import resource_usage

    from finenvs_amd.csrc import build as hip_build

    if not os.path.exists(hip_build.HIPCC):
        pytest.skip("hipcc not available")
    return resource_usage.kernel_table()


def test_streaming_kernels_use_no_scratch(table):
    streaming = [r for r in table if re.match(r"fe_(env|env_promoted|render|describe)_kernel", r["name"])]
    # 5 (dtype x pack width) x 2 (single / multi asset) x (reset + four forms of the step), + render and describe
    assert len([r for r in streaming if r["name"].startswith("fe_env_kernel")]) == 50
    # promoted (f64-action) arithmetic: the tile loop for every (dtype, pack width), the pipeline for f64 observations; full forms
    assert len([r for r in streaming if r["name"].startswith("fe_env_promoted_kernel")]) == (5 + 2) * 2
    assert len([r for r in streaming if r["name"].startswith("fe_render_kernel")]) == 10
    bad = [(r["name"], r["scratch"], r["vgpr_spill"]) for r in streaming if r["scratch"] != 0 or r["vgpr_spill"] != 0]
    assert not bad, f"scratch / VGPR spills in streaming kernels: {bad}"


def test_mlp_rollout_kernels_use_no_scratch(table):
    """H = 128 (NT = 4) kept its sleeve state in scratch memory while libm's branchy tanhf was inlined 128 times into it;
    the tanh activation is now the exact-operation form of the LSTM head (task 7)."""
    mlp = [r for r in table if r["name"].startswith("fe_rollout_mlp_kernel")]
    assert len(mlp) == 6
    bad = [(r["name"], r["scratch"], r["vgpr_spill"]) for r in mlp if r["scratch"] != 0 or r["vgpr_spill"] != 0]
    assert not bad, f"scratch / VGPR spills in the MLP rollout kernels: {bad}"


def test_step_kernels_keep_their_occupancy(table):
    """The launch geometry (fe_env.hip:configure_launch) assumes these wavefronts per SIMD."""
    want = {
        "fe_env_kernel<double, 2, true, false, 0>": 4,   # single asset, f64: 4 workgroups per CU (lean form)
        "fe_env_kernel<double, 2, true, false, 1>": 4,   # ... full form
        "fe_env_kernel<double, 2, true, false, 2>": 4,   # ... lean + host flag
        "fe_env_kernel<float, 4, true, false, 0>": 6,    # single asset, f32: 6
        "fe_env_kernel<double, 2, false, false, 0>": 6,  # multi asset
        "fe_env_kernel<double, 2, true, true, 0>": 7,    # reset()
    }
    got = {r["name"]: r["occupancy"] for r in table}
    for name, waves in want.items():
        assert got[name] >= waves, (name, got[name], waves)


def test_lean_step_kernel_has_fewer_scalar_spills_than_the_full_one(table):
    """fe_env_kernel<..., FORM = 0> exists so that the optional outputs' pointers never become live scalars
    (profiles/r03_microbench/lean_vs_full.txt): 48 SGPR spills in round 2's only form, ~10 in the lean one (17 since round 5,
    when the lean forms took over the action copy: one more resident pointer, same launch time)."""
    got = {r["name"]: r["sgpr_spill"] for r in table}
    assert got["fe_env_kernel<double, 2, true, false, 0>"] <= 20
    assert got["fe_env_kernel<double, 2, true, false, 0>"] < got["fe_env_kernel<double, 2, true, false, 1>"]
    assert got["fe_env_kernel<double, 2, true, false, 2>"] < got["fe_env_kernel<double, 2, true, false, 1>"]


def test_committed_table_matches_this_build(table):
    path = os.path.join(ROOT, "profiles", "r05_resource_usage.txt")
    text = open(path).read()
    for r in table:
        if r["name"].startswith(("fe_env_kernel", "fe_render_kernel")):
            line = next((ln for ln in text.splitlines() if ln.startswith(r["name"][:58] + " ")), None)
            assert line is not None, f"{r['name']} missing from {path}: regenerate it"
            cols = line[58:].split()
            assert int(cols[3]) == r["scratch"], (r["name"], line)


def test_params_is_the_first_kernel_argument_wherever_cold_parameters_are_read_from_the_kernarg_segment():
    """fe_step_kernel.h `Cold<true>` reinterprets the kernarg segment pointer as `Params *` (FORM 3 of the single-asset step kernels:
    mode-specific parameters are s_loaded at their use instead of living in spilled SGPRs).  That is only right while `Params` is the
    FIRST by-value argument, at offset 0, of every kernel that reaches account_core / the evaluate tail -- nothing in the language
    enforces it.  The code object's own metadata does: every kernel that takes Params takes it first, and the step
    kernels take nothing else."""
    import resource_usage

    from finenvs_amd.csrc import build as hip_build

    if not os.path.exists(hip_build.HIPCC):
        pytest.skip("hipcc not available")
    args = resource_usage.kernel_arguments()
    step = {n: a for n, a in args.items() if n.startswith(("fe_env_kernel<", "fe_env_promoted_kernel<"))}
    assert len(step) == 64
    sizes = {a[0][1] for a in step.values()}
    assert len(sizes) == 1, sizes
    params_size = sizes.pop()
    for name, a in step.items():
        assert a[0] == (0, params_size, "by_value"), (name, a[0])
        assert all(kind.startswith("hidden_") for _, _, kind in a[1:]), (name, a[1:])  # Params is the ONLY explicit argument
    # the instantiations that actually launder the pointer today
    assert "fe_env_kernel<double, 2, true, false, 3>" in step and "fe_env_promoted_kernel<double, 2, true, 3>" in step
    # every other kernel that takes the parameter block takes it first as well (a fused rollout reusing account_core<., 3> would
    # read the right bytes)
    for name, a in args.items():
        if any(size == params_size and kind == "by_value" for _, size, kind in a):
            assert a[0] == (0, params_size, "by_value"), (name, a[:2])
