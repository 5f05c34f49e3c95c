"""Build the gfx950 shared library behind include/finenvs_amd.h (in-tree, no JIT cache).

    python -m finenvs_amd.csrc.build            # -> finenvs_amd/csrc/libfinenvs_amd.so

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off is part of the
numerical contract (no FMA contraction across the reference's rounding points).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["fe_env.hip", "fe_csv.cpp"]
HEADERS = ["fe_device_common.h", "fe_step_kernel.h", "fe_activations.h", "fe_rollout_kernels.h", "fe_lstm_kernel.h", "fe_aux_kernels.h"]  # included by fe_env.hip
LIB = os.path.join(HERE, "libfinenvs_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
    "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unused-parameter",
    "-I", os.path.join(REPO, "include"),
]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(HERE, h) for h in HEADERS] + [os.path.join(REPO, "include", h) for h in ("finenvs_amd.h", "finenvs_amd_ext.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """The product library.  Takes NO tuning knobs from the environment: what is loaded by default is
    always the default build (experiments go through build_variant and an explicit path)."""
    if force or needs_build():
        tmp = f"{LIB}.{os.getpid()}.tmp"  # per process: several ranks may find the library missing at the same time
        cmd = [HIPCC] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ["-o", tmp]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)  # atomic: never a half-written library, whoever finishes last wins with identical bytes
    return LIB


VARIANT_KNOBS = ("FE_HOIST_FIRST", "FE_F32_WAVES", "FE_COLD_PARAMS")  # the live tunables of fe_device_common.h


def build_variant(tag: str, defines: dict, verbose: bool = False) -> str:
    """An experiment build: csrc/variants/libfinenvs_amd.<tag>.so with the given -D set and
    FE_BUILD_TAG = "<tag>:<defines>".  _lib.load() refuses a tagged library unless its path is passed
    explicitly (_lib.load(path)), so an interrupted experiment can never become the product."""
    for k in defines:
        if k not in VARIANT_KNOBS:
            raise ValueError(f"unknown knob {k}; known: {VARIANT_KNOBS}")
    out_dir = os.path.join(HERE, "variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"libfinenvs_amd.{tag}.so")
    desc = tag + ":" + ",".join(f"{k}={v}" for k, v in sorted(defines.items()))
    cmd = [HIPCC] + FLAGS + [f"-D{k}={v}" for k, v in defines.items()] + [f'-DFE_BUILD_TAG="{desc}"']
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    cmd += [os.path.join(HERE, s) for s in SOURCES] + ["-o", out]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
