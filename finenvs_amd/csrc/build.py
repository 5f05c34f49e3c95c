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
    deps = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(REPO, "include", "finenvs_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [HIPCC] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ["-o", LIB]
        for knob in ("FE_MIN_WAVES_PER_EU", "FE_STORE_AUX", "FE_ROLLOUT_WAVES", "FE_ROLLOUT_NOPOLICY"):  # tuning experiments only
            if os.environ.get(knob):
                cmd.insert(1, f"-D{knob}=" + os.environ[knob])
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
