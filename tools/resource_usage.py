"""Register / scratch table of every kernel in the product library (gfx950 cross-compile, no GPU needed).

    python tools/resource_usage.py [-DKNOB=V ...] [--out profiles/r03_resource_usage.txt]

Compiles finenvs_amd/csrc/fe_env.hip with -Rpass-analysis=kernel-resource-usage (to a scratch .so under /tmp, the
product library is not touched) and prints one line per kernel.  tests/test_resource_usage.py asserts on the same
table (no scratch in the step / render kernels).
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

FIELDS = (
    ("sgpr", r"TotalSGPRs"), ("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
    ("occupancy", r"Occupancy \[waves/SIMD\]"), ("sgpr_spill", r"SGPRs Spill"), ("vgpr_spill", r"VGPRs Spill"),
    ("lds", r"LDS Size \[bytes/block\]"),
)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return [re.sub(r"\(anonymous namespace\)::", "", line).strip() for line in out.splitlines()]


def kernel_table(defines=()):
    """[{name, sgpr, vgpr, agpr, scratch, occupancy, sgpr_spill, vgpr_spill, lds}] for every kernel of the build."""
    from finenvs_amd.csrc import build as B

    with tempfile.TemporaryDirectory(prefix="fe_ru_") as tmp:
        cmd = [B.HIPCC] + B.FLAGS + list(defines) + ["-Rpass-analysis=kernel-resource-usage"]
        cmd += [os.path.join(B.HERE, "fe_env.hip"), "-o", os.path.join(tmp, "ru.so")]
        err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    rows = []
    for block in re.split(r"remark: Function Name: ", err)[1:]:
        row = {"mangled": block.split()[0]}
        for key, pat in FIELDS:
            m = re.search(r"remark:\s+" + pat + r": (\d+)", block)
            row[key] = int(m.group(1)) if m else -1
        rows.append(row)
    for row, name in zip(rows, demangle([r["mangled"] for r in rows])):
        # "void fe_env_kernel<double, 2, true, false>(Params)" -> "fe_env_kernel<double, 2, true, false>"
        row["name"] = re.sub(r"^void ", "", re.sub(r"\(.*\)$", "", name))
    return rows


def kernel_arguments():
    """{demangled kernel name: [(offset, size, value_kind), ...]} from the code object's own metadata (device-only compile,
    clang-offload-bundler --unbundle, llvm-readelf --notes): what the hardware's kernarg segment of each kernel looks like."""
    import yaml

    from finenvs_amd.csrc import build as B

    llvm = os.path.join(os.path.dirname(B.HIPCC), "..", "lib", "llvm", "bin")
    with tempfile.TemporaryDirectory(prefix="fe_ka_") as tmp:
        co, elf = os.path.join(tmp, "dev.co"), os.path.join(tmp, "dev.elf")
        flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.run([B.HIPCC] + flags + ["--offload-device-only", "-c", os.path.join(B.HERE, "fe_env.hip"), "-o", co],
                       capture_output=True, text=True, check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={co}",
                        "--targets=hip-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"], capture_output=True, text=True, check=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", elf], capture_output=True, text=True, check=True).stdout
    doc = notes[notes.index("---"):]
    doc = doc[:doc.index("\n...") + 1] if "\n..." in doc else doc
    meta = yaml.safe_load(doc)
    kernels = meta["amdhsa.kernels"]
    names = demangle([k[".name"] for k in kernels])
    out = {}
    for k, name in zip(kernels, names):
        out[re.sub(r"^void ", "", re.sub(r"\(.*\)$", "", name))] = [(a[".offset"], a[".size"], a[".value_kind"]) for a in k.get(".args", [])]
    return out


def format_table(rows):
    lines = [f"{'kernel':58s} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'scratch B/lane':>14} {'vgpr spill':>10} {'sgpr spill':>10} {'waves/SIMD':>10} {'LDS static':>10}"]
    for r in sorted(rows, key=lambda r: r["name"]):
        lines.append(f"{r['name'][:58]:58s} {r['vgpr']:>4} {r['agpr']:>4} {r['sgpr']:>4} {r['scratch']:>14} {r['vgpr_spill']:>10} {r['sgpr_spill']:>10} {r['occupancy']:>10} {r['lds']:>10}")
    return "\n".join(lines)


if __name__ == "__main__":
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    text = format_table(kernel_table(defs))
    if "--out" in sys.argv:
        path = sys.argv[sys.argv.index("--out") + 1]
        head = "# hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage " + " ".join(defs) + "\n"
        with open(path, "w") as f:
            f.write(head + text + "\n")
    print(text)
