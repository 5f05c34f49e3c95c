"""CPU: every repository path the LIVE documents and sources cite exists.

Round 5 pruned `tools/` and left DESIGN.md, a profile summary and kernel comments pointing at scripts that were gone (VERDICT
round 5, weak #9).  The live set -- what a reader is sent to today -- is checked here: the root documents, the headers, the
product package, bench.py, tools/ and the current round's profile summaries.  NOTES.md and the per-round evidence folders of earlier
rounds are history: they name the scripts that produced them at the time; `profiles/README.md` says where those went."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIVE = (["DESIGN.md", "INTEGRATION.md", "README.md", "BASELINE.md", "bench.py", "__graft_entry__.py", "tools/README.md", "profiles/README.md"]
        + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
        + sorted(glob.glob(os.path.join(ROOT, "finenvs_amd", "**", "*.py"), recursive=True))
        + sorted(glob.glob(os.path.join(ROOT, "finenvs_amd", "csrc", "*.h")))
        + sorted(glob.glob(os.path.join(ROOT, "finenvs_amd", "csrc", "*.hip")))
        + sorted(glob.glob(os.path.join(ROOT, "finenvs_amd", "csrc", "*.cpp")))
        + sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
        + sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_*.md"))))
CITED = re.compile(r"(?<![\w/.<-])((?:tools|profiles|tests|oracle|include|examples|finenvs_amd)/[A-Za-z0-9_./-]*[A-Za-z0-9_])")


REFERENCE_PATHS = ("examples/time_series/", "tests/unit", "tests/integration", "finenvs_amd/...")  # the reference's own tree, cited for parity


def _exists(path):
    if any(ch in path for ch in "<>*{}") or path.startswith(REFERENCE_PATHS):
        return True
    full = os.path.join(ROOT, path)
    if os.path.exists(full):
        return True
    # a cited stem (`profiles/r05_c2` for r05_c2_summary.md / _kernel_stats.csv) or a built artefact that is git-ignored
    if glob.glob(full + "*") or path.endswith((".so", ".o")) or path.startswith("oracle/_ref"):
        return True
    return False


def test_every_path_cited_by_a_live_document_exists():
    missing = []
    for f in LIVE:
        full = f if os.path.isabs(f) else os.path.join(ROOT, f)
        if not os.path.exists(full):
            continue
        for n, line in enumerate(open(full, encoding="utf-8", errors="replace"), 1):
            if "git show" in line or "git log" in line:  # a pointer into history names what is no longer in the tree
                continue
            for m in CITED.finditer(line):
                p = m.group(1).rstrip(".")
                if not _exists(p):
                    missing.append(f"{os.path.relpath(full, ROOT)}:{n}: {p}")
    assert not missing, "cited but absent:\n" + "\n".join(missing)


def test_design_md_has_the_sections_the_tier_asks_for():
    s = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    for heading in ("## 0. The hot-path scope", "## 1. The path and its boundary", "## 2. Oracle", "## 4. Data layout in HBM", "## 5. Kernels",
                    "## 7. Multi-GPU", "## 8. Measurement protocol", "## 9. Known limits", "## 10. Out of scope"):
        assert heading in s, heading
