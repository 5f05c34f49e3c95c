"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/finenvs_amd.h declares, and refuses to compute without a GPU (no fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "finenvs_amd.h")          # the frozen surface
EXT_HEADER = os.path.join(REPO, "include", "finenvs_amd_ext.h")  # experimental policy heads + tuning hook


@pytest.fixture(scope="module")
def lib():
    from finenvs_amd import _lib

    return _lib.load()


def declared_symbols(header=None):
    """Function names a header declares (both headers when none is named)."""
    if header is None:
        return sorted(set(declared_symbols(HEADER)) | set(declared_symbols(EXT_HEADER)))
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fe_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_all_exported_and_bound(lib):
    from finenvs_amd import _lib

    for header, table in ((HEADER, _lib.SIGNATURES), (EXT_HEADER, _lib.EXT_SIGNATURES)):
        names = declared_symbols(header)
        assert len(names) >= 9
        for n in names:
            assert hasattr(lib, n), f"{n} declared in {os.path.basename(header)} but not exported"
            assert n in table, f"{n} has no ctypes signature"
        assert sorted(table) == names
    assert not set(_lib.SIGNATURES) & set(_lib.EXT_SIGNATURES)


def test_the_frozen_header_stays_frozen():
    """include/finenvs_amd.h is SURVEY 8(b)'s list + the 8(f) rows; the policy heads and the tuning hook are confined to
    the experimental header, which takes no new exports (task 7)."""
    core, ext = set(declared_symbols(HEADER)), set(declared_symbols(EXT_HEADER))
    for n in ("fe_version", "fe_env_create", "fe_env_bind_state", "fe_env_reset_obs", "fe_env_step", "fe_env_set_day",
              "fe_env_destroy", "fe_build_logret"):  # SURVEY 8(b) (fe_cpu_step deliberately absent: no CPU path in the product)
        assert n in core
    assert not any("lstm" in n or "mlp" in n or "policy" in n or n == "fe_env_set_launch" for n in core)
    assert ext == {"fe_policy_table", "fe_env_rollout_table", "fe_env_rollout_mlp", "fe_env_rollout_lstm",
                   "fe_lstm_split_workspace_floats", "fe_env_rollout_lstm_split", "fe_lstm_forward", "fe_lstm_activations",
                   "fe_env_set_launch"}
    assert len(core) == 34


def test_version_and_struct_layout(lib, tmp_path):
    from finenvs_amd import _lib

    assert lib.fe_version() == _lib.FE_ABI_VERSION
    # the ctypes mirror of fe_config must match the C compiler's layout
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "finenvs_amd.h"\n'
        'int main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(fe_config), offsetof(fe_config, W),'
        " offsetof(fe_config, starting_balance), offsetof(fe_config, seed), offsetof(fe_config, eval_env));return 0;}\n"
    )
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    F = _lib.FeConfig
    assert got == [C.sizeof(F), F.W.offset, F.starting_balance.offset, F.seed.offset, F.eval_env.offset]


def test_argument_errors_do_not_need_a_gpu(lib):
    from finenvs_amd import _lib

    out = C.c_void_p()
    cfg = _lib.FeConfig(4, 2, 10, 20, 1, 5, 0, 1e4, 0.01, 1.5, 0.25, 0, 0, 0, -1)  # W >= L
    assert lib.fe_env_create(C.byref(cfg), 8, 8, C.byref(out)) == -1
    assert b"W" in lib.fe_last_error()
    cfg = _lib.FeConfig(4, 2, 10, 4, 300, 5, 0, 1e4, 0.01, 1.5, 0.25, 0, 0, 0, -1)  # too many assets
    assert lib.fe_env_create(C.byref(cfg), 8, 8, C.byref(out)) == -1
    assert lib.fe_env_create(None, 8, 8, C.byref(out)) == -1
    assert lib.fe_env_step(None, 8, 8, 8, 8, None) == -1
    assert lib.fe_build_logret(None, None, 0, 0, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_fallback(lib):
    import finenvs_amd
    from finenvs_amd import _lib

    assert lib.fe_device_count() == 0
    out = C.c_void_p()
    cfg = _lib.FeConfig(4, 2, 10, 4, 1, 5, 0, 1e4, 0.01, 1.5, 0.25, 0, 0, 0, -1)
    assert lib.fe_env_create(C.byref(cfg), 8, 8, C.byref(out)) == -2
    assert b"no CPU path" in lib.fe_last_error()
    with pytest.raises(RuntimeError):
        finenvs_amd.TimeSeriesEnv(prices=[[1.0, 1, 1, 1]] * 50, day_id=[0] * 25 + [1] * 25, num_intervals=4)
    with pytest.raises(ValueError, match="HIP path only"):  # SURVEY 8(b)'s backend= keyword exists; "cpu" is refused, not emulated
        finenvs_amd.TimeSeriesEnv(prices=[[1.0, 1, 1, 1]] * 50, day_id=[0] * 25 + [1] * 25, num_intervals=4, backend="cpu")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under finenvs_amd/ may reference it."""
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "finenvs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"\b(from|import)\s+oracle\b|fe_oracle|libfe_oracle", txt):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_only_the_checkers_use_the_oracle():
    """oracle/ is the checker: only tests/ (incl. tests/soak/), __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import or execute it -- not the product (above), and not tools/ or examples/ either."""
    import ast

    bad = []
    for sub in ("tools", "examples"):
        for root, _, files in os.walk(os.path.join(REPO, sub)):
            for f in files:
                if f.endswith((".py", ".sh", ".hip", ".c", ".cpp")):
                    txt = open(os.path.join(root, f), errors="ignore").read()
                    if re.search(r"\b(from|import)\s+oracle\b|fe_oracle\.(py|c)|libfe_oracle|import fe_oracle", txt):
                        bad.append(os.path.join(root, f))
    assert not bad, bad
    # bench.py: exactly one import of the oracle, inside cpu_baseline()
    tree = ast.parse(open(os.path.join(REPO, "bench.py")).read())
    where = [fn.name for fn in ast.walk(tree) if isinstance(fn, ast.FunctionDef)
             for node in ast.walk(fn) if isinstance(node, ast.ImportFrom) and node.module == "oracle"]
    assert where == ["cpu_baseline"], where
    top = [node for node in tree.body if isinstance(node, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(node)]
    assert not top


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md must map every symbol of the header to the reference code it replaces."""
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    missing = [n for n in declared_symbols() if n not in doc]
    assert not missing, f"INTEGRATION.md does not mention: {missing}"


def test_header_cites_reference_lines_for_every_compute_entry_point():
    text = open(HEADER).read() + open(EXT_HEADER).read()
    blocks = re.findall(r"/\*(.*?)\*/\s*((?:int|int64_t|const char \*)\s*\*?fe_[a-z0-9_]+\s*\([^;]*;(?:\s*(?:int|int64_t)\s+fe_[a-z0-9_]+\s*\([^;]*;)*)", text, flags=re.S)
    cited = {}
    for comment, decls in blocks:
        for name in re.findall(r"\b(fe_[a-z0-9_]+)\s*\(", decls):
            cited[name] = bool(re.search(r"TSE:\d+|\.py:\d+", comment))
    housekeeping = {"fe_version", "fe_last_error", "fe_device_count", "fe_env_launch_info", "fe_env_destroy",
                    "fe_env_set_launch", "fe_build_tag", "fe_env_device", "fe_env_logret"}
    need = [n for n in declared_symbols() if n not in housekeeping]
    assert all(cited.get(n, False) for n in need), {n: cited.get(n) for n in need if not cited.get(n, False)}
