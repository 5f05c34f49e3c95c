"""ctypes binding of include/finenvs_amd.h (the C ABI of the HIP hot path).

This is the only place the package touches the native library.  There is no
fallback: if the library cannot be loaded, or no HIP device is visible, the
environment refuses to construct.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libfinenvs_amd.so")

FE_ABI_VERSION = 5
FE_OK, FE_ERR_ARG, FE_ERR_HIP, FE_ERR_STATE = 0, -1, -2, -3
FE_MAX_ASSETS = 256


class FeConfig(C.Structure):
    """struct fe_config of include/finenvs_amd.h."""

    _fields_ = [
        ("N", C.c_int64), ("D", C.c_int64), ("L", C.c_int64),
        ("W", C.c_int32), ("A", C.c_int32),
        ("max_shares", C.c_int32), ("evaluate", C.c_int32),
        ("starting_balance", C.c_double), ("commission", C.c_double),
        ("init_margin", C.c_double), ("maint_margin", C.c_double),
        ("obs_is_f32", C.c_int32), ("redraw_mode", C.c_int32),
        ("seed", C.c_uint64), ("eval_env", C.c_int64),
    ]


# name -> (restype, argtypes); every symbol include/finenvs_amd.h (the frozen surface) declares
_vp, _i64, _i32 = C.c_void_p, C.c_int64, C.c_int32
SIGNATURES = {
    "fe_version": (C.c_int, []),
    "fe_last_error": (C.c_char_p, []),
    "fe_device_count": (C.c_int, []),
    "fe_env_create": (C.c_int, [C.POINTER(FeConfig), _vp, _vp, C.POINTER(_vp)]),
    "fe_env_bind_state": (C.c_int, [_vp] * 10),
    "fe_env_bind_f32_table": (C.c_int, [_vp, _vp]),
    "fe_env_bind_stats": (C.c_int, [_vp, _vp, _vp, _vp]),
    "fe_env_stats_reduce": (C.c_int, [_vp, _vp, _vp]),
    "fe_env_reset_obs": (C.c_int, [_vp, _vp, _vp]),
    "fe_env_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "fe_env_describe": (C.c_int, [_vp, _vp, _vp, _vp]),
    "fe_env_render": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "fe_env_step_notify": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp]),
    "fe_env_step_traj_notify": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp]),
    "fe_env_step_promoted": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp]),
    "fe_host_flag_create": (C.c_int, [C.POINTER(_vp)]),
    "fe_host_flag_destroy": (C.c_int, [_vp]),
    "fe_env_render_n": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "fe_env_check_descriptors": (C.c_int, [_vp, _vp, _i64, C.POINTER(_i64), _vp]),
    "fe_env_step_traj": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fe_env_rollout_linear": (C.c_int, [_vp, _vp, C.c_double, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fe_env_set_day": (C.c_int, [_vp, _i64, _i64, _vp]),
    "fe_env_launch_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "fe_env_destroy": (C.c_int, [_vp]),
    "fe_build_tag": (C.c_char_p, []),
    "fe_env_device": (C.c_int, [_vp]),
    "fe_env_logret": (_vp, [_vp]),
    "fe_build_logret": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "fe_build_logret_tables": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp]),
    "fe_build_tables": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "fe_traj_store": (C.c_int, [_i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fe_traj_returns": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, C.c_double, _vp, _vp, _vp]),
    "fe_csv_count_lines": (_i64, [C.c_char_p]),
    "fe_csv_read": (_i64, [C.c_char_p, _i64, _i32, _vp, _vp, _vp, _vp]),
}
# include/finenvs_amd_ext.h: the experimental in-kernel policy heads and the tuning hook (same library)
EXT_SIGNATURES = {
    "fe_policy_table": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "fe_env_rollout_table": (C.c_int, [_vp, _vp, _vp, C.c_double, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fe_env_rollout_mlp": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fe_env_rollout_lstm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, _i32, _i32, _i32, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _vp]),
    "fe_lstm_activations": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "fe_lstm_split_workspace_floats": (_i64, [_i32, _i64]),
    "fe_env_rollout_lstm_split": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, _i32, _i32, _i32, _vp, _vp, _vp, C.c_float, _vp, _vp,
                                            _vp, _vp, _vp, _vp, _vp, _vp]),
    "fe_lstm_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "fe_env_set_launch": (C.c_int, [_vp, _i32, _i32, _i32]),
}

_lib: Optional[C.CDLL] = None


class FinEnvsNativeError(RuntimeError):
    pass


def load(path: Optional[str] = None) -> C.CDLL:
    """Load libfinenvs_amd.so (building it in-tree first if it is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        try:
            from .csrc import build as _build

            _build.build()
        except Exception as exc:  # noqa: BLE001
            raise FinEnvsNativeError(
                f"native library {p} is missing and could not be built ({exc}); "
                "finenvs_amd has no CPU fallback"
            ) from exc
    lib = C.CDLL(p)
    for name, (res, args) in {**SIGNATURES, **EXT_SIGNATURES}.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.fe_version() != FE_ABI_VERSION:
        raise FinEnvsNativeError(f"ABI mismatch: library {lib.fe_version()} != binding {FE_ABI_VERSION}")
    tag = lib.fe_build_tag()
    if path is None and tag:
        raise FinEnvsNativeError(f"{p} is an experiment build ({tag.decode()}); rebuild the product library "
                                 "(python -m finenvs_amd.csrc.build --force) or load variants by explicit path")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, lib: Optional[C.CDLL] = None) -> None:
    if rc != 0:
        msg = (lib or load()).fe_last_error()
        raise FinEnvsNativeError(f"finenvs_amd native call failed ({rc}): {msg.decode() if msg else ''}")
