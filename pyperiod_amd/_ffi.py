"""ctypes binding of libperiod_hip.so (C ABI declared in include/periodhip.h).

The product has no CPU fallback: if the HIP library is missing or no GPU context can be
created, the error is raised to the caller.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libperiod_hip.so"
LIB_PATH = os.environ.get("PYPERIOD_AMD_LIB") or os.path.join(_HERE, LIB_NAME)  # override = tuning builds

PH_OK, PH_E_ARG, PH_E_HIP, PH_E_NOMEM, PH_E_CAP, PH_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
PH_F64, PH_F32 = 0, 1
PH_FLAG_TRUNC, PH_FLAG_ORTH, PH_FLAG_SINGLE, PH_FLAG_DEVICE, PH_FLAG_NOSYNC = 1, 2, 4, 8, 16
PH_STREAM_DEFAULT = 1  # ph_set_stream handle of the device default stream (its real handle, 0, means "own stream")
PH_SWEEP_NORM, PH_SWEEP_NORM_GAMMA, PH_SWEEP_MAXABS = 0, 1, 2
PH_ST_OK, PH_ST_NO_PERIOD, PH_ST_ITER_CAP, PH_ST_CAP = 0, 1, 2, 3

_vp, _i, _i64, _u, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint, C.c_double
_pi32 = C.c_void_p  # int32 tables are passed as raw addresses of numpy arrays

# name -> argtypes; every function returns int except ph_last_error.  This table is also what
# tests/test_abi.py checks against the header.
SIGNATURES = {
    "ph_version": [],
    "ph_device_count": [C.POINTER(_i)],
    "ph_create": [_i, C.POINTER(_vp)],
    "ph_destroy": [_vp],
    "ph_set_stream": [_vp, _vp],
    "ph_sync": [_vp],
    "ph_timer_begin": [_vp],
    "ph_timer_end": [_vp, C.POINTER(C.c_float)],
    "ph_device_info": [_vp, C.POINTER(_i), C.POINTER(_i)],
    "ph_max_window": [_vp, _i, _u, C.POINTER(_i)],
    "ph_sweep_plan_info": [_vp, _i, _i, C.POINTER(_i), C.POINTER(_i)],
    "ph_m_best_info": [_vp, _i, _i, _i, _i, _i, _u, C.POINTER(_i), C.POINTER(_i)],
    "ph_m_best_plan_info": [_vp, _i, _i, _i, _i, _i, _u, C.POINTER(_i), C.POINTER(_i)],
    "ph_periodic_norm": [_vp, _vp, _i, _i64, _i, _i, _u, _vp],
    "ph_project_batch": [_vp, _vp, _i, _i64, _i, _pi32, _i, _pi32, _pi32, _i, _u, _vp],
    "ph_sweep": [_vp, _vp, _i, _i64, _i, _i, _i, _i, _pi32, _pi32, _i, _u, _vp],
    "ph_m_best": [_vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _pi32, _pi32, _pi32, _pi32, _i, _u, _vp, _vp, _vp, _vp, _vp],
    "ph_profile_enable": [_vp, _i],
    "ph_profile_read": [_vp, C.POINTER(C.c_float), _i, C.POINTER(_i)],
    "ph_small_to_large": [_vp, _vp, _i, _i64, _i, _d, _i, _pi32, _pi32, _i, _u, _i, _vp, _vp, _vp, _vp, _vp],
    "ph_best_correlation": [_vp, _vp, _i, _i64, _i, _i, _i, _d, _pi32, _pi32, _i, _u, _vp, _vp, _vp, _vp],
    "ph_best_frequency": [_vp, _vp, _i, _i64, _i, _i, _i, _pi32, _pi32, _i, _u, _vp, _vp, _vp, _vp],
    "ph_ramanujan_norms": [_vp, _vp, _i, _i64, _i, _i, _i, _u, _vp],
    "ph_dict_project": [_vp, _vp, _vp, _i, _i, _u, _vp],
    "ph_qo_find_periods": [_vp, _vp, _i, _i64, _i, _i, _d, _i, _i, _i, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ph_qo_feasible": [_vp, _i, _i, _i, _i, C.POINTER(_i)],
    "ph_orth_powers": [_vp, _vp, _i, _i64, _i, _i, _i, _u, _vp, _vp, _vp],
    "ph_fold_sums": [_vp, _vp, _i, _i64, _i, _pi32, _pi32, _i, _u, _vp],
    "ph_tile_sum": [_vp, _vp, _i64, _i, _pi32, _pi32, _i, _i, _u, _vp],
}


class PeriodHipError(RuntimeError):
    """A HIP runtime call inside libperiod_hip.so failed."""


class CapacityError(RuntimeError):
    """An output slab was too small (PH_E_CAP); retry with a larger capacity."""


_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64 (same SONAMEs as
    /opt/rocm's).  Two HSA runtimes in one process cannot both own the GPU, so when torch is
    installed its runtime is loaded first -- by path, without importing torch -- and
    libperiod_hip.so then binds to that single copy.  PYPERIOD_AMD_SYSTEM_HIP=1 disables this.
    """
    if os.environ.get("PYPERIOD_AMD_SYSTEM_HIP") == "1":
        return
    import importlib.util
    import sys

    try:
        if "torch" in sys.modules:
            return  # already loaded its runtime
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def load():
    """Load libperiod_hip.so once; raise ImportError with the build recipe if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  pyperiod_amd has no CPU fallback."
        )
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _i
    lib.ph_last_error.argtypes = []
    lib.ph_last_error.restype = C.c_char_p
    lib.ph_profile_name.argtypes = [_vp, _i]
    lib.ph_profile_name.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc: int):
    """Map a PH_E_* status to the Python exception the reference's callers would see."""
    if rc == PH_OK:
        return
    msg = load().ph_last_error().decode("utf-8", "replace")
    if rc == PH_E_ARG:
        raise ValueError(msg)
    if rc == PH_E_NOMEM:
        raise MemoryError(msg)
    if rc == PH_E_CAP:
        raise CapacityError(msg)
    if rc == PH_E_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise PeriodHipError(msg)
