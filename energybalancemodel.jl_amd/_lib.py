"""ctypes binding of the C ABI in include/ebm_hip.h (libebm_hip.so, built in-tree by
``csrc/Makefile`` / ``__graft_entry__.build()``).

This is the only compute path of the package: if the shared library is missing or no GPU is
present the calls fail loudly — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# EBM_LIB: alternative build of the same library (A/B benchmarking of kernel variants)
LIB_PATH = os.environ.get("EBM_LIB") or os.path.join(_HERE, "libebm_hip.so")

# enum ebm_model / ebm_grid / ebm_field / ebm_param (include/ebm_hip.h)
MODEL = {"MIZ": 0, "Classic": 1, "MIZ_IMEX": 2}     # MIZ_IMEX: extension, see include/ebm_hip.h
GRID = {"identity": 0, "nonuniform": 1}
FIELD = {"Ei": 0, "Ew": 1, "h": 2, "D": 3, "phi": 4, "T0": 5, "Tw": 6, "Ti": 7, "n": 8,
         "E": 9, "T": 10, "Tg": 11}
PARAM_ORDER = ("D", "A", "B", "cw", "S0", "S1", "S2", "a0", "a2", "ai", "Fb", "k", "Lf", "F",
               "cg", "tau", "Tm", "m1", "m2", "alpha", "rl", "Dmin", "Dmax", "hmin", "kappa")
EXPORTS = (
    "ebm_create", "ebm_create_ex", "ebm_options_default", "ebm_field_step", "ebm_get_field_as_of", "ebm_destroy", "ebm_last_error", "ebm_version", "ebm_set_field",
    "ebm_get_field", "ebm_hemispheric_mean", "ebm_hemispheric_mean_device", "ebm_get_field_device",
    "ebm_field_device_ptr", "ebm_diffusion", "ebm_zonal_diffusion", "ebm_set_column_forcing", "ebm_set_column_schedule",
    "ebm_set_step_clock", "ebm_set_time_table",
    "ebm_step", "ebm_run", "ebm_run_fused", "ebm_integrate", "ebm_integrate_hemispheric", "ebm_sync", "ebm_get_counters",
    "ebm_reset_counters", "ebm_timer_start", "ebm_timer_stop", "ebm_launch_info",
    "ebm_selftest_divide",
)

_dp = C.POINTER(C.c_double)
_lib = None

STATUS = {0: "EBM_OK", -1: "EBM_ERR_ARG", -2: "EBM_ERR_HIP", -3: "EBM_ERR_UNSUPPORTED", -4: "EBM_ERR_NO_DEVICE",
          -5: "EBM_ERR_STALE"}


class Options(C.Structure):
    """struct ebm_options (include/ebm_hip.h)."""
    _fields_ = [("struct_bytes", C.c_int), ("cells_per_thread", C.c_int), ("use_graph", C.c_int),
                ("prefetch_cols", C.c_int), ("launch_chains", C.c_int), ("integrate_steps_per_launch", C.c_int),
                ("fused_state_in_lds", C.c_int)]


class EBMError(RuntimeError):
    """A C-ABI call returned a negative ebm_status (``.status``)."""
    status = 0


class StaleFieldError(EBMError):
    """EBM_ERR_STALE: a diagnostic field (or the fp64 T0) older than the state was asked for."""


def load():
    """Load libebm_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EBMError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C csrc); "
            "this package has no CPU fallback")
    # One HIP runtime per process: PyTorch ships its own libamdhip64 and the system has another.  If this
    # library pulls in the system's first, a later `import torch` finds a runtime it did not expect
    # ("No HIP GPUs are available").  Importing torch first makes the dynamic loader resolve this
    # library's libamdhip64 dependency to the copy torch has already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.ebm_last_error.restype = C.c_char_p
    lib.ebm_version.restype = C.c_char_p
    lib.ebm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp,
                               C.c_double, C.c_int]
    lib.ebm_create_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp,
                                  C.c_double, C.c_int, C.POINTER(Options)]
    lib.ebm_options_default.argtypes = [C.POINTER(Options)]
    lib.ebm_field_step.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
    lib.ebm_get_field_as_of.argtypes = [C.c_void_p, C.c_int, C.c_longlong, _dp]
    lib.ebm_destroy.argtypes = [C.c_void_p]
    lib.ebm_set_field.argtypes = [C.c_void_p, C.c_int, _dp]
    lib.ebm_get_field.argtypes = [C.c_void_p, C.c_int, _dp]
    lib.ebm_hemispheric_mean.argtypes = [C.c_void_p, C.c_int, _dp]
    lib.ebm_hemispheric_mean_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.ebm_get_field_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.ebm_diffusion.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.ebm_zonal_diffusion.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
    lib.ebm_field_device_ptr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_longlong)]
    lib.ebm_set_column_forcing.argtypes = [C.c_void_p, _dp]
    lib.ebm_set_column_schedule.argtypes = [C.c_void_p, _dp]
    lib.ebm_set_step_clock.argtypes = [C.c_void_p, C.c_longlong]
    lib.ebm_set_time_table.argtypes = [C.c_void_p, C.c_int, _dp]
    lib.ebm_step.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int]
    lib.ebm_run.argtypes = [C.c_void_p, C.c_longlong, C.c_int, _dp, C.c_int]
    lib.ebm_run_fused.argtypes = [C.c_void_p, C.c_longlong, C.c_int, _dp, C.c_int, C.c_int]
    lib.ebm_integrate.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.POINTER(C.c_int), _dp, _dp, _dp, _dp]
    lib.ebm_integrate_hemispheric.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.c_int), _dp, _dp, _dp]
    lib.ebm_sync.argtypes = [C.c_void_p]
    lib.ebm_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    lib.ebm_reset_counters.argtypes = [C.c_void_p]
    lib.ebm_timer_start.argtypes = [C.c_void_p]
    lib.ebm_timer_stop.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.ebm_launch_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.ebm_selftest_divide.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().ebm_last_error().decode()
        err = (StaleFieldError if rc == -5 else EBMError)(f"{what} failed (status {rc} {STATUS.get(rc, '?')}): {msg}")
        err.status = rc
        raise err


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def as_f64(a, shape=None):
    out = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and out.shape != tuple(shape):
        raise ValueError(f"expected array of shape {tuple(shape)}, got {out.shape}")
    return out
