"""ctypes binding of oracle/ebm_oracle.c (TEST INFRASTRUCTURE ONLY, see ebm_oracle.c header).

Used by tests/ as the parity checker at sizes where the NumPy oracle is too slow, and by
bench.py's cpu_baseline leg.  Never imported by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


class _Prefixed:
    """lib.ebmo_xyz -> the build's own prefix (ebmo_ for fp64, ebmol_ for the extended build)."""

    def __init__(self, lib, prefix):
        self._lib, self._prefix = lib, prefix

    def __getattr__(self, name):
        return getattr(self._lib, self._prefix + name[len("ebmo_"):] if name.startswith("ebmo_") else name)


class COracle:
    """fp64 C oracle (default), its OpenMP build (cpu_baseline), or — ``extended=True`` — the same
    source evaluated in 80-bit extended precision between fp64 inputs and outputs (a rounding-error
    measuring stick, see the header of ebm_oracle.c)."""

    def __init__(self, openmp=False, extended=False):
        name = "libebm_oracle_ld.so" if extended else ("libebm_oracle_omp.so" if openmp else "libebm_oracle.so")
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        self.extended = extended
        self.lib = _Prefixed(C.CDLL(path), "ebmol_" if extended else "ebmo_")
        self.lib.ebmo_max_threads.restype = C.c_int

    def max_threads(self):
        return int(self.lib.ebmo_max_threads())

    @staticmethod
    def par_vector(par: dict) -> np.ndarray:
        from ebm_oracle import PARAM_ORDER, default_parval
        return np.array([par.get(k, default_parval[k]) for k in PARAM_ORDER], dtype=np.float64)

    def geometry(self, kind, x, D):
        nx = len(x)
        out = np.zeros((8, nx))
        self.lib.ebmo_geometry(C.c_int(kind), C.c_int(nx), _ptr(np.ascontiguousarray(x)),
                               C.c_double(D), _ptr(out))
        return out

    def thomas(self, a, b, c, d):
        xs = np.zeros_like(b)
        self.lib.ebmo_thomas(C.c_int(len(b)), _ptr(a), _ptr(b), _ptr(c), _ptr(d), _ptr(xs))
        return xs

    def miz_run(self, kind, x, par, dt, ct, ft, fcol, state, nthreads=1, imex=False):
        """state: dict of [ncol, nx] C-contiguous float64 arrays Ei,Ew,h,D,phi,T0 (updated in
        place).  Returns dict of diagnostics Tw,Ti,n,E,T and counters (solves, failures).
        imex=True: the implicit-diffusion extension (NOT the reference's scheme, see ebm_oracle.py)."""
        nx = len(x)
        ncol = state["Ei"].shape[0]
        for k in ("Ei", "Ew", "h", "D", "phi", "T0"):
            assert state[k].flags.c_contiguous and state[k].shape == (ncol, nx)
        diag = {k: np.empty((ncol, nx)) for k in ("Tw", "Ti", "n", "E", "T")}
        ct = np.ascontiguousarray(ct, dtype=np.float64)
        ft = np.ascontiguousarray(ft, dtype=np.float64)
        fc = None if fcol is None else np.ascontiguousarray(fcol, dtype=np.float64)
        counters = (C.c_longlong * 2)(0, 0)
        pv = self.par_vector(par)
        xx = np.ascontiguousarray(x, dtype=np.float64)
        self.lib.ebmo_miz_run(
            C.c_int(kind), C.c_int(nx), C.c_int(ncol), _ptr(xx), _ptr(pv), C.c_double(dt),
            C.c_int(len(ct)), _ptr(ct), _ptr(ft), _ptr(fc),
            *[_ptr(state[k]) for k in ("Ei", "Ew", "h", "D", "phi", "T0")],
            *[_ptr(diag[k]) for k in ("Tw", "Ti", "n", "E", "T")],
            counters, C.c_int(nthreads), C.c_int(1 if imex else 0))
        return diag, (int(counters[0]), int(counters[1]))

    def zonal(self, x, par, dt, nlon, T):
        """The zonal diffusion substep (ebm_zonal_diffusion, an extension; include/ebm_hip.h) for T[nmember*nlon, nx]:
        returns (U, Z)."""
        T = np.ascontiguousarray(T, dtype=np.float64)
        ncol, nx = T.shape
        assert ncol % nlon == 0
        U, Z = np.empty_like(T), np.empty_like(T)
        self.lib.ebmo_zonal(C.c_int(nx), C.c_int(nlon), C.c_int(ncol // nlon), _ptr(np.ascontiguousarray(x, dtype=np.float64)),
                            C.c_double(par["D"]), C.c_double(par["cw"]), C.c_double(dt), _ptr(T), _ptr(U), _ptr(Z))
        return U, Z

    def miz2d_run(self, kind, x, par, dt, nlon, ct, ft, fcol, state):
        """CHECKER-ONLY EXPERIMENT (tests/test_oracle_zonal.py): the zonal substep coupled to the column step of the
        implicit-diffusion extension by operator splitting.  state holds Ei, Ew, h, D, phi, T0 AND T ([nmember*nlon, nx],
        updated in place; T is the previous step's output temperature, zeros for a start from rest).  Returns the
        diagnostics Tw, Ti, n, E (T is in the state) and the counters."""
        nx = len(x)
        ncol = state["Ei"].shape[0]
        assert ncol % nlon == 0
        for k in ("Ei", "Ew", "h", "D", "phi", "T0", "T"):
            assert state[k].flags.c_contiguous and state[k].shape == (ncol, nx) and state[k].dtype == np.float64, k
        diag = {k: np.empty((ncol, nx)) for k in ("Tw", "Ti", "n", "E")}
        ct = np.ascontiguousarray(ct, dtype=np.float64)
        ft = np.ascontiguousarray(ft, dtype=np.float64)
        fc = None if fcol is None else np.ascontiguousarray(fcol, dtype=np.float64)
        counters = (C.c_longlong * 2)(0, 0)
        self.lib.ebmo_miz2d_run(
            C.c_int(kind), C.c_int(nx), C.c_int(nlon), C.c_int(ncol // nlon), _ptr(np.ascontiguousarray(x, dtype=np.float64)),
            _ptr(self.par_vector(par)), C.c_double(dt), C.c_int(len(ct)), _ptr(ct), _ptr(ft), _ptr(fc),
            *[_ptr(state[k]) for k in ("Ei", "Ew", "h", "D", "phi", "T0")],
            *[_ptr(diag[k]) for k in ("Tw", "Ti", "n", "E")], _ptr(state["T"]),
            counters)
        return diag, (int(counters[0]), int(counters[1]))

    def T0eq(self, kind, x, par, ct, f, h, Ew, phi, T0):
        nx = len(x)
        res = np.empty(nx)
        pv = self.par_vector(par)
        self.lib.ebmo_T0eq(C.c_int(kind), C.c_int(nx), _ptr(np.ascontiguousarray(x)), _ptr(pv),
                           C.c_double(ct), C.c_double(f), _ptr(np.ascontiguousarray(h)),
                           _ptr(np.ascontiguousarray(Ew)), _ptr(np.ascontiguousarray(phi)),
                           _ptr(np.ascontiguousarray(T0)), _ptr(res))
        return res

    def classic_run(self, x, par, dt, ct_i, ct_ip1, ft, fcol, state, nthreads=1):
        """state: dict of [ncol, nx] arrays E, Tg (updated in place).  Returns dict T, h."""
        nx = len(x)
        ncol = state["E"].shape[0]
        out = {k: np.empty((ncol, nx)) for k in ("T", "h")}
        ct_i = np.ascontiguousarray(ct_i, dtype=np.float64)
        ct_ip1 = np.ascontiguousarray(ct_ip1, dtype=np.float64)
        ft = np.ascontiguousarray(ft, dtype=np.float64)
        fc = None if fcol is None else np.ascontiguousarray(fcol, dtype=np.float64)
        pv = self.par_vector(par)
        xx = np.ascontiguousarray(x, dtype=np.float64)
        self.lib.ebmo_classic_run(
            C.c_int(nx), C.c_int(ncol), _ptr(xx), _ptr(pv), C.c_double(dt), C.c_int(len(ft)),
            _ptr(ct_i), _ptr(ct_ip1), _ptr(ft), _ptr(fc), _ptr(state["E"]), _ptr(state["Tg"]),
            _ptr(out["T"]), _ptr(out["h"]), C.c_int(nthreads))
        return out
