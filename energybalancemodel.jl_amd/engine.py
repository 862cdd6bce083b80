"""Engine: a thin object wrapper over one ebm_handle_t (include/ebm_hip.h).

Owns the device-resident state of ``ncol`` independent meridians (longitudes and/or ensemble
members) of ``nlat`` cells, including the T0 warm start that the reference keeps in a
module-level closure (reference src/miz.jl:47,64).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _lib
from ._lib import FIELD, GRID, MODEL, PARAM_ORDER, as_f64, check, dptr

MIZ_PROGNOSTIC = ("Ei", "Ew", "h", "D", "phi")
MIZ_DIAGNOSTIC = ("Tw", "Ti", "n", "E", "T")
CLASSIC_PROGNOSTIC = ("E", "Tg")
CLASSIC_DIAGNOSTIC = ("T", "h")


def schedule_words(forcing):
    """The 9 words of one column's schedule (include/ebm_hip.h): base, peak, cool, the two rates
    and Forcing.domain[1:]; a constant Forcing is a schedule that never leaves its first hold."""
    if forcing.constant:
        return [forcing.base, forcing.base, forcing.base, 0.0, 0.0, np.inf, np.inf, np.inf, np.inf]
    d = forcing.domain
    return [forcing.base, forcing.peak, forcing.cool, float(forcing.rates[0]), float(forcing.rates[1]),
            float(d[1]), float(d[2]), float(d[3]), float(d[4])]


def cos2pit(t: float) -> float:
    """cos(2.0*pi*t) exactly as the reference writes it (src/miz.jl:11, src/classic.jl:24)."""
    return math.cos(2.0 * math.pi * t)


def param_vector(par, defaults) -> np.ndarray:
    get = par.get if hasattr(par, "get") else (lambda k, d: getattr(par, k, d))
    return np.array([get(k, defaults[k]) for k in PARAM_ORDER], dtype=np.float64)


def _env_int(name):
    v = os.environ.get(name)
    return None if v in (None, "") else int(v)


class Engine:
    """``cells_per_thread`` (2 or 4; default 4), ``use_graph`` (True / False; default: by size), ``prefetch_cols`` and
    ``launch_chains`` (1 or 2; default 1), ``integrate_steps_per_launch`` (default 64; 1 = ``integrate`` launches every step)
    and ``fused_state_in_lds`` (True / False; default: by column count) are the launch options of ``ebm_create_ex`` (include/ebm_hip.h: struct ebm_options).
    The LIBRARY reads no environment variable; this mirror maps EBM_CELLS_PER_THREAD, EBM_GRAPH and
    EBM_PREFETCH_COLS to those options when the corresponding argument is left at None — the knobs of the
    test suite and of the A/B timing scripts under tests/tools/."""

    def __init__(self, model: str, grid_kind: str, x, params25, dt: float, ncol: int = 1,
                 device: int = 0, *, cells_per_thread=None, use_graph=None, prefetch_cols=None, launch_chains=None,
                 integrate_steps_per_launch=None, fused_state_in_lds=None):
        if model not in MODEL:
            raise ValueError(f"unknown model {model!r}: expected 'MIZ', 'Classic' or the extension 'MIZ_IMEX'")
        self.lib = _lib.load()
        self.model, self.grid_kind = model, grid_kind
        self.x = as_f64(x)
        self.nlat, self.ncol, self.dt = int(self.x.shape[0]), int(ncol), float(dt)
        self.params = as_f64(params25, (len(PARAM_ORDER),))
        opt = _lib.Options()
        check(self.lib.ebm_options_default(C.byref(opt)), "ebm_options_default")
        if cells_per_thread is None:
            cells_per_thread = _env_int("EBM_CELLS_PER_THREAD")
            if cells_per_thread == 2 and (self.nlat > 1536 or model == "MIZ_IMEX"):
                cells_per_thread = None              # the knob asks for 2 "where it exists"
        if use_graph is None and _env_int("EBM_GRAPH") is not None:
            use_graph = bool(_env_int("EBM_GRAPH"))
        if prefetch_cols is None:
            prefetch_cols = _env_int("EBM_PREFETCH_COLS")
        if cells_per_thread is not None:
            opt.cells_per_thread = int(cells_per_thread)
        if use_graph is not None:
            opt.use_graph = int(bool(use_graph))
        if prefetch_cols is not None:
            opt.prefetch_cols = max(0, int(prefetch_cols))
        if launch_chains is not None:
            opt.launch_chains = int(launch_chains)
        if integrate_steps_per_launch is not None:
            opt.integrate_steps_per_launch = int(integrate_steps_per_launch)
        if fused_state_in_lds is not None:
            opt.fused_state_in_lds = int(bool(fused_state_in_lds))
        h = C.c_void_p()
        gk = GRID["identity"] if grid_kind == "identity" else GRID["nonuniform"]
        check(self.lib.ebm_create_ex(C.byref(h), MODEL[model], gk, self.nlat, self.ncol,
                                     dptr(self.x), dptr(self.params), self.dt, int(device), C.byref(opt)),
              "ebm_create")
        self._h = h
        self.nt = None

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.ebm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- state ------------------------------------------------------------------------
    @property
    def prognostic(self):
        return MIZ_PROGNOSTIC if self.model.startswith("MIZ") else CLASSIC_PROGNOSTIC

    @property
    def diagnostic(self):
        return MIZ_DIAGNOSTIC if self.model.startswith("MIZ") else CLASSIC_DIAGNOSTIC

    def set_field(self, name: str, values):
        a = as_f64(values).reshape(self.ncol, self.nlat)
        check(self.lib.ebm_set_field(self._h, FIELD[name], dptr(a)), f"ebm_set_field({name})")

    def get_field(self, name: str) -> np.ndarray:
        out = np.empty((self.ncol, self.nlat))
        check(self.lib.ebm_get_field(self._h, FIELD[name], dptr(out)), f"ebm_get_field({name})")
        return out

    def get_field_as_of(self, name: str, step: int) -> np.ndarray:
        """The field as written by global step ``step`` (0-based), however far the state has moved on since;
        StaleFieldError if another step wrote it last (ebm_get_field_as_of)."""
        out = np.empty((self.ncol, self.nlat))
        check(self.lib.ebm_get_field_as_of(self._h, FIELD[name], int(step), dptr(out)), f"ebm_get_field_as_of({name})")
        return out

    def field_step(self, name: str) -> dict:
        """dict(written_step, state_step, current): which step last wrote the field, where the state is, and
        whether the field may be read now (ebm_field_step)."""
        w, s_, c = C.c_longlong(), C.c_longlong(), C.c_int()
        check(self.lib.ebm_field_step(self._h, FIELD[name], C.byref(w), C.byref(s_), C.byref(c)), "ebm_field_step")
        return dict(written_step=w.value, state_step=s_.value, current=bool(c.value))

    def set_state(self, state: dict):
        for k, v in state.items():
            self.set_field(k, v)

    def get_state(self, names=None) -> dict:
        names = names or (self.prognostic + self.diagnostic)
        return {k: self.get_field(k) for k in names}

    def hemispheric_mean(self, name: str) -> np.ndarray:
        """Per-column hemispheric mean of a field, reduced on the device (ebm_hemispheric_mean)."""
        out = np.empty(self.ncol)
        check(self.lib.ebm_hemispheric_mean(self._h, FIELD[name], dptr(out)), "ebm_hemispheric_mean")
        return out

    def hemispheric_mean_device(self, name: str, dev_ptr: int):
        """Same reduction, result left on the device at ``dev_ptr`` (``ncol`` doubles, e.g. the
        ``data_ptr()`` of a torch tensor on this engine's device)."""
        check(self.lib.ebm_hemispheric_mean_device(self._h, FIELD[name], C.c_void_p(dev_ptr)),
              "ebm_hemispheric_mean_device")

    def get_field_device(self, name: str, dev_ptr: int):
        """Device-to-device copy of a field, packed [ncol][nlat], to ``dev_ptr``."""
        check(self.lib.ebm_get_field_device(self._h, FIELD[name], C.c_void_p(dev_ptr)),
              "ebm_get_field_device")

    def diffusion(self, temp, base=None) -> np.ndarray:
        """``diffusion!(base, temp, st, par)`` / ``diffusion(T, st, par)`` (reference
        src/infrastructure.jl:495-533) per column on the device: base + D d/dx[(1-x^2) d temp/dx]."""
        t = as_f64(temp).reshape(self.ncol, self.nlat)
        b = None if base is None else as_f64(base).reshape(self.ncol, self.nlat)
        out = np.empty((self.ncol, self.nlat))
        check(self.lib.ebm_diffusion(self._h, dptr(t), dptr(b), dptr(out)), "ebm_diffusion")
        return out

    def zonal_diffusion(self, temp, nlon: int):
        """The zonal partner of ``diffusion`` as a backward-Euler substep over dt (ebm_zonal_diffusion; an extension, not in
        the reference): the columns are read as members of ``nlon`` longitudes each (column = member*nlon + longitude,
        periodic); returns (U, Z) for ``temp`` [ncol, nlat] — the zonally diffused field and the heat-flux convergence
        (U - temp) cw/dt = D/((1-x^2) dlambda^2) d2U/dl2."""
        t = as_f64(temp).reshape(self.ncol, self.nlat)
        U, Z = np.empty((self.ncol, self.nlat)), np.empty((self.ncol, self.nlat))
        check(self.lib.ebm_zonal_diffusion(self._h, int(nlon), dptr(t), dptr(U), dptr(Z)), "ebm_zonal_diffusion")
        return U, Z

    def field_device_ptr(self, name: str):
        p, pitch = C.c_void_p(), C.c_longlong()
        check(self.lib.ebm_field_device_ptr(self._h, FIELD[name], C.byref(p), C.byref(pitch)),
              "ebm_field_device_ptr")
        return p.value, pitch.value

    def set_column_forcing(self, fcol):
        a = None if fcol is None else as_f64(fcol, (self.ncol,))
        check(self.lib.ebm_set_column_forcing(self._h, dptr(a)), "ebm_set_column_forcing")

    def set_column_schedules(self, forcings):
        """Per-column Forcing schedules evaluated on the device (ebm_set_column_schedule):
        ``forcings`` is a sequence of ``ncol`` Forcing objects, or None to clear."""
        if forcings is None:
            check(self.lib.ebm_set_column_schedule(self._h, None), "ebm_set_column_schedule")
            return
        if len(forcings) != self.ncol:
            raise ValueError(f"expected {self.ncol} Forcing objects, got {len(forcings)}")
        a = np.array([schedule_words(f) for f in forcings], dtype=np.float64)
        check(self.lib.ebm_set_column_schedule(self._h, dptr(a)), "ebm_set_column_schedule")

    def set_step_clock(self, step: int):
        check(self.lib.ebm_set_step_clock(self._h, int(step)), "ebm_set_step_clock")

    def set_time_table(self, t_in_year):
        """t_in_year = st.t; uploads cos(2.0*pi*t_i)."""
        tab = np.array([cos2pit(float(t)) for t in t_in_year], dtype=np.float64)
        self.nt = len(tab)
        self.ttab = tab
        check(self.lib.ebm_set_time_table(self._h, self.nt, dptr(tab)), "ebm_set_time_table")

    # -- stepping ---------------------------------------------------------------------
    def step(self, ct: float, ct_next: float, f: float, write_diag: bool = True):
        check(self.lib.ebm_step(self._h, ct, ct_next, f, int(write_diag)), "ebm_step")

    def run(self, first_step: int, nsteps: int, f_steps=None, diag_last: bool = True,
            steps_per_launch: int = 1):
        """``nsteps`` steps from global step ``first_step``.  ``steps_per_launch`` = K > 1 fuses K
        consecutive steps into one launch (ebm_run_fused; bit-identical results, no per-step
        output in between)."""
        a = None if f_steps is None else as_f64(f_steps, (nsteps,))
        if steps_per_launch > 1:
            check(self.lib.ebm_run_fused(self._h, int(first_step), int(nsteps), dptr(a), int(diag_last),
                                         int(steps_per_launch)), "ebm_run_fused")
        else:
            check(self.lib.ebm_run(self._h, int(first_step), int(nsteps), dptr(a), int(diag_last)),
                  "ebm_run")

    def integrate(self, nt, dur, f_steps, lastonly, winter_inx, summer_inx, names,
                  want_raw=True, want_seasonal=True, want_avg=True, out=None):
        """ebm_integrate: returns dict(raw, winter, summer, avg), each [nvars, n, ncol, nlat].
        Outputs that are not wanted are None; with neither raw nor avg the diagnostic fields are
        only written on the steps whose snapshot is taken.  ``out``: a dict returned by an earlier call of the
        same shape, whose arrays are filled again instead of allocating new ones (a caller that integrates year
        after year into the same buffers)."""
        nv = len(names)
        fields = (C.c_int * nv)(*[FIELD[n] for n in names])
        nraw = nt if lastonly else nt * dur
        f = None if f_steps is None else as_f64(f_steps, (nt * dur,))
        shape = (nv, dur, self.ncol, self.nlat)

        def reuse(key, wanted, shp):
            a = None if out is None else out.get(key)
            if not wanted:
                return None
            if a is not None and (a.shape != shp or a.dtype != np.float64 or not a.flags.c_contiguous):
                raise ValueError(f"out[{key!r}] must be a C-contiguous float64 array of shape {shp}")
            return a
        raw = reuse("raw", want_raw, (nv, nraw, self.ncol, self.nlat))
        if want_raw and raw is None:
            raw = np.empty((nv, nraw, self.ncol, self.nlat))
        def mk(key, wanted):
            a = reuse(key, wanted, shape)
            return a if (a is not None or not wanted) else np.full(shape, np.nan)
        winter, summer, avg = mk("winter", want_seasonal), mk("summer", want_seasonal), mk("avg", want_avg)
        check(self.lib.ebm_integrate(self._h, nt, dur, dptr(f), int(lastonly), int(winter_inx),
                                     int(summer_inx), nv, fields, dptr(raw), dptr(winter),
                                     dptr(summer), dptr(avg)), "ebm_integrate")
        return dict(raw=raw, winter=winter, summer=summer, avg=avg)

    def integrate_hemispheric(self, nt, dur, f_steps, winter_inx, summer_inx, names):
        """ebm_integrate_hemispheric: per variable, year and column the hemispheric mean (reference
        src/utilities.jl:397-403) of the winter snapshot, the summer snapshot and the annual mean, reduced
        on the device — dict(winter, summer, avg), each [nvars, dur, ncol].  The data of the reference's
        hysteresis plot (src/plot.jl:173-225) for every column, without the fields crossing the bus."""
        nv = len(names)
        fields = (C.c_int * nv)(*[FIELD[n] for n in names])
        f = None if f_steps is None else as_f64(f_steps, (nt * dur,))
        out = {k: np.full((nv, dur, self.ncol), np.nan) for k in ("winter", "summer", "avg")}
        check(self.lib.ebm_integrate_hemispheric(self._h, nt, dur, dptr(f), int(winter_inx), int(summer_inx), nv, fields,
                                                 dptr(out["winter"]), dptr(out["summer"]), dptr(out["avg"])),
              "ebm_integrate_hemispheric")
        return out

    def sync(self):
        check(self.lib.ebm_sync(self._h), "ebm_sync")

    # -- measurement ------------------------------------------------------------------
    def counters(self) -> dict:
        c = (C.c_longlong * 4)()
        check(self.lib.ebm_get_counters(self._h, c), "ebm_get_counters")
        return dict(steps=c[0], solves=c[1], cap_hits=c[2], launches=c[3])

    def reset_counters(self):
        check(self.lib.ebm_reset_counters(self._h), "ebm_reset_counters")

    def timer_start(self):
        check(self.lib.ebm_timer_start(self._h), "ebm_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(self.lib.ebm_timer_stop(self._h, C.byref(ms)), "ebm_timer_stop")
        return float(ms.value)

    def launch_info(self) -> dict:
        info = (C.c_int * 4)()
        check(self.lib.ebm_launch_info(self._h, info), "ebm_launch_info")
        return dict(threads=info[0], cells_per_thread=info[1], lds_bytes=info[2], workgroups=info[3])
