"""Host-side mirror of the reference's operator surface for the time-stepping path.

Same names, argument meaning and error behaviour as the Julia package
(reference src/infrastructure.jl; exports src/EnergyBalanceModel.jl:79-82):

    Collection, SpaceTime, Forcing, Solutions, default_parval, miz_paramset,
    classic_paramset, default_parameters, step_ (Julia: step!), integrate

All numerics run on the GPU through the C ABI (include/ebm_hip.h); this module only builds the
inputs (grid, forcing schedule, parameters), owns the handles, and lays results out like the
reference's ``Solutions``.  Julia symbols become strings: ``:MIZ`` -> "MIZ", ``:Classic`` ->
"Classic"; ``SpaceTime{sin}`` -> ``SpaceTime("sin", ...)``.
"""
from __future__ import annotations

import math
import os
import warnings
from fractions import Fraction

import numpy as np

from .engine import Engine, cos2pit, param_vector


class Collection(dict):
    """``Collection{V}``: a Dict{Symbol,V} with dot access (src/infrastructure.jl:39-49)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise KeyError(key) from None       # Julia: KeyError on a missing property

    def __setattr__(self, key, val):
        self[key] = val

    def propertynames(self):
        return set(self.keys())


# default parameter values, src/infrastructure.jl:407-433
default_parval = Collection(
    D=0.6, A=193.0, B=2.1, cw=9.8, S0=420.0, S1=338.0, S2=240.0, a0=0.7, a2=0.1, ai=0.4,
    Fb=4.0, k=2.0, Lf=9.5, F=0.0, cg=0.01 * 9.8, tau=1e-5, Tm=0.0, m1=1.6e-6 * 31536000,
    m2=1.36, alpha=0.66, rl=0.5, Dmin=1.0, Dmax=156.0, hmin=0.1, kappa=0.01 * 31536000,
)
miz_paramset = frozenset((
    "D", "A", "B", "cw", "S0", "S1", "S2", "a0", "a2", "ai", "Fb", "k", "Lf", "Tm", "m1", "m2",
    "alpha", "rl", "Dmin", "Dmax", "hmin", "kappa"))
classic_paramset = frozenset((
    "D", "A", "B", "cw", "S0", "S1", "S2", "a0", "a2", "ai", "Fb", "k", "Lf", "F", "cg", "tau"))


def default_parameters(model) -> Collection:
    """src/infrastructure.jl:447-474: a model symbol (anything but "MIZ" gives the classic set)
    or an explicit parameter-name set."""
    if isinstance(model, (set, frozenset)):
        keys = model
    else:
        keys = miz_paramset if model == "MIZ" else classic_paramset
    return Collection({k: default_parval[k] for k in keys})


def _exact_range(start: float, step: float, n: int) -> np.ndarray:
    """Elements of a Julia float range (TwicePrecision: start + i*step evaluated essentially
    exactly, rounded once)."""
    a, s = Fraction(start), Fraction(step)
    return np.array([float(a + i * s) for i in range(n)], dtype=np.float64)


class SpaceTime:
    """``SpaceTime{F}(nx, nt, dur; winter, summer)`` — src/infrastructure.jl:109-141.

    ``kind`` plays the role of the type parameter F: "identity" (u in (0,1), x = u) or "sin"
    (u in (0, pi/2), x = sin(u)).  A callable F with an explicit ``urange`` is accepted too
    (any such grid uses the non-uniform diffusion stencil, like the reference).
    """

    def __init__(self, kind="identity", nx=None, nt=None, dur=None, *, winter=0.26125,
                 summer=0.77375, urange=None):
        if nx is None or nt is None or dur is None:
            raise TypeError("SpaceTime(kind, nx, nt, dur)")
        self.F = kind
        self.nx, self.nt, self.dur = int(nx), int(nt), int(dur)
        if kind == "identity" and urange is None:
            # 1/(2nx) : 1/nx : 1-1/(2nx) lifts to exact rationals in Julia's range constructor
            self.u = np.array([float(Fraction(2 * i + 1, 2 * self.nx)) for i in range(self.nx)])
            self.x = self.u.copy()
        else:
            if urange is None:
                if kind != "sin":
                    raise ValueError("urange is required for a custom grid function")
                urange = (0.0, math.pi / 2.0)
            fn = math.sin if kind == "sin" else kind
            dx = (urange[1] - urange[0]) / self.nx
            self.u = _exact_range(urange[0] + dx / 2.0, dx, self.nx)
            self.x = np.array([fn(float(v)) for v in self.u], dtype=np.float64)
        self.dt = 1.0 / self.nt
        self.t = np.array([float(Fraction(2 * i + 1, 2 * self.nt)) for i in range(self.nt)])
        self.T = np.array([float(Fraction(2 * i + 1, 2 * self.nt)) for i in range(self.nt * self.dur)])
        # round(Int, x) is half-to-even in Julia, as is Python's round()
        self.winter = Collection(t=winter, inx=int(round(self.nt * winter)))
        self.summer = Collection(t=summer, inx=int(round(self.nt * summer)))

    @property
    def grid_kind(self) -> str:
        return "identity" if self.F == "identity" else "nonuniform"

    def __repr__(self):
        return f"SpaceTime{{{self.F}}}({self.nx}, {self.nt}, {self.dur})"


class Forcing:
    """``Forcing(base)`` / ``Forcing(base, peak, cool, holdyrs, rates)`` —
    src/infrastructure.jl:208-241; evaluation :294-307."""

    def __init__(self, base, peak=None, cool=None, holdyrs=None, rates=None):
        if peak is None:
            self.constant = True
            self.base = self.peak = self.cool = float(base)
            self.holdyrs, self.rates, self.domain = (0, 0), (0.0, 0.0), (0, 0, 0, 0, 0)
            return
        self.constant = False
        dom = [0, 0, 0, 0, 0]
        for i in range(1, 5):
            dom[i] += holdyrs[0]
        warming = (peak - base) / rates[0]
        if not (rates[0] > 0 and float(warming).is_integer()):
            raise ValueError(f"Warming time must be positive integer. Got {warming} y.")
        for i in range(2, 5):
            dom[i] += int(warming)
        for i in range(3, 5):
            dom[i] += holdyrs[1]
        cooling = (cool - peak) / rates[1]
        if not (rates[1] < 0 and float(cooling).is_integer()):
            raise ValueError(f"Cooling time must be positive integer. Got {cooling} y.")
        dom[4] += int(cooling)
        self.base, self.peak, self.cool = float(base), float(peak), float(cool)
        self.holdyrs, self.rates, self.domain = tuple(holdyrs), tuple(rates), tuple(dom)

    def __call__(self, T: float) -> float:
        if self.constant:
            return self.base
        d = self.domain
        if T < d[1]:
            return self.base
        elif T < d[2]:
            return self.base + self.rates[0] * (T - d[1])
        elif T < d[3]:
            return self.peak
        elif T < d[4]:
            return self.peak + self.rates[1] * (T - d[3])
        return self.cool

    def __repr__(self):
        return f"Forcing({self.base})" if self.constant else f"Forcing({self.base} ↗ {self.peak} ↘ {self.cool})"


class Solutions:
    """``Solutions{F,C}`` — src/infrastructure.jl:333-383.

    ``raw.E[ti]`` is the enthalpy vector at stored time ``ts[ti]``;
    ``seasonal.avg.T[y]`` the annual-mean temperature of year ``y`` (0-based here).
    """

    def __init__(self, st, forcing, par, init, varnames, lastonly=True, debug=None):
        self.spacetime, self.forcing, self.parameters, self.initconds = st, forcing, par, init
        self.lastonly, self.debug = lastonly, debug
        if lastonly:
            self.ts = np.array([float(Fraction(st.dur - 1) + Fraction(2 * i + 1, 2 * st.nt))
                                for i in range(st.nt)])
        else:
            self.ts = st.T.copy()
        self.raw = Collection({v: None for v in varnames})
        self.seasonal = Collection(
            winter=Collection({v: None for v in varnames}),
            summer=Collection({v: None for v in varnames}),
            avg=Collection({v: None for v in varnames}),
        )

    def __repr__(self):
        return (f"Solutions{{{self.spacetime.F}, {self.forcing.constant}}} with {len(self.raw)} "
                f"solution variables on {self.spacetime.nx} latitudinal gridboxes and "
                f"{len(self.ts)} timesteps")


MIZ_SOLVARS = ("E", "T", "h", "Ei", "Ew", "Ti", "Tw", "D", "phi", "n")
CLASSIC_SOLVARS = ("E", "T", "h")
_INIT_VARS = {"MIZ": ("Ei", "Ew", "h", "D", "phi"), "Classic": ("E", "Tg"), "MIZ_IMEX": ("Ei", "Ew", "h", "D", "phi")}


def _is_miz(model) -> bool:
    """"MIZ" or "MIZ_IMEX" — the latter an EXTENSION with no counterpart in the reference: the MIZ model
    with its meridional diffusion treated linearly implicitly (include/ebm_hip.h, EBM_MODEL_MIZ_IMEX), for
    time steps beyond the explicit limit.  Pass ``default_parameters("MIZ")`` with it."""
    return model in ("MIZ", "MIZ_IMEX")


def _check_model(model):
    if model not in ("MIZ", "Classic", "MIZ_IMEX"):
        # Julia: MethodError — no step!(::Val{model}, ...) method (src/infrastructure.jl:594)
        raise ValueError(f"no step! method for model {model!r}: expected 'MIZ' or 'Classic'")


def _new_engine(model, st, par, ncol=1, device=0) -> Engine:
    """The reference's own shapes are ONE meridian: such a run is latency-bound on a handful of waves, and two
    latitudes per thread (ebm_options.cells_per_thread = 2, meridians of up to 1536 cells) put twice as many of
    them to work.  The choice is made HERE, explicitly, and never by the library from the number of columns: the
    partition of the tridiagonal solves — hence their rounding — depends on it (include/ebm_hip.h)."""
    cells = 2 if (ncol == 1 and st.nx <= 1536 and model != "MIZ_IMEX" and os.environ.get("EBM_CELLS_PER_THREAD") is None) else None
    return Engine(model, st.grid_kind, st.x, param_vector(par, default_parval), st.dt, ncol, device,
                  cells_per_thread=cells)


def classic_time_index(t: float, dt: float, nt: int) -> int:
    """src/classic.jl:45: round(Int, mod1((t + dt/2)*nt, nt)), 1-based."""
    y = (t + dt / 2.0) * nt
    m = math.fmod(y, nt)
    if m < 0:
        m += nt
    if m == 0.0:
        m = float(nt)
    return int(round(m))


# The reference keeps the T0 warm start (and the diffusion / statics caches) in module-level
# closures that survive between step! calls (src/miz.jl:47, src/classic.jl:7-16).  The mirror of
# that for direct step_ calls is a module-level cache of engines keyed by everything the cached
# state depends on.
_step_engines: dict = {}


def reset_step_state():
    """Drop the hidden per-(model, grid, parameters) state kept between step_ calls."""
    for e in _step_engines.values():
        e.close()
    _step_engines.clear()


def step_(model, t, f, vars, st, par, *, debug=None, verbose=False, device=0):
    """``step!(Val(model), t, f, vars, st, par; debug, verbose)`` — one forward step.

    MIZ (src/miz.jl:150-196): reads Ei, Ew, h, D, phi from ``vars``, rebinds those and sets
    Tw, Ti, n, E, T.  Classic (src/classic.jl:37-71): reads E, Tg; rebinds them and sets T, h.
    Returns ``vars``.  The T0 warm start persists between calls like the reference's hidden
    state.  ``debug`` expressions cannot cross the C ABI and are rejected.
    """
    _check_model(model)
    if debug is not None:
        raise NotImplementedError("debug expressions are evaluated inside the reference's step! "
                                  "(src/miz.jl:188-191); they cannot cross the C ABI")
    pv = param_vector(par, default_parval)
    key = (model, st.grid_kind, st.nx, st.x.tobytes(), st.dt, pv.tobytes(), device)
    eng = _step_engines.get(key)
    if eng is None:
        eng = _step_engines[key] = _new_engine(model, st, par, 1, device)
    for k in _INIT_VARS[model]:
        eng.set_field(k, np.asarray(vars[k], dtype=np.float64).reshape(1, st.nx))
    before = eng.counters()["cap_hits"] if (verbose and _is_miz(model)) else 0
    if _is_miz(model):
        eng.step(cos2pit(t), 0.0, float(f), True)
    else:
        i = classic_time_index(t, st.dt, st.nt)                  # column of S, 1-based
        eng.step(cos2pit(float(st.t[i - 1])), cos2pit(float(st.t[i % st.nt])), float(f), True)
    names = MIZ_SOLVARS if _is_miz(model) else ("E", "Tg", "T", "h")
    for k in names:
        vars[k] = eng.get_field(k)[0]
    if verbose and _is_miz(model) and eng.counters()["cap_hits"] > before:
        warnings.warn(f"Solving for T0 failed at t={t}.")        # src/miz.jl:61-63
    return vars


def integrate(model, st, forcing, par, init, *, lastonly=True, debug=None, verbose=False,
              device=0) -> Solutions:
    """``integrate(model, st, forcing, par, init; lastonly, debug, verbose)`` —
    src/infrastructure.jl:615-636 with savesol! (:549-591) running on the device.

    For "MIZ" ``init`` must contain Ei, Ew, h, D, phi; for "Classic" E and Tg.  (At the
    reference commit ``integrate(:Classic, ...)`` throws on the ``verbose`` keyword — SURVEY
    F7; here it works and stores E, T, h as :621 prescribes.)"""
    _check_model(model)
    if debug is not None:
        raise NotImplementedError("debug expressions cannot cross the C ABI (src/miz.jl:188-191)")
    names = MIZ_SOLVARS if _is_miz(model) else CLASSIC_SOLVARS
    sols = Solutions(st, forcing, par, init, names, lastonly, debug)
    f_steps = np.array([forcing(float(T)) for T in st.T], dtype=np.float64)
    with _new_engine(model, st, par, 1, device) as eng:
        for k in _INIT_VARS[model]:
            eng.set_field(k, np.asarray(init[k], dtype=np.float64).reshape(1, st.nx))
        eng.set_time_table(st.t)
        out = eng.integrate(st.nt, st.dur, f_steps, lastonly, st.winter.inx, st.summer.inx, names)
        cnt = eng.counters()
    for vi, v in enumerate(names):
        sols.raw[v] = out["raw"][vi, :, 0, :]
        sols.seasonal.winter[v] = out["winter"][vi, :, 0, :]
        sols.seasonal.summer[v] = out["summer"][vi, :, 0, :]
        sols.seasonal.avg[v] = out["avg"][vi, :, 0, :]
    sols.counters = cnt
    if verbose and cnt["cap_hits"] > 0:
        warnings.warn(f"Solving for T0 failed at {cnt['cap_hits']} time steps.")
    return sols
