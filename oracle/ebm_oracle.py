"""CPU oracle (NumPy fp64) for the energy-balance time-stepping hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and
only as the checker.  The shipped path is the HIP library behind ``include/ebm_hip.h``.

PARITY UNPINNED.  The reference (waylonwh/EnergyBalanceModel.jl v0.3.0) is Julia; no Julia
toolchain exists in this pipeline and the reference's single golden file
(``test/solution_1year.jld2``) is absent from the mount, so this restatement cannot be checked
against outputs of the reference itself.  It is pinned only by (i) the known answers the
reference's own docstrings state (grid end points, forcing breakpoints, parameter counts) and
(ii) closed forms derived from the reference text (tests/test_oracle.py).

Every function cites the reference lines it restates (paths relative to /root/reference).
Operation ORDER follows the reference expression by expression (Julia never contracts a*b+c
into an FMA, ``x^2`` is ``x*x``, ``a*b/c`` is ``(a*b)/c``, ``min``/``clamp`` propagate NaN,
``Bool*Float`` is a strong zero).  The one place that is NOT a transcription is the nonlinear
solve for the ice surface temperature T0 (src/miz.jl:55-60): the reference calls the
third-party NonlinearSolve.TrustRegion (abstol 1e-8, reltol 1e-6); here the piecewise-linear
system is solved exactly by an active-set (semismooth Newton) iteration with a Thomas solve,
which lands on the root the reference's iteration approximates.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

# --------------------------------------------------------------------------------------
# parameters                                                   src/infrastructure.jl:407-474
# --------------------------------------------------------------------------------------
PARAM_ORDER = (
    "D", "A", "B", "cw", "S0", "S1", "S2", "a0", "a2", "ai", "Fb", "k", "Lf", "F", "cg",
    "tau", "Tm", "m1", "m2", "alpha", "rl", "Dmin", "Dmax", "hmin", "kappa",
)

default_parval = {
    "D": 0.6, "A": 193.0, "B": 2.1, "cw": 9.8, "S0": 420.0, "S1": 338.0, "S2": 240.0,
    "a0": 0.7, "a2": 0.1, "ai": 0.4, "Fb": 4.0, "k": 2.0, "Lf": 9.5, "F": 0.0,
    "cg": 0.01 * 9.8, "tau": 1e-5, "Tm": 0.0, "m1": 1.6e-6 * 31536000, "m2": 1.36,
    "alpha": 0.66, "rl": 0.5, "Dmin": 1.0, "Dmax": 156.0, "hmin": 0.1,
    "kappa": 0.01 * 31536000,
}
miz_paramset = (
    "D", "A", "B", "cw", "S0", "S1", "S2", "a0", "a2", "ai", "Fb", "k", "Lf", "Tm", "m1",
    "m2", "alpha", "rl", "Dmin", "Dmax", "hmin", "kappa",
)
classic_paramset = (
    "D", "A", "B", "cw", "S0", "S1", "S2", "a0", "a2", "ai", "Fb", "k", "Lf", "F", "cg", "tau",
)


def default_parameters(model: str) -> dict:
    """src/infrastructure.jl:447-474 (anything that is not :MIZ gets the classic set)."""
    keys = miz_paramset if model == "MIZ" else classic_paramset
    return {k: default_parval[k] for k in keys}


# --------------------------------------------------------------------------------------
# grid                                                         src/infrastructure.jl:109-141
# --------------------------------------------------------------------------------------
def _exact_range(start: float, step: float, n: int) -> np.ndarray:
    """Elements of a Julia float range: start + i*step evaluated (nearly) exactly in
    TwicePrecision and rounded once.  Restated with exact rationals."""
    a, s = Fraction(start), Fraction(step)
    return np.array([float(a + i * s) for i in range(n)], dtype=np.float64)


def _round_half_even(v: float) -> int:
    return int(round(v))  # Python's round() is half-to-even, like Julia's round(Int, x)


class SpaceTime:
    """src/infrastructure.jl:109-141.  kind: "identity" (u in (0,1)) or "sin" (u in (0,pi/2))."""

    def __init__(self, kind, nx, nt, dur, winter=0.26125, summer=0.77375):
        self.kind, self.nx, self.nt, self.dur = kind, int(nx), int(nt), int(dur)
        if kind == "identity":
            # 1/(2nx) : 1/nx : 1-1/(2nx) lifts to rationals in Julia's float-range constructor
            self.u = np.array([float(Fraction(2 * i + 1, 2 * nx)) for i in range(nx)])
            self.x = self.u.copy()
        elif kind == "sin":
            dx = (math.pi / 2.0 - 0.0) / nx
            self.u = _exact_range(0.0 + dx / 2.0, dx, nx)
            self.x = np.array([math.sin(v) for v in self.u])
        else:
            raise ValueError(kind)
        self.dt = 1.0 / nt
        # range(dt/2, 1-dt/2, nt) and dt/2:dt:dur-dt/2 both land on (2i-1)/(2nt)
        self.t = np.array([float(Fraction(2 * i + 1, 2 * nt)) for i in range(nt)])
        self.T = np.array([float(Fraction(2 * i + 1, 2 * nt)) for i in range(nt * dur)])
        self.winter_t, self.summer_t = winter, summer
        self.winter_inx = _round_half_even(nt * winter)  # 1-based index inside the year
        self.summer_inx = _round_half_even(nt * summer)


class Forcing:
    """src/infrastructure.jl:208-241 (construction), :294-307 (evaluation)."""

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


# --------------------------------------------------------------------------------------
# Julia IEEE helpers
# --------------------------------------------------------------------------------------
def jl_min(x, y):
    """Julia's min(::Float64, ::Float64): NaN-propagating, -0.0 < +0.0."""
    x = np.asarray(x, dtype=np.float64)
    y = np.broadcast_to(np.asarray(y, dtype=np.float64), x.shape)
    diff = x - y
    out = np.where(np.signbit(diff), x, y)
    return np.where(np.isnan(x) | np.isnan(y), diff, out)


def jl_clamp(x, lo, hi):
    """Julia's clamp: ifelse(x > hi, hi, ifelse(x < lo, lo, x)); NaN stays NaN."""
    return np.where(x > hi, hi, np.where(x < lo, lo, x))


def bool_mul(x, b):
    """Julia's *(x::Float64, b::Bool) = ifelse(b, x, copysign(0.0, x)) (strong zero)."""
    return np.where(b, x, np.copysign(0.0, x))


def condset(to, val, mask):
    """src/utilities.jl:406-412 with the predicate already applied to the reference vector."""
    out = to.copy()
    out[mask] = val
    return out


def zeroref(v, ref):
    """src/utilities.jl:415: entries where ref is zero (iszero(-0.0) is true) become 0.0."""
    return condset(v, 0.0, ref == 0.0)


# --------------------------------------------------------------------------------------
# diffusion operator                                           src/infrastructure.jl:476-533
# --------------------------------------------------------------------------------------
class DiffusionGeometry:
    """Per-latitude constants of the meridional diffusion operator for one grid.

    kind == "identity": the three diagonals of ``par.D * get_diffop(nx)``
        (src/infrastructure.jl:480-497).
    kind == "sin" (any non-uniform grid): the cached vectors of src/infrastructure.jl:509-518.
    ``lo, di, up`` are the same operator as plain tridiagonal coefficients
    (Dif_k(v) = lo_k v_{k-1} + di_k v_k + up_k v_{k+1}); they are used only to assemble the
    Jacobian of the T0 system, never to evaluate the reference's expressions.
    """

    def __init__(self, kind, x, D):
        nx = len(x)
        self.kind, self.nx, self.D = kind, nx, D
        if kind == "identity":
            dx = 1.0 / nx
            xb = np.arange(1, nx, dtype=np.float64) / nx      # dx:dx:1-dx
            lam = (1.0 - xb * xb) / (dx * dx)
            sub = np.zeros(nx)
            sup = np.zeros(nx)
            sub[1:] = lam                                      # A[k,k-1] = lambda_{k-1}
            sup[:-1] = lam                                     # A[k,k+1] = lambda_k
            l1 = np.concatenate(([0.0], -lam))
            l2 = np.concatenate((-lam, [0.0]))
            l3 = (-l1) - l2
            diag = -l3
            self.sub, self.diag, self.sup = D * sub, D * diag, D * sup   # par.D * diffop
            self.lo, self.di, self.up = self.sub, self.diag, self.sup
        else:
            xe = np.concatenate(([-x[0]], x, [2.0 - x[-1]]))
            diffx = xe[1:] - xe[:-1]                           # length nx+1
            xxph = (xe[2:] + xe[1:-1]) / 2.0
            xxmh = (xe[1:-1] + xe[:-2]) / 2.0
            self.mph = 1.0 - xxph * xxph
            self.mmh = 1.0 - xxmh * xxmh
            self.w = xxph - xxmh
            self.dxp = diffx[1:].copy()                        # diffx[i]   for cell k
            self.dxm = diffx[:-1].copy()                       # diffx[i-1] for cell k
            up = D * self.mph / (self.dxp * self.w)
            lo = D * self.mmh / (self.dxm * self.w)
            up[-1] = 0.0                                       # diffT[end] = 0 (pole)
            lo[0] = 0.0                                        # diffT[1]   = 0 (equator)
            self.lo, self.up, self.di = lo, up, -(lo + up)

    def add(self, base, temp):
        """diffusion!(base, temp, st, par): returns base + D*d/dx[(1-x^2) d temp/dx]."""
        nx = self.nx
        if self.kind == "identity":
            # CSC SpMV of (D*L): row k accumulates columns k-1, k, k+1 in that order from 0.0
            y = np.zeros(nx)
            y[1:] = y[1:] + self.sub[1:] * temp[:-1]
            y = y + self.diag * temp
            y[:-1] = y[:-1] + self.sup[:-1] * temp[1:]
            return base + y
        dTp = np.zeros(nx)
        dTm = np.zeros(nx)
        d = temp[1:] - temp[:-1]
        dTp[:-1] = d
        dTm[1:] = d
        return base + (self.D * ((self.mph * dTp) / self.dxp - (self.mmh * dTm) / self.dxm)) / self.w


# --------------------------------------------------------------------------------------
# MIZ model                                                    src/miz.jl
# --------------------------------------------------------------------------------------
def insolation(x, ct, par):
    """S(x,t) of src/miz.jl:11,14 with ct = cos(2.0*pi*t)."""
    return par["S0"] - par["S1"] * x * ct - par["S2"] * (x * x)


def solar_add(base, x, ct, ice, par):
    """solar! src/miz.jl:9-14."""
    S = insolation(x, ct, par)
    if ice:
        return base + par["ai"] * S
    return base + (par["a0"] - par["a2"] * (x * x)) * S


def Tbar(Ti, Tw, phi):
    """src/miz.jl:21-26: Ti .*= phi; Ti += (1-phi)*Tw."""
    return Ti * phi + (1.0 - phi) * Tw


def water_temp(Ew, phi, par):
    """src/miz.jl:30."""
    return par["Tm"] + Ew / ((1.0 - phi) * par["cw"])


def ice_temp(T0, par):
    """src/miz.jl:31."""
    return jl_min(T0, par["Tm"])


def T0eq(T0, x, ct, hp, Tw, phi, f, geom, par):
    """Residual of the ice-surface energy balance, src/miz.jl:33-45."""
    vec = par["k"] * (par["Tm"] - T0) / hp
    vec = solar_add(vec, x, ct, True, par)
    vec = vec + ((-par["A"]) - par["B"] * (T0 - par["Tm"]))
    vec = geom.add(vec, Tbar(ice_temp(T0, par), Tw, phi))
    vec = vec + f
    return vec


def thomas(a, b, c, d):
    """Plain Thomas algorithm (no pivoting), reciprocal-free divisions, fixed operation order:
        cp_0 = c_0/b_0, dp_0 = d_0/b_0
        den = b_i - a_i*cp_{i-1}; cp_i = c_i/den; dp_i = (d_i - a_i*dp_{i-1})/den
        x_n = dp_n; x_i = dp_i - cp_i*x_{i+1}
    oracle/ebm_oracle.c follows the same order so the two restatements agree bit for bit."""
    n = len(b)
    a, b, c, d = (v.tolist() for v in (a, b, c, d))
    cp = [0.0] * n
    dp = [0.0] * n
    cp[0] = c[0] / b[0]
    dp[0] = d[0] / b[0]
    for i in range(1, n):
        den = b[i] - a[i] * cp[i - 1]
        cp[i] = c[i] / den
        dp[i] = (d[i] - a[i] * dp[i - 1]) / den
    xs = [0.0] * n
    xs[n - 1] = dp[n - 1]
    for i in range(n - 2, -1, -1):
        xs[i] = dp[i] - cp[i] * xs[i + 1]
    return np.array(xs)


MAX_NEWTON = 1000   # NonlinearSolve's default maxiters (src/miz.jl:55-60 passes none)


def solve_T0(T0_warm, x, ct, hp, Tw, phi, f, geom, par):
    """Root of T0eq (src/miz.jl:33-45) by active-set Newton.

    With v = T0 - Tm the system reads
        -(k/hp + B) v + Dif(phi*min(v,0)) = -(ai*S - A + Dif((1-phi)(Tw-Tm)) + f)
    (Dif of the constant Tm vanishes: zero-flux ends).  On a fixed active set s = [v<0] it is
    linear and tridiagonal.  Start from the warm start's active set (src/miz.jl:47,52-54,64),
    solve, re-evaluate the set, stop when it repeats.  Returns (T0, n_solves, converged).
    """
    Tm = par["Tm"]
    dd = par["k"] / hp + par["B"]
    r = (1.0 - phi) * (Tw - Tm)
    rp = np.concatenate((r[1:], [0.0]))
    rm = np.concatenate(([0.0], r[:-1]))
    rhs = par["ai"] * insolation(x, ct, par) - par["A"] + (geom.lo * rm + geom.di * r + geom.up * rp) + f
    s = T0_warm < Tm
    v = None
    for it in range(1, MAX_NEWTON + 1):
        g = np.where(s, phi, 0.0)
        gp = np.concatenate((g[1:], [0.0]))
        gm = np.concatenate(([0.0], g[:-1]))
        a = geom.lo * gm
        c = geom.up * gp
        b = geom.di * g - dd
        v = thomas(a, b, c, -rhs)
        s_new = v < 0.0
        if np.array_equal(s_new, s):
            return v + Tm, it, True
        s = s_new
    return v + Tm, MAX_NEWTON, False


def solveTi(T0_warm, x, ct, h, Tw, phi, f, geom, par):
    """src/miz.jl:47-68.  Returns (Ti, T0) — T0 is the new warm start (hidden state :64)."""
    hp = condset(h, par["hmin"], h == 0.0)
    T0, nit, ok = solve_T0(T0_warm, x, ct, hp, Tw, phi, f, geom, par)
    Ti = ice_temp(T0, par)
    Ti = zeroref(Ti, h)
    return Ti, T0, nit, ok


def wlat(Tw, par):
    """src/miz.jl:71 — precedence as written: m1*(Tw - Tm^m2)."""
    return par["m1"] * (Tw - par["Tm"] ** par["m2"])


def concentration(Ei, h, par):
    """src/miz.jl:74-80."""
    phi = -Ei / (par["Lf"] * h)
    phi = zeroref(phi, h)
    return condset(phi, 1.0, phi > 1.0)


def num(D, phi, par):
    """src/miz.jl:83-87."""
    n = phi / (par["alpha"] * (D * D))
    return zeroref(n, D)


def area_lead(D, phi, n, par):
    """src/miz.jl:90-93."""
    Dr = D + 2.0 * par["rl"]
    ring = par["alpha"] * n * (Dr * Dr - D * D)
    return jl_min(ring, 1.0 - phi)


def vert_flux(x, ct, ice, Ti, Tw, phi, f, geom, par):
    """src/miz.jl:96-101."""
    tb = Tbar(Ti, Tw, phi)
    L = par["A"] + par["B"] * (tb - par["Tm"])
    sol = solar_add(np.zeros_like(x), x, ct, ice, par)
    dif = geom.add(np.zeros_like(x), tb)
    return sol - L + dif + par["Fb"] + f


def lat_flux(h, D, Tw, phi, par):
    """src/miz.jl:103-107."""
    Flat = phi * h * par["Lf"] * wlat(Tw, par) * math.pi / (par["alpha"] * D)
    return zeroref(Flat, D)


def redistributeE(rEi, rEw):
    """src/miz.jl:109-117."""
    cEi = jl_clamp(rEi, -np.inf, 0.0)
    cEw = jl_clamp(rEw, 0.0, np.inf)
    psiEidt = rEi - cEi
    psiEwdt = rEw - cEw
    return cEi + psiEwdt, cEw + psiEidt, psiEidt, psiEwdt


def split_psiEw(psiEw, phi, Al):
    """src/miz.jl:120-125."""
    Ql = Al / (1.0 - phi) * psiEw
    Ql = condset(Ql, 0.0, phi == 1.0)
    return Ql, psiEw - Ql


def psinplus(Qp, par):
    """src/miz.jl:127."""
    return -Qp / (par["Lf"] * par["alpha"] * (par["Dmin"] * par["Dmin"]) * par["hmin"])


def average(f, fn, n, dn):
    """src/miz.jl:129-134."""
    total = n + dn
    avgd = (n * f + dn * fn) / total
    return zeroref(avgd, total)


def D_t(h, D, Tw, phi, Ql, par):
    """src/miz.jl:140-146."""
    lat_melt = (-math.pi / 2.0 * par["alpha"]) * wlat(Tw, par)
    lat_grow = -D / (2.0 * par["Lf"] * h * phi) * Ql
    weld = par["kappa"] * par["alpha"] / 4.0 * phi * (D * D * D)
    lat_grow = zeroref(lat_grow, h)
    return lat_melt + lat_grow + weld


def implicit_diffusion_correction(dE, dt, geom, par):
    """EXTENSION, not in the reference — "parity unpinned" by construction.  Defined in
    include/ebm_hip.h (EBM_MODEL_MIZ_IMEX); this is the checker's restatement of that definition, the C
    oracle restates it again, and the HIP kernel is compared with both.

    The reference's step treats the meridional diffusion D d/dx[(1-x^2) dTbar/dx] inside the vertical
    fluxes explicitly (src/miz.jl:96-101), which limits dt to cw*dx^2/(2D): more than 800,000 steps per
    year at 4096 latitudes.  Linearly implicit correction: let dE = dt*(phi*Fvi + (1-phi)*Fvw) be the
    explicit increment of a cell's TOTAL enthalpy (the lateral flux cancels in the sum) and assume the
    surface temperature follows it with the water's heat capacity, u = dE_new/cw — an upper bound of the
    true response (heat that melts or grows ice changes no temperature), which is what makes the scheme
    stable.  Then dE_new = dE + dt*Dif(u), i.e.

        (I - (dt/cw)*Dif) dE_new = dE          Dif = the operator's plain tridiagonal (lo, di, up)

    one tridiagonal solve per meridian with a state-independent matrix, and the step continues with the
    diffusion term of BOTH vertical fluxes corrected by (dE_new - dE)/dt.  Backward Euler on the diffusion
    of the increment: no explicit limit from the grid spacing, first-order consistent with the reference's
    step (the correction vanishes as dt -> 0)."""
    theta = dt / par["cw"]
    a = -(theta * geom.lo)
    c = -(theta * geom.up)
    b = 1.0 + theta * (geom.lo + geom.up)
    return (thomas(a, b, c, dE) - dE) / dt


def zonal_substep(T, x, dt, nlon, par):
    """EXTENSION, not in the reference (the zonal half) — "parity unpinned" by construction.  Defined in
    include/ebm_hip.h (ebm_zonal_diffusion); this is the checker's restatement, the C oracle restates it with a different
    algorithm (Thomas + Sherman-Morrison), the HIP kernel uses a third (periodic Thomas carrying the last unknown).

    ``T`` is [nmember*nlon, nx] (column = member*nlon + longitude, periodic in longitude).  Per member and latitude k the
    periodic tridiagonal system
        (1 + 2 a_k) U_l - a_k (U_{l-1} + U_{l+1}) = T_l,     a_k = (dt/cw) D / ((1 - x_k)(1 + x_k) dlambda^2),  dlambda = 2 pi / nlon
    is circulant: its eigenvectors are the discrete Fourier modes exp(2 pi i m l / nlon) with eigenvalues
    1 + a_k (2 - 2 cos(2 pi m / nlon)) — solved here exactly that way (real FFT along the longitude axis).
    Returns (U, Z) with Z = (U - T) cw/dt, the backward-Euler zonal heat-flux convergence D/((1-x^2) dlambda^2) d2U."""
    T = np.asarray(T, dtype=np.float64)
    ncol, nx = T.shape
    assert ncol % nlon == 0 and nlon >= 3
    dl = 2.0 * math.pi / nlon
    a = (dt / par["cw"]) * par["D"] / (((1.0 - x) * (1.0 + x)) * (dl * dl))            # [nx]
    lam = 2.0 - 2.0 * np.cos(2.0 * math.pi * np.arange(nlon // 2 + 1) / nlon)          # [nlon//2+1]
    Tm = T.reshape(ncol // nlon, nlon, nx)
    U = np.fft.irfft(np.fft.rfft(Tm, axis=1) / (1.0 + lam[None, :, None] * a[None, None, :]), n=nlon, axis=1)
    U = U.reshape(ncol, nx)
    return U, (U - T) * (par["cw"] / dt)


def step_miz(ct, f, vars, T0_warm, x, dt, geom, par, imex=False, zon=None):
    """One MIZ step, src/miz.jl:150-196.

    ``vars`` holds Ei, Ew, h, D, phi (1-D arrays).  Returns (new vars dict with all 10
    variables, new warm start T0, n_solves, converged).  ``ct`` = cos(2.0*pi*t).
    ``imex=True``: the extension of implicit_diffusion_correction (NOT the reference's scheme);
    ``zon``: with it, a zonal term Z (zonal_substep of the previous step's output T) added to the diffusion term of both
    vertical fluxes before the implicit solve — the operator-split COUPLING EXPERIMENT of tests/test_oracle_zonal.py, which
    is how the instability over thin ice was found; it exists only in the checker, the library ships the operator alone."""
    Ei, Ew, h, D, phi = (vars[k] for k in ("Ei", "Ew", "h", "D", "phi"))
    with np.errstate(all="ignore"):
        Tw = water_temp(Ew, phi, par)
        Tw = condset(Tw, 0.0, np.isnan(Tw))
        Ti, T0, nit, ok = solveTi(T0_warm, x, ct, h, Tw, phi, f, geom, par)
        n = num(D, phi, par)
        Fvi = vert_flux(x, ct, True, Ti, Tw, phi, f, geom, par)
        Fvw = vert_flux(x, ct, False, Ti, Tw, phi, f, geom, par)
        if zon is not None:
            assert imex, "the zonal term belongs to the implicit-diffusion extension"
            tb = Tbar(Ti, Tw, phi)                               # vert_flux with dif + Z in place of dif
            L = par["A"] + par["B"] * (tb - par["Tm"])
            dif0 = geom.add(np.zeros_like(x), tb) + zon
            Fvi = solar_add(np.zeros_like(x), x, ct, True, par) - L + dif0 + par["Fb"] + f
            Fvw = solar_add(np.zeros_like(x), x, ct, False, par) - L + dif0 + par["Fb"] + f
        if imex:
            corr = implicit_diffusion_correction((phi * Fvi + (1.0 - phi) * Fvw) * dt, dt, geom, par)
            tb = Tbar(Ti, Tw, phi)                               # vert_flux with dif + corr in place of dif
            L = par["A"] + par["B"] * (tb - par["Tm"])
            dif = geom.add(np.zeros_like(x), tb)
            if zon is not None:
                dif = dif + zon
            dif = dif + corr
            Fvi = solar_add(np.zeros_like(x), x, ct, True, par) - L + dif + par["Fb"] + f
            Fvw = solar_add(np.zeros_like(x), x, ct, False, par) - L + dif + par["Fb"] + f
        Flat = lat_flux(h, D, Tw, phi, par)
        rEi = Ei + (phi * Fvi + Flat) * dt
        rEw = Ew + ((1.0 - phi) * Fvw - Flat) * dt
        Ei_n, Ew_n, psiEidt, psiEwdt = redistributeE(rEi, rEw)
        Al = area_lead(D, phi, n, par)
        Ql, Qp = split_psiEw(psiEwdt / dt, phi, Al)
        dn = dt * psinplus(Qp, par)
        rD = D + D_t(h, D, Tw, phi, Ql, par) * dt
        D_n = average(rD, par["Dmin"], n, dn)
        D_n = jl_clamp(D_n, par["Dmin"], par["Dmax"])
        D_n = zeroref(D_n, Ei_n)
        rh = h + ((-1.0 / par["Lf"]) * Fvi) * dt
        rh = jl_clamp(rh, 0.0, np.inf)
        h_n = average(rh, par["hmin"], n, dn)
        phi_n = concentration(Ei_n, h_n, par)
        Ei_n = zeroref(Ei_n, h_n)
        E = phi_n * Ei_n + (1.0 - phi_n) * Ew_n
        T = Tbar(Ti, Tw, phi_n)
        Ti_out = condset(Ti, np.nan, Ei_n == 0.0)
        Tw_out = condset(Tw, np.nan, phi_n > 0.99)
    out = dict(Ei=Ei_n, Ew=Ew_n, h=h_n, D=D_n, phi=phi_n, Tw=Tw_out, Ti=Ti_out, n=n, E=E, T=T)
    return out, T0, nit, ok


# --------------------------------------------------------------------------------------
# classic model                                                src/classic.jl
# --------------------------------------------------------------------------------------
class ClassicStatics:
    """get_statics, src/classic.jl:12-34 (kappa kept as its three diagonals; the reference
    stores the same tridiagonal matrix densely)."""

    def __init__(self, st_x, nx, dt, par):
        self.cg_tau = par["cg"] / par["tau"]
        self.dt_tau = dt / par["tau"]
        self.dc = self.dt_tau * self.cg_tau
        g = DiffusionGeometry("identity", st_x, 1.0)           # get_diffop(nx), unscaled
        dtD = dt * par["D"]
        one = 1.0 + self.dt_tau
        self.k_sub = 0.0 - (dtD * g.sub) / par["cg"]
        self.k_sup = 0.0 - (dtD * g.sup) / par["cg"]
        self.k_diag = one - (dtD * g.diag) / par["cg"]
        self.M = par["B"] + self.cg_tau
        self.aw = par["a0"] - par["a2"] * (st_x * st_x)
        self.kLf = par["k"] * par["Lf"]
        self.S_base = par["S0"] - par["S2"] * (st_x * st_x)

    def S(self, x, ct, par):
        """Column of S for a time with cos(2.0*pi*t) = ct, src/classic.jl:23-24."""
        return self.S_base - (par["S1"] * ct) * x


def classic_time_index(t, dt, nt):
    """src/classic.jl:45: i = round(Int, mod1((t + dt/2)*nt, nt)), 1-based."""
    y = (t + dt / 2.0) * nt
    m = math.fmod(y, nt)
    if m < 0:
        m += nt
    if m == 0.0:
        m = float(nt)
    return _round_half_even(m)


def step_classic(ct_i, ct_ip1, f, vars, x, dt, stat, par):
    """One classic (WE15) step, src/classic.jl:37-71.

    ct_i = cos(2.0*pi*st.t[i]) and ct_ip1 the same for column i+1 of S (wrapping to column 1
    after nt, src/classic.jl:25).  vars holds E, Tg.  Returns dict(E, Tg, T, h)."""
    E, Tg = vars["E"], vars["Tg"]
    A, cw, ai, Lf, Fb = par["A"], par["cw"], par["ai"], par["Lf"], par["Fb"]
    with np.errstate(all="ignore"):
        S_i = stat.S(x, ct_i, par)
        S_ip1 = stat.S(x, ct_ip1, par)
        alpha = bool_mul(stat.aw, E > 0.0) + bool_mul(np.full_like(E, ai), E < 0.0)
        C = alpha * S_i + stat.cg_tau * Tg - A + f
        T0 = C / (stat.M - stat.kLf / E)
        T = bool_mul(E / cw, E >= 0.0) + bool_mul(bool_mul(T0, E < 0.0), T0 < 0.0)
        E = E + dt * (C - stat.M * T + Fb)
        den = stat.M - stat.kLf / E
        q = bool_mul(bool_mul(stat.dc / den, T0 < 0.0), E < 0.0)
        rhs = Tg + stat.dt_tau * (
            bool_mul(E / cw, E >= 0.0)
            + bool_mul(bool_mul((ai * S_ip1 - A + f) / den, T0 < 0.0), E < 0.0)
        )
        Tg_n = thomas(stat.k_sub, stat.k_diag - q, stat.k_sup, rhs)
        h = bool_mul(-E / Lf, E < 0.0)
    return dict(E=E, Tg=Tg_n, T=T, h=h)


# --------------------------------------------------------------------------------------
# driver                                                       src/infrastructure.jl:536-636
# --------------------------------------------------------------------------------------
MIZ_SOLVARS = ("E", "T", "h", "Ei", "Ew", "Ti", "Tw", "D", "phi", "n")
CLASSIC_SOLVARS = ("E", "T", "h")


def cos2pit(t: float) -> float:
    """cos(2.0*pi*t) as the reference writes it (src/miz.jl:11, src/classic.jl:24)."""
    return math.cos(2.0 * math.pi * t)


class Solutions:
    """src/infrastructure.jl:333-383 — raw[var][ti], seasonal.{winter,summer,avg}[var][year]."""

    def __init__(self, st, varnames, lastonly=True):
        self.spacetime, self.lastonly = st, lastonly
        if lastonly:
            self.ts = np.array([float(Fraction(st.dur - 1) + Fraction(2 * i + 1, 2 * st.nt))
                                for i in range(st.nt)])
        else:
            self.ts = st.T.copy()
        self.raw = {v: [None] * len(self.ts) for v in varnames}
        self.winter = {v: [None] * st.dur for v in varnames}
        self.summer = {v: [None] * st.dur for v in varnames}
        self.avg = {v: [None] * st.dur for v in varnames}


def integrate(model, st, forcing, par, init, lastonly=True, imex=False):
    """integrate + savesol!, src/infrastructure.jl:549-591, 615-636.

    The classic branch drives step_classic directly (at the reference commit
    ``integrate(:Classic, ...)`` throws on the ``verbose`` keyword — SURVEY F7 — so this is
    what a fixed caller would do)."""
    nt, dt, x = st.nt, st.dt, st.x
    vars = {k: np.array(v, dtype=np.float64) for k, v in init.items()}
    names = MIZ_SOLVARS if model == "MIZ" else CLASSIC_SOLVARS
    sols = Solutions(st, names, lastonly)
    annual = {v: [None] * nt for v in names}
    stats = dict(solves=0, failures=0)
    if model == "MIZ":
        geom = DiffusionGeometry(st.kind, x, par["D"])
        T0 = np.zeros(st.nx)
    else:
        stat = ClassicStatics(x, st.nx, dt, par)
    ctab = [cos2pit(t) for t in st.t]
    for tinx in range(1, nt * st.dur + 1):                   # 1-based like the reference
        ti = (tinx - 1) % nt + 1
        f = forcing(float(st.T[tinx - 1]))
        if model == "MIZ":
            vars, T0, nit, ok = step_miz(ctab[ti - 1], f, vars, T0, x, dt, geom, par, imex=imex)
            stats["solves"] += nit
            stats["failures"] += 0 if ok else 1
        else:
            i = classic_time_index(float(st.t[ti - 1]), dt, nt)
            out = step_classic(ctab[i - 1], ctab[i % nt], f, vars, x, dt, stat, par)
            vars = out
        year = int(math.ceil(st.T[tinx - 1]))
        for v in names:
            cp = vars[v].copy()
            annual[v][ti - 1] = cp
            if not lastonly:
                sols.raw[v][tinx - 1] = cp
            elif tinx > nt * st.dur - nt:
                sols.raw[v][ti - 1] = cp
            if ti == st.winter_inx:
                sols.winter[v][year - 1] = cp
            elif ti == st.summer_inx:
                sols.summer[v][year - 1] = cp
            elif ti == nt:
                with np.errstate(all="ignore"):
                    sols.avg[v][year - 1] = np.sum(np.stack(annual[v]), axis=0) / nt
    sols.final = vars
    sols.stats = stats
    if model == "MIZ":
        sols.T0 = T0
    return sols
