"""How far can the reference's own T0 be from the root the oracle (and the GPU) compute?  MEASURED.

The one place where the oracle is not a transcription is ``solveTi`` (reference src/miz.jl:47-68):
the reference hands the residual ``T0eq`` (:33-45) to the third-party ``NonlinearSolve.TrustRegion``
with ``reltol = 1e-6, abstol = 1e-8`` (:55-60) and accepts whatever iterate first satisfies the
solver's termination test; the oracle solves the piecewise-linear system exactly (active-set Newton).
Julia is not available here, so the reference's iterate cannot be produced; what CAN be
done on the CPU is to put OTHER trust-region solvers, stopped as early as the reference's tolerances
allow, behind the transcribed residual, run the reference test's configuration
(test/runtests.jl:22-32: SpaceTime{sin}(180, 2000, 1), Forcing(0.0), zero state) through the ten steps
the reference test looks at (:40-41), and compare the step-10 state of all ten variables with the
exact-root trajectory under the reference test's own criterion (:42-46: NaN -> 0, elementwise
``isapprox``, rtol = sqrt(eps), atol = 0).

Three stand-ins for the reference's solver, all started from the reference's warm start (the previous
step's T0, zeros at step 1):
  * a dogleg trust-region Newton with the analytic Jacobian of ``T0eq`` and the radius-update constants
    NonlinearSolve documents for ``TrustRegion()`` (shrink 1/4 below rho = 1/4, expand x2 above 3/4,
    accept above 1e-4, initial radius = max radius / 11), returning the FIRST iterate with
    max|T0eq| <= abstol — the earliest stop the reference's tolerance allows;
  * SciPy's MINPACK ``hybr`` (Powell's dogleg trust region with Broyden updates, no analytic Jacobian),
    xtol = the reference's reltol;
  * an adversarial perturbation: the exact root moved along J^-1 r with |r_k| = abstol, i.e. a T0 whose
    residual sits AT the acceptance threshold in every cell (random signs, 8 seeds) — the worst any
    solver honouring abstol could hand back, and nothing a Newton-type iteration produces.
Measured (this file, -s): the two real solvers end 6e-13 (trust region) and 2e-11 (hybr) from the
exact-root trajectory at step 10, elementwise relative, every variable — four and three orders inside
the reference test's 1.49e-8.  The adversarial T0s (1.5e-9 K from the root) end up to 2e-8 away in
single cells of Tw and n: with a residual parked AT abstol in every cell, two builds of the reference
could themselves disagree at the level of the reference's own test tolerance.
This turns "the oracle would pass the reference's own test" from an estimate into a measurement; it
does NOT pin the oracle to the reference (parity stays unpinned by environment).
"""
import numpy as np
import pytest
from scipy import optimize

SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))
VARS = ("E", "T", "h", "Ei", "Ew", "Ti", "Tw", "D", "phi", "n")
ABSTOL, RELTOL = 1e-8, 1e-6                 # src/miz.jl:58-59


def isapprox_all(a, b):
    """test/runtests.jl:42-46: NaN -> 0 on both sides, all(isapprox.(a, b)) with rtol = sqrt(eps), atol = 0."""
    a, b = np.nan_to_num(a, nan=0.0), np.nan_to_num(b, nan=0.0)
    return bool(np.all(np.abs(a - b) <= SQRT_EPS * np.maximum(np.abs(a), np.abs(b))))


def worst_rel(a, b):
    a, b = np.nan_to_num(a, nan=0.0), np.nan_to_num(b, nan=0.0)
    m = np.maximum(np.abs(a), np.abs(b))
    with np.errstate(invalid="ignore", divide="ignore"):
        r = np.where(m > 0, np.abs(a - b) / m, 0.0)
    return float(r.max())


def jacobian(o, T0, hp, phi, geom, par):
    """d T0eq / d T0: -diag(k/hp + B) + Dif * diag(phi * [T0 < Tm]) (tridiagonal, dense here)."""
    nx = len(T0)
    g = np.where(T0 < par["Tm"], phi, 0.0)
    J = np.diag(-(par["k"] / hp + par["B"]) + geom.di * g)
    J[np.arange(1, nx), np.arange(nx - 1)] = geom.lo[1:] * g[:-1]
    J[np.arange(nx - 1), np.arange(1, nx)] = geom.up[:-1] * g[1:]
    return J


def trust_region_dogleg(F, Jf, u0, abstol=ABSTOL, maxiter=1000):
    """Dogleg trust-region Newton, radius update as documented for NonlinearSolve.TrustRegion()'s
    default (Simple) scheme; returns the FIRST iterate with max|F| <= abstol."""
    u = u0.copy()
    fu = F(u)
    if np.max(np.abs(fu)) <= abstol:
        return u, 0
    max_radius = max(float(np.linalg.norm(fu)), float(u.max() - u.min()))
    radius = max_radius / 11.0
    for it in range(1, maxiter + 1):
        J = Jf(u)
        g = J.T @ fu
        dn = -np.linalg.solve(J, fu)
        if np.linalg.norm(dn) <= radius:
            d = dn
        else:
            Jg = J @ g
            dc = -(g @ g) / (Jg @ Jg) * g
            if np.linalg.norm(dc) >= radius:
                d = -radius * g / np.linalg.norm(g)
            else:                                    # on the dogleg between the Cauchy and the Newton point
                w = dn - dc
                a, b, c = w @ w, 2.0 * (dc @ w), dc @ dc - radius * radius
                d = dc + (-b + np.sqrt(b * b - 4.0 * a * c)) / (2.0 * a) * w
        fn = F(u + d)
        pred = 0.5 * (fu @ fu) - 0.5 * np.sum((fu + J @ d) ** 2)
        rho = (0.5 * (fu @ fu) - 0.5 * (fn @ fn)) / pred if pred > 0 else -1.0
        if rho < 0.25:
            radius *= 0.25
        elif rho > 0.75:
            radius = min(2.0 * radius, max_radius)
        if rho > 1e-4:
            u, fu = u + d, fn
            if np.max(np.abs(fu)) <= abstol:
                return u, it
    raise AssertionError("trust region did not converge")


def run_ten_steps(oracle, monkeypatch, solver, nsteps=10):
    """The reference test's configuration with `solver(T0_warm, F, Jf) -> T0` behind solveTi; returns
    the state after every step, the residuals max|T0eq(T0)| and the distances to the exact root."""
    o = oracle
    st = o.SpaceTime("sin", 180, 2000, 1)
    par = o.default_parameters("MIZ")
    geom = o.DiffusionGeometry("sin", st.x, par["D"])
    exact = o.solve_T0
    log = {"resid": [], "dist": []}

    def patched(T0_warm, x, ct, hp, Tw, phi, f, g, p):
        F = lambda T0: o.T0eq(T0, x, ct, hp, Tw, phi, f, g, p)                  # noqa: E731
        Jf = lambda T0: jacobian(o, T0, hp, phi, g, p)                          # noqa: E731
        root, nit, ok = exact(T0_warm, x, ct, hp, Tw, phi, f, g, p)
        T0 = root if solver is None else solver(T0_warm, F, Jf, root)
        log["resid"].append(float(np.max(np.abs(F(T0)))))
        log["dist"].append(float(np.max(np.abs(T0 - root))))
        return T0, nit, ok

    monkeypatch.setattr(o, "solve_T0", patched)
    vars_ = {k: np.zeros(st.nx) for k in ("Ei", "Ew", "h", "D", "phi")}
    T0 = np.zeros(st.nx)
    states = []
    for s in range(nsteps):
        out, T0, _, _ = o.step_miz(o.cos2pit(float(st.t[s])), 0.0, vars_, T0, st.x, st.dt, geom, dict(par))
        vars_ = {k: out[k] for k in ("Ei", "Ew", "h", "D", "phi")}
        states.append(out)
    monkeypatch.setattr(o, "solve_T0", exact)
    return states, log


def solver_tr(T0_warm, F, Jf, root):
    return trust_region_dogleg(F, Jf, T0_warm)[0]


def solver_hybr(T0_warm, F, Jf, root):
    sol = optimize.root(F, T0_warm, method="hybr", options={"xtol": RELTOL})
    assert sol.success, sol.message
    return sol.x


def make_adversarial(seed):
    rng = np.random.default_rng(seed)

    def solver(T0_warm, F, Jf, root):
        r = ABSTOL * rng.choice([-1.0, 1.0], size=len(root))
        return root + np.linalg.solve(Jf(root), r)          # F(T0) = r up to the active set
    return solver


@pytest.fixture(scope="module")
def exact_states(oracle):
    mp = pytest.MonkeyPatch()
    states, log = run_ten_steps(oracle, mp, None)
    mp.undo()
    assert max(log["resid"]) < ABSTOL                       # the exact root passes the reference's own acceptance test
    return states


@pytest.mark.parametrize("name", ["trust_region_first_acceptable_iterate", "scipy_hybr_xtol_1e-6"])
def test_step10_state_under_reference_tolerances(oracle, monkeypatch, exact_states, name):
    solver = solver_tr if name.startswith("trust") else solver_hybr
    states, log = run_ten_steps(oracle, monkeypatch, solver)
    # what the stand-in solver delivered
    assert max(log["resid"]) <= (ABSTOL if name.startswith("trust") else 1e-6), log["resid"]
    worst = {v: worst_rel(states[9][v], exact_states[9][v]) for v in VARS}
    print(f"\n[{name}] max|T0eq| per step: " + " ".join(f"{r:.1e}" for r in log["resid"]))
    print(f"[{name}] max|T0 - root| per step: " + " ".join(f"{d:.1e}" for d in log["dist"]))
    print(f"[{name}] step-10 worst elementwise relative difference: "
          + ", ".join(f"{v} {worst[v]:.1e}" for v in VARS))
    for v in VARS:                                           # the reference test's criterion, variable by variable
        assert isapprox_all(states[9][v], exact_states[9][v]), (v, worst[v])
    # and far inside it: a full Newton step on a piecewise-linear system lands on the root
    assert max(worst.values()) <= 1e-10, worst


@pytest.mark.parametrize("seed", range(8))
def test_step10_state_with_residual_at_the_acceptance_threshold(oracle, monkeypatch, exact_states, seed):
    """Worst case the reference's abstol admits: every step's T0 has |T0eq| = 1e-8 in every cell."""
    states, log = run_ten_steps(oracle, monkeypatch, make_adversarial(seed))
    assert 0.5 * ABSTOL <= max(log["resid"]) <= 2.0 * ABSTOL
    worst = {v: worst_rel(states[9][v], exact_states[9][v]) for v in VARS}
    print(f"\n[adversarial {seed}] max|T0 - root| {max(log['dist']):.2e} K; step-10 worst relative difference: "
          + ", ".join(f"{v} {worst[v]:.1e}" for v in VARS))
    assert max(log["dist"]) <= 6e-9                          # abstol / min(k/h + B): SURVEY 8(c)'s estimate
    strict = [v for v in VARS if not isapprox_all(states[9][v], exact_states[9][v])]
    print(f"[adversarial {seed}] variables outside the reference test's elementwise isapprox: {strict or 'none'}")
    # measured worst case over the 8 seeds: 2.1e-8 (Tw, n) — the same order as the reference test's own
    # rtol = 1.49e-8; bounded here at a few times the measurement
    assert max(worst.values()) <= 1e-7, worst
