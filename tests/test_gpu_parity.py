"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle.

Tolerances (fp64) are MEASURED, not allowed for.  Everything except the tridiagonal solves is a
bit-exact restatement, so from identical inputs one step differs from the oracle only through the
T0 (or Tg) solve, whose forward error is cond(J)*eps for ANY backward-stable algorithm (the oracle's
Thomas, the reference's dense LU, the GPU's partition + cyclic reduction); along a trajectory that
difference is then carried by the model's own dynamics.  Every comparison below records the error it
observed (conftest.record_error -> gpurun_out/measured_errors.jsonl; the round's table is committed
as profiles/r02_measured_errors.jsonl) and its bar is

    bar = min(10 x the error measured on MI355X, the bar BASELINE.json / the reference test gives)

with the measured value written next to it.  Bars are at most 1e-10 — BASELINE.json's figure, two
orders inside the reference test's own isapprox rtol 1.49e-8 (test/runtests.jl:46) — with these stated
exceptions, each explained where it is used: (1) T0 / Ti of 4096-cell meridians during the first tens of steps
from the zero state, where T0 is extremely sensitive to the concentration of the newly frozen cells at the ice
edge: the bar is what that sensitivity, computed in the test cell by cell from the columns' own h, phi and T0,
makes of the differences in phi and Ew actually observed (t0_error_explained_by_state below), times a stated constant (measured 2.0e-10 at step 10,
2.2e-9 at step 20 while every other field is below 1e-12; the same bar in tests/test_gpu_configs.py);
(2) the implicit-diffusion extension at time steps hundreds of times beyond the explicit limit (<= 9e-10,
tests/test_gpu_imex.py); (3) trajectories the model itself amplifies beyond 1e-10, where the bar is a multiple
of the fp64 oracle's own distance from its 80-bit build (tiny sizes on the identity grid) or of the oracle's
4-ulp envelope (the shadowing tests, the year-long run on the identity grid, TOL_YEAR).  profiles/r02_error_budget.txt adds, per configuration, how far the GPU and
the fp64 oracle EACH are from the same model evaluated in 80-bit extended precision: the GPU is as
close to it as the oracle is (test_gpu_error_budget.py).
Long trajectories: the reference's own test configuration (sin grid, nx = 180, nt = 2000) is
chaotic — a 1-ulp perturbation of the ORACLE grows to O(1) within the year (DESIGN.md
"Sensitivity") — so a year-long comparison is only meaningful on the identity grid, where the
same perturbation stays below 1e-9.
NaN sentinels (src/miz.jl:193-194) must coincide exactly.
"""
import numpy as np
import pytest

from conftest import load_golden, record_error, scaled_err

pytestmark = pytest.mark.gpu

TOL_STEP = 1e-11     # one step from a golden state: measured <= 2.1e-12 (identity grid, step 1000)
TOL_SHORT = 1e-10    # BASELINE.json's bar; trajectories of <= 50 steps at nlat <= 257 measure <= 5.7e-11
TOL_TRAJ10 = 1e-12   # the reference test's own comparison point (step 10 from zero): measured 8.5e-14
TOL_YEAR = 1e-7      # 2000 steps on the identity grid: measured 7.3e-9
PROG = ("Ei", "Ew", "h", "D", "phi")
DIAG = ("Tw", "Ti", "n", "E", "T")
ALL = PROG + ("T0",) + DIAG


def make_engine(pkg, model, st, par, ncol=1):
    return pkg.Engine(model, st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval),
                      st.dt, ncol, device=0)


def ctab(pkg, st):
    return np.array([pkg.cos2pit(float(t)) for t in st.t])


def check_all(got, ref, tol, names=ALL, what=""):
    errs = {k: scaled_err(got[k], ref[k]) for k in names}
    worst = max(errs, key=errs.get)
    record_error(what, worst, errs[worst], tol)
    for k in names:
        assert errs[k] <= tol, f"{what}: {k} scaled error {errs[k]:.3e} > {tol:.1e}"


def t0_error_explained_by_state(oracle, kind, x, par, ref, got):
    """What the observed difference between two runs' STATES (phi, Ew) makes of their T0, cell by cell.

    Row k of the T0 system (src/miz.jl:33-45) reads  -(k/h' + B) v_k + Dif(phi [v<0] v)_k = -(ai S - A + Dif((1-phi)(Tw-Tm))_k + f),
    v = T0 - Tm, Dif_k(u) = lo_k u_{k-1} - (lo_k + up_k) u_k + up_k u_{k+1} with lo + up ~ 0.81 D nlat^2 (1.4e7 at 4096
    latitudes) — over a diagonal of only k/hmin + B = 22 where new ice is thin and sparse.  A change of phi or of
    r = (1-phi)(Tw-Tm) = Ew/cw in cells k-1 ... k+1 therefore moves v_k by up to
        (lo_k + up_k) (|v_k| [v_k<0] |dphi| + |dEw|/cw) / (k/h'_k + B + (lo_k + up_k) phi_k [v_k<0]):
    1e4 ... 1e6 kelvin per unit of concentration at the advancing ice edge of a freeze-up.  The newly frozen cells' enthalpy
    is the small remainder of the water's, so its rounding is relatively large: two runs differ by ~1e-14 in phi there, hence
    by ~1e-10 in T0 — in any arithmetic (checked on the CPU: the fp64 oracle against its own 80-bit build, 180 ... 4096
    latitudes, both grids, steps 5 ... 60: the observed T0 difference is 0.1 ... 1.8 x this expression).
    `ref`, `got`: dicts of [ncol, nlat] arrays h, phi, T0, Ew (the state a step ends with stands in for the one the next
    begins with).  Returns the maximum over cells and columns."""
    geom = oracle.DiffusionGeometry("identity" if kind == "identity" else "sin", x, par["D"])
    lu = geom.lo + geom.up

    def nbr(d):                      # max over cells k-1, k, k+1
        e = np.concatenate(([0.0], np.abs(d), [0.0]))
        return np.maximum(np.maximum(e[:-2], e[1:-1]), e[2:])
    worst = 0.0
    for i in range(ref["phi"].shape[0]):
        hp = np.where(ref["h"][i] == 0.0, par["hmin"], ref["h"][i])
        dd = par["k"] / hp + par["B"]
        v = np.abs(ref["T0"][i] - par["Tm"])
        act = (ref["T0"][i] < par["Tm"]).astype(float)
        dphi = nbr(got["phi"][i] - ref["phi"][i])
        dr = nbr((got["Ew"][i] - ref["Ew"][i]) / par["cw"])
        worst = max(worst, float(np.max(lu * (v * act * dphi + dr) / (dd + lu * ref["phi"][i] * act))))
    return worst


# ---- golden fixtures: the reference test's configuration -------------------------------------
@pytest.mark.parametrize("kind", ["sin", "identity"])
def test_miz_trajectory_from_zero_matches_golden(pkg, kind, cells):
    """test/runtests.jl:22-47: zero initial state, compare the state after step 10 (the index
    the reference test checks); also steps 1 and 2."""
    g = load_golden(f"miz_{kind}_180_2000.npz")
    st = pkg.SpaceTime(kind, 180, 2000, 1)
    assert np.array_equal(st.x, g["x"])
    with make_engine(pkg, "MIZ", st, pkg.default_parameters("MIZ")) as eng:
        assert eng.launch_info()["cells_per_thread"] == cells
        eng.set_time_table(st.t)
        done = 0
        for s in (1, 2, 10):
            eng.run(done, s - done)
            done = s
            got = {k: v[0] for k, v in eng.get_state(ALL).items()}
            check_all(got, {k: g[f"s{s}_{k}"] for k in ALL}, TOL_TRAJ10, what=f"{kind} step {s}")
        assert eng.counters()["cap_hits"] == 0


@pytest.mark.parametrize("kind", ["sin", "identity"])
@pytest.mark.parametrize("s", [10, 100, 522, 1000, 1548, 1999])
def test_miz_one_step_from_golden_state(pkg, kind, s, cells):
    """Load the oracle's state after step s (freeze-up, winter, melt season, year end), take
    ONE step on the GPU, compare with the oracle's state after step s+1."""
    g = load_golden(f"miz_{kind}_180_2000.npz")
    st = pkg.SpaceTime(kind, 180, 2000, 1)
    with make_engine(pkg, "MIZ", st, pkg.default_parameters("MIZ")) as eng:
        eng.set_time_table(st.t)
        for k in PROG + ("T0",):
            eng.set_field(k, g[f"s{s}_{k}"][None])
        eng.run(s, 1)
        got = {k: v[0] for k, v in eng.get_state(ALL).items()}
    check_all(got, {k: g[f"s{s+1}_{k}"] for k in ALL}, TOL_STEP, what=f"{kind} step {s}->{s+1}")


def test_miz_year_long_on_stable_grid(pkg, coracle):
    """Identity grid: one full year (2000 steps) stays within TOL_YEAR of the oracle."""
    g = load_golden("miz_identity_180_2000.npz")
    st = pkg.SpaceTime("identity", 180, 2000, 1)
    with make_engine(pkg, "MIZ", st, pkg.default_parameters("MIZ")) as eng:
        eng.set_time_table(st.t)
        eng.run(0, 2000)
        got = {k: v[0] for k, v in eng.get_state(ALL).items()}
        cnt = eng.counters()
    check_all(got, {k: g[f"s2000_{k}"] for k in ALL}, TOL_YEAR, what="identity year")
    assert cnt["cap_hits"] == 0 and cnt["steps"] == 2000


def test_miz_year_on_reference_config_shadows_the_oracle(pkg, coracle):
    """The reference test's own configuration (sin grid, nx = 180, nt = 2000) amplifies a 1-ulp
    perturbation to O(1) within the year, so a year-long pointwise tolerance is meaningless there.
    What can be asked: along the year the GPU's distance from the oracle stays inside the envelope
    by which the ORACLE itself moves when one parameter is perturbed by a few ulps, and the
    year-end climate (hemispheric means, ice edge) agrees."""
    nlat, nt = 180, 2000
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    ct = ctab(pkg, st)
    checkpoints = (10, 50, 100, 200, 400, 700, 1000, 1500, 2000)

    def oracle_run(p):
        state = {k: np.zeros((1, nlat)) for k in PROG + ("T0",)}
        out, done = {}, 0
        for s in checkpoints:
            diag, _ = coracle.miz_run(1, st.x, dict(p), st.dt, ct[done:s], np.zeros(s - done), None, state)
            out[s] = {k: v[0].copy() for k, v in dict(state, **diag).items()}
            done = s
        return out

    ref = oracle_run(par)
    envelope = {s: 0.0 for s in checkpoints}
    for scale in (1.0 + 2.0 ** -50, 1.0 - 2.0 ** -50):           # Fb moved by 4 ulps, either way
        pert = dict(par)
        pert["Fb"] = par["Fb"] * scale
        alt = oracle_run(pert)
        for s in checkpoints:
            envelope[s] = max(envelope[s], max(scaled_err(alt[s][k], ref[s][k]) for k in PROG))
    with make_engine(pkg, "MIZ", st, par) as eng:
        eng.set_time_table(st.t)
        done = 0
        for s in checkpoints:
            eng.run(done, s - done)
            done = s
            got = {k: v[0] for k, v in eng.get_state(ALL).items()}
            dist = max(scaled_err(got[k], ref[s][k]) for k in PROG)
            record_error(f"reference config shadow step {s} (envelope {envelope[s]:.2e})", "PROG", dist, max(TOL_SHORT, 5.0 * envelope[s]))
            assert dist <= max(TOL_SHORT, 5.0 * envelope[s]), f"step {s}: {dist:.2e} vs envelope {envelope[s]:.2e}"
        assert eng.counters()["cap_hits"] == 0
    # year-end climate
    end = ref[2000]
    assert abs(pkg.hemispheric_mean(got["T"], st.x) - pkg.hemispheric_mean(end["T"], st.x)) < 0.05
    assert abs(pkg.hemispheric_mean(got["phi"], st.x) - pkg.hemispheric_mean(end["phi"], st.x)) < 5e-3
    assert abs(int(np.argmax(got["phi"] > 0)) - int(np.argmax(end["phi"] > 0))) <= 1


def test_headline_meridians_shadow_the_oracle(pkg, coracle):
    """The bench configuration (4096 latitudes, sin grid, nt = 2^20) is as sensitive at the ice edge
    as the reference's test configuration: beyond ~100 steps cells flip between ice and water in
    one run and not in the other.  Same shadowing criterion, on 8 meridians of the bench's forcing
    ramp: through 1000 steps the GPU stays inside 5x the oracle's own 4-ulp envelope."""
    nlat, nt, ncol = 4096, 1048576, 8
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    pert = dict(par)
    pert["Fb"] = par["Fb"] * (1.0 + 2.0 ** -50)
    fcol = 0.5 * np.sin(2.0 * np.pi * np.arange(ncol) / ncol)
    ct = ctab(pkg, st)
    ref = {k: np.zeros((ncol, nlat)) for k in PROG + ("T0",)}
    alt = {k: np.zeros((ncol, nlat)) for k in PROG + ("T0",)}
    with make_engine(pkg, "MIZ", st, par, ncol) as eng:
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        done = 0
        for s in (10, 20, 50, 200, 500, 1000):
            n = s - done
            coracle.miz_run(1, st.x, dict(par), st.dt, ct[done:s], np.zeros(n), fcol, ref)
            coracle.miz_run(1, st.x, pert, st.dt, ct[done:s], np.zeros(n), fcol, alt)
            eng.run(done, n, None, False)
            done = s
            got = eng.get_state(PROG)
            dist = max(scaled_err(got[k], ref[k]) for k in PROG)
            envelope = max(scaled_err(alt[k], ref[k]) for k in PROG)
            # the first tens of steps are not yet in the sensitive regime: there the prognostics must agree to 1e-12 outright
            bar = 1e-12 if s <= 20 else max(TOL_SHORT, 5.0 * envelope)
            record_error(f"headline shadow step {s} (envelope {envelope:.2e})", "PROG", dist, bar)
            assert dist <= bar, f"step {s}: {dist:.2e} vs {envelope:.2e}"
        assert eng.counters()["cap_hits"] == 0


def test_t0_meets_reference_solver_criterion(pkg, coracle):
    """The reference accepts T0 when |T0eq(T0)| <= abstol = 1e-8 (src/miz.jl:58-59): the
    GPU's T0 must satisfy the reference's residual function at that level."""
    g = load_golden("miz_sin_180_2000.npz")
    st = pkg.SpaceTime("sin", 180, 2000, 1)
    par = pkg.default_parameters("MIZ")
    for s in (10, 522, 1548):
        with make_engine(pkg, "MIZ", st, par) as eng:
            eng.set_time_table(st.t)
            for k in PROG + ("T0",):
                eng.set_field(k, g[f"s{s}_{k}"][None])
            eng.run(s, 1)
            T0 = eng.get_field("T0")[0]
        res = coracle.T0eq(1, st.x, dict(par), pkg.cos2pit(float(st.t[s])), 0.0,
                           g[f"s{s}_h"], g[f"s{s}_Ew"], g[f"s{s}_phi"], T0)
        record_error(f"|T0eq(T0_gpu)| at 180 cells, step {s}", "T0eq", float(np.max(np.abs(res))), 1e-8)
        assert np.max(np.abs(res)) < 1e-8, (s, float(np.max(np.abs(res))))


@pytest.mark.parametrize("nlat,nt,steps", [(1440, 131072, (10, 20, 60)), (4096, 1048576, (10, 20, 40))])
def test_t0_meets_reference_solver_criterion_at_high_resolution(pkg, coracle, nlat, nt, steps):
    """The same acceptance test where the T0 system is 60 ... 500 times worse conditioned than at 180 cells (BASELINE
    configs[1] and [3]; cond(J) grows with nlat^2): from the oracle's state at steps 10 / 20 / 40-60 of the freeze-up
    the GPU takes one step and its T0 is put into the reference's residual function (a transcription of T0eq,
    src/miz.jl:33-45, evaluated by the checker in fp64 AND in 80-bit extended precision).

    Measured on MI355X: 1.0e-10 ... 5.6e-10 at 1440 cells and 9.7e-11 ... 4.1e-10 at 4096 cells, in fp64 and in 80-bit
    evaluation alike, and the same figures for the oracle's own T0 — twenty times inside abstol; the residual function's own
    rounding noise stays below that as well (the 80-bit and fp64 evaluations of the same T0 differ by < 1e-10)."""
    import __graft_entry__ as graft
    ld = graft.load_oracle()[1].COracle(extended=True)
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    ct = ctab(pkg, st)
    fcol = np.array([0.0, 0.5])
    state = {k: np.zeros((2, nlat)) for k in PROG + ("T0",)}
    done = 0
    for s in steps:
        coracle.miz_run(1, st.x, dict(par), st.dt, ct[done:s], np.zeros(s - done), fcol, state)
        done = s
        with make_engine(pkg, "MIZ", st, par, 2) as eng:
            eng.set_column_forcing(fcol)
            eng.set_time_table(st.t)
            eng.set_state(state)
            eng.run(s, 1)
            T0_gpu = eng.get_field("T0")
        nxt = {k: v.copy() for k, v in state.items()}
        coracle.miz_run(1, st.x, dict(par), st.dt, ct[s:s + 1], np.zeros(1), fcol, nxt)
        worst = {"gpu": 0.0, "gpu_ld": 0.0, "oracle": 0.0}
        for c in range(2):
            args = (1, st.x, dict(par), float(ct[s]), float(fcol[c]), state["h"][c], state["Ew"][c], state["phi"][c])
            worst["gpu"] = max(worst["gpu"], float(np.max(np.abs(coracle.T0eq(*args, T0_gpu[c])))))
            worst["gpu_ld"] = max(worst["gpu_ld"], float(np.max(np.abs(ld.T0eq(*args, T0_gpu[c])))))
            worst["oracle"] = max(worst["oracle"], float(np.max(np.abs(coracle.T0eq(*args, nxt["T0"][c])))))
        record_error(f"|T0eq(T0_gpu)| at {nlat} cells, step {s}: 80-bit evaluation", "T0eq", worst["gpu_ld"], 1e-8)
        record_error(f"|T0eq(T0_gpu)| at {nlat} cells, step {s}: fp64 evaluation (oracle's own T0: {worst['oracle']:.2e})", "T0eq",
                     worst["gpu"], 1e-8)
        assert worst["gpu_ld"] <= 1e-8 and worst["gpu"] <= 1e-8, (s, worst)
        assert np.any(state["phi"] > 0)


@pytest.mark.parametrize("kind,nlat,nt", [("identity", 1024, 131072), ("sin", 4096, 1048576)])
def test_t0_solve_accuracy_extended_precision(pkg, oracle, coracle, kind, nlat, nt):
    """The T0 system of one step, solved in 80-bit extended precision on the host (Thomas on
    the oracle's coefficients, converged active set), is the truth both the GPU and the oracle
    approximate: the GPU's partition + cyclic-reduction solve must be as accurate as the
    oracle's fp64 Thomas (same order of magnitude), and both within cond*eps."""
    st = pkg.SpaceTime(kind, nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    kid = 0 if kind == "identity" else 1
    state = {k: np.zeros((1, nlat)) for k in PROG + ("T0",)}
    ct = ctab(pkg, st)
    spin = 60
    coracle.miz_run(kid, st.x, dict(par), st.dt, ct[:spin], np.zeros(spin), np.array([1.0]), state)
    with make_engine(pkg, "MIZ", st, par) as eng:
        eng.set_state(state)
        eng.set_column_forcing(np.array([1.0]))
        eng.set_time_table(st.t)
        eng.run(spin, 1)
        T0_gpu = eng.get_field("T0")[0]
    pre = {k: v.copy() for k, v in state.items()}
    coracle.miz_run(kid, st.x, dict(par), st.dt, ct[spin:spin + 1], np.zeros(1), np.array([1.0]), state)
    T0_cpu = state["T0"][0]
    # assemble the converged linear system exactly as oracle.solve_T0 does, solve it in longdouble
    p = dict(par)
    geom = oracle.DiffusionGeometry(kind, st.x, p["D"])
    h, Ew, phi = pre["h"][0], pre["Ew"][0], pre["phi"][0]
    with np.errstate(all="ignore"):
        Tw = oracle.water_temp(Ew, phi, p)
    Tw = np.where(np.isnan(Tw), 0.0, Tw)
    hp = np.where(h == 0, p["hmin"], h)
    L = np.longdouble
    dd = L(p["k"]) / hp.astype(L) + L(p["B"])
    r = (1 - phi.astype(L)) * (Tw.astype(L) - L(p["Tm"]))
    rp = np.concatenate((r[1:], [L(0)])); rm = np.concatenate(([L(0)], r[:-1]))
    x = st.x.astype(L)
    S = L(p["S0"]) - L(p["S1"]) * x * L(ct[spin]) - L(p["S2"]) * x * x
    rhs = L(p["ai"]) * S - L(p["A"]) + (geom.lo * rm + geom.di * r + geom.up * rp) + L(1.0)
    sset = T0_cpu < p["Tm"]
    g = np.where(sset, phi, 0.0).astype(L)
    gp = np.concatenate((g[1:], [L(0)])); gm = np.concatenate(([L(0)], g[:-1]))
    a, c, b, d = geom.lo * gm, geom.up * gp, geom.di * g - dd, -rhs
    cp = np.zeros(nlat, L); dp = np.zeros(nlat, L)
    cp[0] = c[0] / b[0]; dp[0] = d[0] / b[0]
    for i in range(1, nlat):
        den = b[i] - a[i] * cp[i - 1]
        cp[i] = c[i] / den; dp[i] = (d[i] - a[i] * dp[i - 1]) / den
    v = np.zeros(nlat, L); v[-1] = dp[-1]
    for i in range(nlat - 2, -1, -1):
        v[i] = dp[i] - cp[i] * v[i + 1]
    truth = (v + L(p["Tm"])).astype(np.float64)
    assert np.array_equal(truth < p["Tm"], sset)
    e_gpu = float(np.max(np.abs(T0_gpu - truth) / np.maximum(1.0, np.abs(truth))))
    e_cpu = float(np.max(np.abs(T0_cpu - truth) / np.maximum(1.0, np.abs(truth))))
    bound = 1e-12                                            # measured 9.2e-14 (1024), 7.1e-14 (4096)
    record_error(f"T0 vs 80-bit solve {kind} {nlat} (gpu)", "T0", e_gpu, bound)
    record_error(f"T0 vs 80-bit solve {kind} {nlat} (oracle)", "T0", e_cpu, bound)
    assert e_gpu <= bound and e_cpu <= bound, (e_gpu, e_cpu)
    assert e_gpu <= 20 * e_cpu + 1e-14, (e_gpu, e_cpu)


# ---- sizes, raggedness, many columns ----------------------------------------------------------
@pytest.mark.parametrize("kind,nlat,ncol,nt,spin,nsteps,measured", [
    ("sin", 2, 1, 100, 0, 5, 1.8e-16),             # smallest legal grid
    ("sin", 63, 3, 2000, 5, 20, 2.8e-13),          # less than one wave of chunks
    ("sin", 255, 2, 8000, 10, 30, 1.4e-11),        # odd nlat: pitch padding, ragged last chunk
    ("identity", 257, 2, 8000, 10, 30, 5.7e-11),
    ("sin", 1000, 5, 60000, 20, 20, 6.7e-12),      # 256 threads, ragged
    ("sin", 1440, 2, 131072, 50, 20, 2.3e-12),     # BASELINE configs[1]: 1-D MIZ, 1440 bands
    ("sin", 1101, 3, 100000, 30, 20, 1.6e-12),      # two cells per thread: 768 threads for 551 chunks (padding waves)
    ("identity", 1024, 8, 262144, 50, 20, 1.2e-13),  # nt: 2x the explicit stability limit cw dx^2/(2D)
    ("sin", 4096, 6, 1048576, 50, 10, 2.6e-13),    # BASELINE configs[3] meridian length, 1024 threads: the maximum
    ("sin", 4093, 2, 1048576, 20, 5, 1.1e-13),     # ragged at the maximum workgroup size
])
def test_miz_sizes_vs_oracle(pkg, coracle, kind, nlat, ncol, nt, spin, nsteps, measured, cells):
    """Spin up with the oracle (ice edge, open water and T0 solve all live), hand the state to
    the GPU, advance both, compare.  Per-column forcing differs per column."""
    if cells == 2 and nlat > 1536:
        pytest.skip("two cells per thread exist up to 1536-cell meridians")
    st = pkg.SpaceTime(kind, nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    kid = 0 if kind == "identity" else 1
    fcol = 2.0 * np.sin(2 * np.pi * (np.arange(ncol) + 0.3) / ncol)
    state = {k: np.zeros((ncol, nlat)) for k in PROG + ("T0",)}
    ct = ctab(pkg, st)
    if spin:
        coracle.miz_run(kid, st.x, dict(par), st.dt, ct[:spin], np.zeros(spin), fcol, state)
    with make_engine(pkg, "MIZ", st, par, ncol) as eng:
        assert eng.launch_info()["cells_per_thread"] == cells
        eng.set_state(state)
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(spin, nsteps)
        got = eng.get_state(ALL)
        cnt = eng.counters()
    diag, ocnt = coracle.miz_run(kid, st.x, dict(par), st.dt, ct[spin:spin + nsteps], np.zeros(nsteps), fcol, state)
    ref = dict(state)
    ref.update(diag)
    check_all(got, ref, min(TOL_SHORT, 10.0 * measured), what=f"{kind} {nlat}x{ncol}")
    assert cnt["solves"] == ocnt[0] and cnt["cap_hits"] == 0     # same active-set iteration path


@pytest.mark.parametrize("model", ["MIZ", "Classic"])
def test_graph_replay_equals_direct_launches(pkg, monkeypatch, model):
    """ebm_run replays a captured hipGraph of 64 step launches on launch-bound shapes (per-step
    scalars through a device table).  Same launches, same arguments: bitwise identical state, with
    a varying forcing and a run length that exercises replay + remainder + diagnostic last step."""
    g = load_golden("classic_identity_180_2000.npz")
    st = pkg.SpaceTime("identity" if model == "Classic" else "sin", 180, 2000, 1)
    par = pkg.default_parameters(model)
    nsteps = 64 * 3 + 17
    f_steps = 0.3 * np.sin(np.arange(nsteps) / 7.0)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("EBM_GRAPH", mode)
        with make_engine(pkg, model, st, par, 2) as eng:
            if model == "Classic":
                eng.set_state(dict(E=np.tile(g["s0_E"], (2, 1)), Tg=np.tile(g["s0_Tg"], (2, 1))))
            eng.set_column_forcing(np.array([0.0, 1.0]))
            eng.set_time_table(st.t)
            eng.run(1990, nsteps, f_steps, True)          # wraps around the year end
            out[mode] = eng.get_state()
            assert eng.counters()["steps"] == nsteps
    for k in out["0"]:
        assert np.array_equal(out["0"][k], out["1"][k], equal_nan=True), k


@pytest.mark.parametrize("nlat,ncol,nt", [(4096, 520, 1048576), (2048, 600, 262144)])
def test_l2_prefetch_does_not_change_results(pkg, monkeypatch, nlat, ncol, nt):
    """Long meridians (one or two workgroups per CU): every workgroup prefetches into L2 the inputs
    of the workgroup `EBM_PREFETCH_COLS` columns ahead (default: CU count x workgroups per CU) with
    LDS-DMA loads whose data is discarded.  Off, default and an odd distance: bitwise identical state."""
    nsteps = 12
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    fcol = 2.0 * np.sin(np.arange(ncol) / 9.0)
    out = {}
    for mode in ("0", None, "7"):
        if mode is None:
            monkeypatch.delenv("EBM_PREFETCH_COLS", raising=False)
        else:
            monkeypatch.setenv("EBM_PREFETCH_COLS", mode)
        with make_engine(pkg, "MIZ", st, par, ncol) as eng:
            eng.set_column_forcing(fcol)
            eng.set_time_table(st.t)
            eng.run(0, nsteps, None, True)
            out[mode] = eng.get_state()
    for mode in (None, "7"):
        for k in out["0"]:
            assert np.array_equal(out["0"][k], out[mode][k], equal_nan=True), (mode, k)
    assert np.any(out["0"]["Ew"] != 0.0)


def test_unsupported_and_bad_arguments(pkg):
    st = pkg.SpaceTime("sin", 4097, 100, 1)                 # one workgroup owns a meridian: <= 4096 cells
    with pytest.raises(pkg.EBMError, match="not supported"):
        make_engine(pkg, "MIZ", st, pkg.default_parameters("MIZ"))
    with pytest.raises(pkg.EBMError, match="not supported"):
        make_engine(pkg, "Classic", st, pkg.default_parameters("Classic"))
    st = pkg.SpaceTime("sin", 64, 100, 1)
    with make_engine(pkg, "MIZ", st, pkg.default_parameters("MIZ")) as eng:
        with pytest.raises(pkg.EBMError, match="not part of this model"):
            eng.get_field("Tg")
        with pytest.raises(pkg.EBMError, match="ebm_set_time_table"):
            eng.run(0, 1)
        with pytest.raises(ValueError):
            eng.set_field("Ei", np.zeros(63))
    par = pkg.default_parameters("MIZ")
    par.Tm = -1.0                                            # Tm^1.36: DomainError in Julia
    with pytest.raises(pkg.EBMError, match="DomainError"):
        make_engine(pkg, "MIZ", st, par)


# ---- edge cases of the state ------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(24))
def test_randomized_states_one_step(pkg, coracle, seed, monkeypatch):
    """Seeded fuzz: random grid length and kind, random (physically loose) states mixing open
    water, thin and thick ice, phi = 0 / 1 / in between, floes at Dmin / Dmax / 0, inconsistent
    Ei, random warm-start signs, random forcing and time of year, perturbed parameters.  One step
    on the GPU against the oracle from identical inputs; sentinels must coincide."""
    monkeypatch.setenv("EBM_CELLS_PER_THREAD", "2" if seed % 2 else "4")      # both launch geometries
    rng = np.random.default_rng(1000 + seed)
    nlat = int(rng.choice([2, 3, 5, 17, 64, 65, 127, 180, 256, 300, 511, 777]))
    ncol = int(rng.integers(1, 5))
    kind = "sin" if rng.random() < 0.6 else "identity"
    nt = int(max(2000, 0.7 * nlat * nlat))                    # near the explicit stability limit
    st = pkg.SpaceTime(kind, nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    for k in ("D", "A", "B", "S1", "a0", "ai", "Fb", "k", "m1", "rl", "kappa"):
        par[k] = par[k] * float(rng.uniform(0.8, 1.25))
    shape = (ncol, nlat)
    ice = rng.random(shape) < 0.6
    h = np.where(ice, rng.choice([par["hmin"], 0.3, 1.0, 3.0], size=shape) * rng.uniform(0.5, 1.5, shape), 0.0)
    phi = np.where(ice, rng.choice([0.0, 0.05, 0.5, 0.995, 1.0], size=shape), 0.0)
    D = np.where(ice, rng.choice([0.0, par["Dmin"], 10.0, par["Dmax"]], size=shape), 0.0)
    Ei = -par["Lf"] * h * phi * np.where(rng.random(shape) < 0.8, 1.0, rng.uniform(0.0, 2.0, shape))
    Ew = par["cw"] * rng.uniform(-0.5, 12.0, shape) * (1.0 - 0.9 * phi)
    if seed % 6 != 5:
        # a saturated cell (phi == 1) with water enthalpy left in it has Tw = Ew/0 = Inf, which turns its
        # whole column's T0 system into NaNs: kept in every sixth seed (NaN propagation), removed in the
        # others so that the comparison is one of finite numbers (0/0 -> NaN -> 0, src/miz.jl:157)
        Ew = np.where(phi == 1.0, 0.0, Ew)
    T0 = rng.uniform(-20.0, 5.0, shape)
    state = dict(Ei=Ei, Ew=Ew, h=h, D=D, phi=phi, T0=T0)
    state = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in state.items()}
    fcol = rng.uniform(-5.0, 5.0, ncol)
    ti = int(rng.integers(0, nt))
    ct = ctab(pkg, st)[ti:ti + 1]
    f_step = np.array([float(rng.uniform(-2.0, 2.0))])
    with make_engine(pkg, "MIZ", st, par, ncol) as eng:
        eng.set_state(state)
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(ti, 1, f_step, True)
        got = eng.get_state(ALL)
        cnt = eng.counters()
    kid = 0 if kind == "identity" else 1
    diag, ocnt = coracle.miz_run(kid, st.x, dict(par), st.dt, ct, f_step, fcol, state)
    ref = dict(state)
    ref.update(diag)
    for k in ALL:
        assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])), f"{k}: NaN pattern (seed {seed})"
    finite = float(np.mean(np.isfinite(ref["T0"])))
    if seed % 6 != 5:
        assert finite == 1.0, finite                              # a comparison of numbers, not of NaN patterns
    check_all(got, ref, 1.2e-11, what=f"seed {seed} {kind} {nlat}x{ncol} finite {finite:.2f}")   # measured <= 1.2e-12
    assert cnt["solves"] == ocnt[0], (cnt, ocnt)


def test_nan_inf_and_saturated_states(pkg, coracle):
    """phi == 1 (division by zero in water_temp: 0/0 -> NaN -> 0, x/0 -> Inf kept), h == 0 with
    phi != 0, D == 0 with ice, NaN in a prognostic: the sentinels and non-finite values must
    propagate exactly as in the oracle (semantics traps)."""
    nlat, ncol = 64, 4
    st = pkg.SpaceTime("sin", nlat, 4000, 1)
    par = pkg.default_parameters("MIZ")
    rng = np.random.default_rng(7)
    state = {k: np.zeros((ncol, nlat)) for k in PROG + ("T0",)}
    phi = rng.random((ncol, nlat))
    phi[:, 40:] = 1.0                                       # saturated
    phi[:, :10] = 0.0
    h = np.where(phi > 0, 0.1 + rng.random((ncol, nlat)), 0.0)
    state["phi"], state["h"] = phi, h
    state["Ei"] = -9.5 * h * phi
    state["Ew"] = np.where(phi < 1, rng.random((ncol, nlat)) * 2, 0.0)
    state["Ew"][1, 45] = 0.3                                # Ew/0 -> +Inf water temperature
    state["D"] = np.where(phi > 0, 1.0 + 50 * rng.random((ncol, nlat)), 0.0)
    state["D"][2, 20] = 0.0                                 # ice with D == 0
    state["h"][2, 25] = 0.0                                 # phi != 0 but h == 0
    state["Ei"][3, 30] = np.nan                             # NaN prognostic
    state["T0"] = -5.0 * phi
    ct = ctab(pkg, st)
    with make_engine(pkg, "MIZ", st, par, ncol) as eng:
        eng.set_state(state)
        eng.set_time_table(st.t)
        eng.run(1000, 1)
        got = eng.get_state(ALL)
    diag, _ = coracle.miz_run(1, st.x, dict(par), st.dt, ct[1000:1001], np.zeros(1), None, state)
    ref = dict(state)
    ref.update(diag)
    for k in ALL:
        a, b = got[k], ref[k]
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        assert np.array_equal(np.isinf(a), np.isinf(b)), k
        fin = np.isfinite(b)
        # columns 0 and 2 are finite everywhere; a NaN/Inf cell contaminates its column's solve
        assert scaled_err(np.where(fin, a, 0.0), np.where(fin, b, 0.0)) <= 1e-9, k


def test_columns_are_independent_and_deterministic(pkg):
    """Replicated columns give bitwise identical results (no cross-column coupling, no
    dependence on workgroup placement); two runs are bitwise identical."""
    nlat, ncol = 512, 96
    st = pkg.SpaceTime("sin", nlat, 16384, 1)
    par = pkg.default_parameters("MIZ")
    fcol = np.tile(np.linspace(-2, 2, 8), ncol // 8)
    outs = []
    for _ in range(2):
        with make_engine(pkg, "MIZ", st, par, ncol) as eng:
            eng.set_column_forcing(fcol)
            eng.set_time_table(st.t)
            eng.run(0, 60)
            outs.append(eng.get_state(ALL))
    for k in ALL:
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k
        a = outs[0][k].reshape(ncol // 8, 8, nlat)
        assert all(np.array_equal(a[0], a[i], equal_nan=True) for i in range(1, ncol // 8)), k


# ---- classic model ----------------------------------------------------------------------------
def test_classic_matches_golden(pkg, cells):
    g = load_golden("classic_identity_180_2000.npz")
    st = pkg.SpaceTime("identity", 180, 2000, 1)
    with make_engine(pkg, "Classic", st, pkg.default_parameters("Classic")) as eng:
        eng.set_state(dict(E=g["s0_E"][None], Tg=g["s0_Tg"][None]))
        eng.set_time_table(st.t)
        done = 0
        for s in (1, 2, 10, 522, 2000):
            eng.run(done, s - done)
            done = s
            got = {k: v[0] for k, v in eng.get_state(("E", "Tg", "T", "h")).items()}
            check_all(got, {k: g[f"s{s}_{k}"] for k in ("E", "Tg", "T", "h")},
                      TOL_SHORT if s <= 10 else TOL_YEAR, names=("E", "Tg", "T", "h"), what=f"classic step {s}")


@pytest.mark.parametrize("nlat,ncol", [(1024, 16), (333, 3)])
def test_classic_sizes_vs_oracle(pkg, coracle, nlat, ncol, cells):
    """BASELINE configs[2] shape (1024 latitudes; columns = longitudes with perturbed forcing)."""
    st = pkg.SpaceTime("identity", nlat, 2000, 1)
    par = pkg.default_parameters("Classic")
    Ts = 30.0 - 45.0 * st.x ** 2
    E0 = np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5)
    state = dict(E=np.tile(E0, (ncol, 1)), Tg=np.tile(Ts, (ncol, 1)))
    fcol = 0.5 * np.sin(2 * np.pi * np.arange(ncol) / ncol)
    ct = ctab(pkg, st)
    nsteps = 40
    with make_engine(pkg, "Classic", st, par, ncol) as eng:
        eng.set_state(state)
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(0, nsteps)
        got = eng.get_state(("E", "Tg", "T", "h"))
    idx = np.arange(nsteps)
    out = coracle.classic_run(st.x, dict(par), st.dt, ct[idx], ct[(idx + 1) % st.nt], np.zeros(nsteps), fcol, state)
    ref = dict(state)
    ref.update(out)
    measured = {1024: 8.2e-12, 333: 5.8e-14}[nlat]
    check_all(got, ref, 10.0 * measured, names=("E", "Tg", "T", "h"), what=f"classic {nlat}x{ncol}")


@pytest.mark.parametrize("seed", range(8))
def test_classic_randomized_states_one_step(pkg, coracle, seed, monkeypatch):
    monkeypatch.setenv("EBM_CELLS_PER_THREAD", "2" if seed % 2 else "4")
    """Seeded fuzz of the classic step: random length, random enthalpies around zero (so that the
    Bool masks E > 0, E < 0, E >= 0, T0 < 0 all flip within a column, including E == 0 exactly),
    random ghost layer, forcing and time index."""
    rng = np.random.default_rng(77 + seed)
    nlat = int(rng.choice([2, 9, 64, 129, 180, 500, 1024]))
    ncol = int(rng.integers(1, 4))
    st = pkg.SpaceTime("identity", nlat, 2000, 1)
    par = pkg.default_parameters("Classic")
    E = par["cw"] * rng.uniform(-8.0, 25.0, (ncol, nlat))
    E[rng.random((ncol, nlat)) < 0.05] = 0.0
    Tg = rng.uniform(-25.0, 30.0, (ncol, nlat))
    state = dict(E=np.ascontiguousarray(E), Tg=np.ascontiguousarray(Tg))
    fcol = rng.uniform(-3.0, 3.0, ncol)
    ti = int(rng.integers(0, st.nt))
    ct = ctab(pkg, st)
    with make_engine(pkg, "Classic", st, par, ncol) as eng:
        eng.set_state(state)
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(ti, 1, None, True)
        got = eng.get_state(("E", "Tg", "T", "h"))
    out = coracle.classic_run(st.x, dict(par), st.dt, ct[[ti]], ct[[(ti + 1) % st.nt]], np.zeros(1), fcol, state)
    ref = dict(state)
    ref.update(out)
    for k in ("E", "T", "h"):                                  # pointwise physics: bit-exact
        assert np.array_equal(got[k], ref[k], equal_nan=True), f"{k} (seed {seed})"
    record_error(f"classic fuzz seed {seed} nlat {nlat}", "Tg", scaled_err(got["Tg"], ref["Tg"]), 5e-14)
    assert scaled_err(got["Tg"], ref["Tg"]) <= 5e-14, f"Tg (seed {seed})"      # measured <= 5.1e-15


def test_device_division_is_ieee(pkg):
    """The physics is a bit-exact restatement only if the device's fp64 division is IEEE: the
    kernels use a rcp + Newton + residual + v_div_fixup sequence without the exponent rescaling of
    the compiler's expansion.  It must equal host division bit for bit for every operand whose
    magnitude (and quotient) is within 2^+-500, and for zero / infinite / NaN operands."""
    from energybalancemodel_jl_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    n = 1 << 20
    a = rng.normal(size=n) * 2.0 ** rng.integers(-250, 250, n)
    b = rng.normal(size=n) * 2.0 ** rng.integers(-250, 250, n)
    # near-halfway and physically typical operands
    a[:1000] = 1.0 + rng.integers(0, 1 << 20, 1000) * 2.0 ** -52
    b[:1000] = 3.0 + rng.integers(0, 1 << 20, 1000) * 2.0 ** -51
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 9.5, 1e-300, 1e300])
    aa, bb = np.meshgrid(special, special)
    a[1000:1100], b[1000:1100] = aa.ravel(), bb.ravel()
    q = np.empty(n)
    _lib.check(lib.ebm_selftest_divide(0, n, _lib.dptr(a), _lib.dptr(b), _lib.dptr(q)), "ebm_selftest_divide")
    with np.errstate(all="ignore"):
        ref = a / b
    moderate = np.ones(n, bool)
    moderate[1000:1100] = ~((np.abs(aa.ravel()) == 1e-300) | (np.abs(aa.ravel()) == 1e300) |
                            (np.abs(bb.ravel()) == 1e-300) | (np.abs(bb.ravel()) == 1e300))
    same = (q.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(q) & np.isnan(ref))
    assert same[moderate].all(), int((~same[moderate]).sum())


def test_classic_on_sin_grid_uses_uniform_operator(pkg, coracle):
    """get_statics builds kappa from get_diffop(nx) whatever the grid type (src/classic.jl:21): on
    SpaceTime{sin} the classic model still diffuses with the uniform-x operator while x enters the
    insolation and co-albedo.  Same here."""
    st = pkg.SpaceTime("sin", 200, 2000, 1)
    par = pkg.default_parameters("Classic")
    Ts = 30.0 - 45.0 * st.x ** 2
    state = dict(E=np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5)[None].copy(), Tg=Ts[None].copy())
    ct = ctab(pkg, st)
    nsteps = 60
    with make_engine(pkg, "Classic", st, par, 1) as eng:
        eng.set_state(state)
        eng.set_time_table(st.t)
        eng.run(0, nsteps)
        got = eng.get_state(("E", "Tg", "T", "h"))
    idx = np.arange(nsteps)
    out = coracle.classic_run(st.x, dict(par), st.dt, ct[idx], ct[(idx + 1) % st.nt], np.zeros(nsteps), None, state)
    ref = dict(state)
    ref.update(out)
    check_all(got, ref, TOL_SHORT, names=("E", "Tg", "T", "h"), what="classic on sin grid")


# ---- the reference's operator surface ------------------------------------------------------------
def test_step_bang_surface(pkg):
    """step!(Val(:MIZ), t, f, vars, st, par) called directly, 10 times from the zero state
    (hidden T0 warm start persisting between calls), equals golden step 10."""
    g = load_golden("miz_sin_180_2000.npz")
    pkg.reset_step_state()
    st = pkg.SpaceTime("sin", 180, 2000, 1)
    par = pkg.default_parameters("MIZ")
    vars_ = pkg.Collection({k: np.zeros(180) for k in PROG})
    for ti in range(10):
        out = pkg.step_("MIZ", float(st.t[ti]), 0.0, vars_, st, par)
        assert out is vars_
    assert set(vars_.propertynames()) == set(pkg.MIZ_SOLVARS)
    check_all(vars_, {k: g[f"s10_{k}"] for k in PROG + DIAG}, TOL_SHORT, names=PROG + DIAG, what="step_ x10")
    pkg.reset_step_state()


def test_step_bang_classic(pkg):
    g = load_golden("classic_identity_180_2000.npz")
    st = pkg.SpaceTime("identity", 180, 2000, 1)
    par = pkg.default_parameters("Classic")
    vars_ = pkg.Collection(E=g["s0_E"].copy(), Tg=g["s0_Tg"].copy())
    for ti in range(2):
        pkg.step_("Classic", float(st.t[ti]), 0.0, vars_, st, par)
    check_all(vars_, {k: g[f"s2_{k}"] for k in ("E", "Tg", "T", "h")}, TOL_SHORT, names=("E", "Tg", "T", "h"))
    pkg.reset_step_state()


@pytest.mark.parametrize("lastonly", [True, False])
def test_integrate_surface_matches_oracle(pkg, oracle, lastonly):
    """integrate(:MIZ, ...) -> Solutions: raw / seasonal.{winter,summer,avg} / ts laid out and
    filled as savesol! does (src/infrastructure.jl:549-591).  Identity grid, 2 years, varying
    forcing."""
    st = pkg.SpaceTime("identity", 90, 500, 2)
    ost = oracle.SpaceTime("identity", 90, 500, 2)
    par = pkg.default_parameters("MIZ")
    init = pkg.Collection({k: np.zeros(90) for k in PROG})
    forcing = pkg.Forcing(0.0, 2.0, 0.0, (1, 0), (2.0, -2.0))
    oforcing = oracle.Forcing(0.0, 2.0, 0.0, (1, 0), (2.0, -2.0))
    sols = pkg.integrate("MIZ", st, forcing, par, init, lastonly=lastonly)
    ref = oracle.integrate("MIZ", ost, oforcing, dict(par), dict(init), lastonly=lastonly)
    assert np.array_equal(sols.ts, ref.ts)
    assert set(sols.raw.propertynames()) == set(oracle.MIZ_SOLVARS)
    for v in oracle.MIZ_SOLVARS:
        assert sols.raw[v].shape == (len(ref.ts), 90)
        assert scaled_err(sols.raw[v], np.stack(ref.raw[v])) <= 1e-8, v
        for y in range(2):
            assert scaled_err(sols.seasonal.winter[v][y], ref.winter[v][y]) <= 1e-8, v
            assert scaled_err(sols.seasonal.summer[v][y], ref.summer[v][y]) <= 1e-8, v
            assert scaled_err(sols.seasonal.avg[v][y], ref.avg[v][y]) <= 1e-8, v
    assert sols.counters["steps"] == 1000


def test_integrate_seasonal_snapshots_only(pkg):
    """ebm_integrate with only the winter/summer snapshots requested runs the steps in between
    without the diagnostic stores; snapshots and final state are bitwise those of the full run."""
    st = pkg.SpaceTime("sin", 180, 2000, 1)
    par = pkg.default_parameters("MIZ")
    names = ("Ei", "Ew", "h", "D", "phi", "Tw", "Ti", "n", "E", "T")
    out, final = {}, {}
    for mode in ("full", "seasonal"):
        with make_engine(pkg, "MIZ", st, par, 2) as eng:
            eng.set_column_forcing(np.array([0.0, 1.5]))
            eng.set_time_table(st.t)
            out[mode] = eng.integrate(st.nt, 1, None, True, st.winter.inx, st.summer.inx, names,
                                      want_raw=(mode == "full"), want_avg=(mode == "full"))
            final[mode] = eng.get_state()
    assert out["seasonal"]["raw"] is None and out["seasonal"]["avg"] is None
    for k in ("winter", "summer"):
        assert np.array_equal(out["full"][k], out["seasonal"][k], equal_nan=True), k
    for k in final["full"]:
        assert np.array_equal(final["full"][k], final["seasonal"][k], equal_nan=True), k
    assert np.array_equal(out["full"]["raw"][:, st.winter.inx - 1], out["full"]["winter"][:, 0], equal_nan=True)


def test_column_schedules_match_per_column_oracle(pkg, coracle):
    """every column its own Forcing{false}, evaluated on the device at the
    model time st.T[tinx] of each step.  Three members (constant, two different ramps), 3 years
    on the identity grid in two ebm_run calls (time continuity across calls and graph replays);
    the oracle runs each member alone with the host-evaluated forcing series."""
    nlat, nt, dur = 90, 500, 3
    st = pkg.SpaceTime("identity", nlat, nt, dur)
    par = pkg.default_parameters("MIZ")
    members = [pkg.Forcing(0.5), pkg.Forcing(0.0, 2.0, 0.0, (1, 0), (2.0, -2.0)),
               pkg.Forcing(-1.0, 1.0, 0.0, (0, 1), (2.0, -1.0))]
    total = nt * dur
    with make_engine(pkg, "MIZ", st, par, len(members)) as eng:
        eng.set_time_table(st.t)
        eng.set_column_schedules(members)
        eng.run(0, 777, None, False)
        eng.run(777, total - 777, None, True)
        got = eng.get_state(ALL)
    ct = np.array([pkg.cos2pit(float(st.t[i % nt])) for i in range(total)])
    for c, forcing in enumerate(members):
        f_steps = np.array([forcing(float(T)) for T in st.T])
        assert len(f_steps) == total
        state = {k: np.zeros((1, nlat)) for k in PROG + ("T0",)}
        diag, _ = coracle.miz_run(0, st.x, dict(par), st.dt, ct, f_steps, None, state)
        ref = dict(state, **diag)
        check_all({k: got[k][c] for k in ALL}, {k: ref[k][0] for k in ALL}, TOL_YEAR, what=f"member {c}")
    # the members really differ
    assert scaled_err(got["E"][1], got["E"][0]) > 1e-3 and scaled_err(got["E"][2], got["E"][1]) > 1e-3


def test_step_clock_and_schedule_with_single_steps(pkg):
    """ebm_step evaluates the schedules at the handle's step clock: single steps from a set
    clock equal the same steps issued through ebm_run, bit for bit."""
    st = pkg.SpaceTime("sin", 180, 2000, 2)
    par = pkg.default_parameters("MIZ")
    members = [pkg.Forcing(0.0, 2.0, 0.0, (1, 0), (2.0, -2.0)), pkg.Forcing(1.0)]
    out = {}
    for mode in ("run", "step"):
        with make_engine(pkg, "MIZ", st, par, 2) as eng:
            eng.set_time_table(st.t)
            eng.set_column_schedules(members)
            first = 2100                                   # second year: member 0 is on its ramp
            if mode == "run":
                eng.run(first, 7, None, True)
            else:
                eng.set_step_clock(first)
                for i in range(7):
                    ti = (first + i) % st.nt
                    eng.step(eng.ttab[ti], eng.ttab[(ti + 1) % st.nt], 0.0, True)
            out[mode] = eng.get_state()
    for k in out["run"]:
        assert np.array_equal(out["run"][k], out["step"][k], equal_nan=True), k
    assert np.any(out["run"]["Ew"][0] != out["run"]["Ew"][1])


def test_forcing_set_after_graph_capture_is_honoured(pkg, monkeypatch):
    """A per-column forcing installed after ebm_run has captured its launch graph must reach the
    kernels (the captured nodes hold the old argument values: the graph is rebuilt)."""
    st = pkg.SpaceTime("sin", 180, 2000, 1)
    par = pkg.default_parameters("MIZ")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("EBM_GRAPH", mode)
        with make_engine(pkg, "MIZ", st, par, 2) as eng:
            eng.set_time_table(st.t)
            eng.run(0, 200, None, False)
            eng.set_column_forcing(np.array([0.0, 3.0]))
            eng.run(200, 200, None, True)
            out[mode] = eng.get_state()
    for k in out["0"]:
        assert np.array_equal(out["0"][k], out["1"][k], equal_nan=True), k
    assert np.any(out["1"]["Ew"][0] != out["1"]["Ew"][1])


def test_hemispheric_mean_on_device_is_bit_exact(pkg):
    """ebm_hemispheric_mean reduces in the reference's order (src/utilities.jl:397-403): equal,
    bit for bit, to the sequential host loop on the downloaded field — values, NaNs and all."""
    for nlat, ncol in ((180, 3), (1000, 5), (4096, 2)):
        st = pkg.SpaceTime("sin", nlat, 2000 if nlat == 180 else 1048576, 1)
        par = pkg.default_parameters("MIZ")
        with make_engine(pkg, "MIZ", st, par, ncol) as eng:
            eng.set_column_forcing(np.linspace(-2.0, 2.0, ncol))
            eng.set_time_table(st.t)
            eng.run(0, 40, None, True)
            for name in ("T", "phi", "Ti"):                # Ti carries NaN sentinels
                field = eng.get_field(name)
                got = eng.hemispheric_mean(name)
                ref = np.empty(ncol)
                for c in range(ncol):
                    acc = 0.0
                    for i in range(nlat - 1):
                        acc += (field[c, i] + field[c, i + 1]) * (st.x[i + 1] - st.x[i]) / 2.0
                    ref[c] = acc
                assert np.array_equal(got, ref, equal_nan=True), (nlat, name)
                assert np.array_equal(got, pkg.hemispheric_mean(field, st.x), equal_nan=True)


def test_integrate_classic_surface(pkg, oracle):
    g = load_golden("classic_identity_180_2000.npz")
    st = pkg.SpaceTime("identity", 180, 2000, 1)
    init = pkg.Collection(E=g["s0_E"].copy(), Tg=g["s0_Tg"].copy())
    sols = pkg.integrate("Classic", st, pkg.Forcing(0.0), pkg.default_parameters("Classic"), init)
    assert set(sols.raw.propertynames()) == {"E", "T", "h"}          # src/infrastructure.jl:621
    for v in ("E", "T", "h"):
        assert scaled_err(sols.raw[v][9], g[f"s10_{v}"]) <= TOL_SHORT
        assert scaled_err(sols.raw[v][1999], g[f"s2000_{v}"]) <= TOL_YEAR
        assert scaled_err(sols.seasonal.winter[v][0], sols.raw[v][521]) == 0.0


# ---- BASELINE.json full size: size-independent properties ---------------------------------------
def test_full_size_4096x2048_properties(pkg, oracle, coracle):
    """configs[3] at full size (4096 x 2048, 384 MiB of state): (1) 32 sampled columns agree with the oracle at steps
    10, 20 and 40 from the zero state — the prognostics, which carry the run, to 1e-12; T0, Ti and the temperatures
    formed from them within what the columns' own sensitivity dT0/dphi makes of the observed difference in phi
    and Ew (t0_error_explained_by_state: during the first tens of steps T0 at the advancing ice edge depends on the
    concentration of the newly frozen cells with a factor 1e4 ... 1e6 — the fp64 oracle itself is 1.6e-8 from its own 80-bit
    evaluation there), (2) columns with equal forcing are bitwise equal, (3) the
    active-set iteration never hits its cap."""
    nlat, ncol, nt = 4096, 2048, 1048576
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    fcol = 0.5 * np.sin(2.0 * np.pi * (np.arange(ncol) % 64) / 64)       # 32 replicas of 64 forcings
    sample = np.arange(0, 64, 2)
    state = {k: np.zeros((len(sample), nlat)) for k in PROG + ("T0",)}
    ct = ctab(pkg, st)
    SOLVE = ("T0", "Ti", "T")             # what the T0 solve enters directly; everything else only through the dynamics
    # Bar for T0 / Ti / T: 4 x what the columns' own sensitivity makes of the observed state difference, + 1e-12.  The stated
    # constant 4 covers that the state a step ENDS with stands in for the one it began with (on the CPU, oracle against
    # its 80-bit build, the observed T0 difference is at most 1.8 x the expression); the ratios on MI355X are recorded.
    CONST = 4.0
    with make_engine(pkg, "MIZ", st, par, ncol) as eng:
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        done = 0
        for s_ in (10, 20, 40):
            eng.run(done, s_ - done)
            got = eng.get_state(ALL)
            diag, _ = coracle.miz_run(1, st.x, dict(par), st.dt, ct[done:s_], np.zeros(s_ - done), fcol[sample], state)
            done = s_
            ref = dict(state)
            ref.update(diag)
            sub = {k: got[k][sample] for k in ALL}
            errs = {k: scaled_err(sub[k], ref[k]) for k in ALL}
            explained = t0_error_explained_by_state(oracle, "sin", st.x, par, ref, sub)
            bar = CONST * explained + 1e-12
            solve_err = max(errs[k] for k in SOLVE)
            other = max(errs[k] for k in ALL if k not in SOLVE)
            record_error(f"4096x2048 sample, step {s_}: T0/Ti/T; explained by the observed state difference: {explained:.2e} "
                         f"(ratio {solve_err / max(explained, 1e-300):.2f})", max(SOLVE, key=errs.get), solve_err, bar)
            record_error(f"4096x2048 sample, step {s_}: every other field", max((k for k in ALL if k not in SOLVE), key=errs.get), other, 1e-12)
            assert solve_err <= bar, (s_, solve_err, explained)
            assert other <= 1e-12, (s_, errs)
        cnt = eng.counters()
    assert cnt["cap_hits"] == 0
    for k in ALL:
        a = got[k].reshape(32, 64, nlat)
        assert np.array_equal(a[0], a[17], equal_nan=True) and np.array_equal(a[0], a[31], equal_nan=True), k


# ---- an anchor outside the oracle: the analytic Legendre-mode decay (tests/test_analytic_solutions.py) ------------
@pytest.mark.parametrize("kind,nlat,nt,nsteps,limit", [
    ("identity", 256, 32000, 3200, 0.14),          # a tenth of a year at a quarter of the explicit limit: 0.132
    ("sin", 256, 32000, 3200, 0.41),               # 0.395
    ("sin", 1024, 512000, 12800, 0.12),            # a fortieth of a year, four waves per meridian: 0.108
])
def test_explicit_step_reproduces_the_analytic_legendre_decay(pkg, kind, nlat, nt, nsteps, limit, cells):
    """Open water, no insolation, A = Fb = f = 0: cw dT/dt = D d/dx[(1-x^2) dT/dx] - B T.  One step of the reference's
    scheme multiplies the P_n component of T by 1 - n(n+1) dt D/cw - dt B/cw exactly; the HIP path's distance to that
    analytic solution is the second-order stencil's spatial error (x nlat^2: the measured constants above, the same
    as the restatement's) — a check of a7-a10 and the open-water update that involves neither oracle nor author.
    Columns carry different mode amplitudes (the problem is linear)."""
    from test_analytic_solutions import legendre_setup
    st, par, exact = legendre_setup(pkg, kind, nlat, nt)
    amp = np.array([1.0, 0.25, 2.0])
    state = {k: np.zeros((3, nlat)) for k in PROG + ("T0",)}
    state["Ew"] = par["cw"] * exact(0, amp)
    with make_engine(pkg, "MIZ", st, par, 3) as eng:
        eng.set_state(state)
        eng.set_time_table(st.t)
        eng.run(0, nsteps)
        got = eng.get_state(PROG)
    assert not got["Ei"].any() and not got["phi"].any()              # stayed open water
    err = float(np.max(np.abs(got["Ew"] / par["cw"] - exact(nsteps, amp)) / amp[:, None])) * nlat**2
    record_error(f"analytic Legendre decay, {kind} {nlat}, {nsteps} steps: error x nlat^2", "Ew/cw", err, limit)
    assert 0.5 * limit < err < limit, err


def test_step1_closed_forms_on_the_gpu(pkg, cells):
    """The closed forms of step 1 from the all-zero state (test/runtests.jl:24-31; derived from src/miz.jl:156-187 in
    SURVEY 8c), checked on the HIP path directly — no oracle in between: Tw = Ti = n = Flat = 0,
    Fvw = (a0 - a2 x^2) S - A + Fb; cells with rEw = dt Fvw < 0 freeze: Ei = rEw, Ew = 0, D = Dmin, h = hmin,
    phi = min(1, -Ei/(Lf hmin)); the others keep Ew = rEw.  Sentinels of src/miz.jl:193-194."""
    st = pkg.SpaceTime("sin", 180, 2000, 1)
    par = pkg.default_parameters("MIZ")
    with make_engine(pkg, "MIZ", st, par, 1) as eng:
        eng.set_time_table(st.t)
        eng.run(0, 1, None, True)
        out = {k: v[0] for k, v in eng.get_state(ALL).items()}
    x, ct = st.x, pkg.cos2pit(float(st.t[0]))
    S = par["S0"] - par["S1"] * x * ct - par["S2"] * x * x
    rEw = st.dt * ((par["a0"] - par["a2"] * x * x) * S - par["A"] + par["Fb"])
    frz = rEw < 0
    assert frz.any() and (~frz).any() and np.all(out["n"] == 0.0)
    np.testing.assert_allclose(out["Ei"][frz], rEw[frz], rtol=1e-13)
    np.testing.assert_allclose(out["Ew"][~frz], rEw[~frz], rtol=1e-13)
    assert np.all(out["Ew"][frz] == 0.0) and np.all(out["Ei"][~frz] == 0.0)
    assert np.all(out["D"][frz] == par["Dmin"]) and np.all(out["D"][~frz] == 0.0)
    np.testing.assert_allclose(out["h"][frz], par["hmin"], rtol=1e-13)
    np.testing.assert_allclose(out["phi"][frz], np.minimum(1.0, -rEw[frz] / (par["Lf"] * par["hmin"])), rtol=1e-12)
    assert np.all(np.isnan(out["Ti"][~frz])) and not np.isnan(out["Ti"][frz]).any()
    assert np.all(out["Tw"][out["phi"] <= 0.99] == 0.0)               # old Tw (zero state), not a sentinel there


@pytest.mark.parametrize("kind,nlat", [("sin", 180), ("identity", 1000), ("sin", 4096)])
def test_t0_without_diffusion_is_the_pointwise_closed_form(pkg, kind, nlat, cells):
    """With D = 0 the ice-surface balance of src/miz.jl:33-43 decouples:
    k (Tm - T0)/h' + ai S - A - B (T0 - Tm) + f = 0, h' = (h == 0 ? hmin : h)  =>  T0 - Tm = (ai S - A + f)/(k/h' + B),
    whatever the active set.  The HIP path's T0 (partition + cyclic reduction on a diagonal system, through the
    active-set iteration) against that closed form — no oracle involved."""
    if cells == 2 and nlat > 1536:
        pytest.skip("two cells per thread exist up to 1536-cell meridians")
    st = pkg.SpaceTime(kind, nlat, 2000, 1)
    par = dict(pkg.default_parameters("MIZ"))
    par["D"] = 0.0
    rng = np.random.default_rng(nlat)
    ice = rng.random((2, nlat)) < 0.7
    h = np.where(ice, rng.uniform(0.1, 4.0, (2, nlat)), 0.0)
    phi = np.where(ice, rng.uniform(0.05, 1.0, (2, nlat)), 0.0)
    state = {"h": h, "phi": phi, "Ei": -par["Lf"] * h * phi, "D": np.where(ice, 50.0, 0.0),
             "Ew": np.where(phi < 1.0, rng.uniform(0.0, 5.0, (2, nlat)), 0.0), "T0": np.zeros((2, nlat))}
    fcol = np.array([-3.0, 40.0])                                    # column 1: strong forcing, surface at the melting point in places
    with make_engine(pkg, "MIZ", st, par, 2) as eng:
        eng.set_state(state)
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(700, 1, None, True)                                  # t = 0.35: sun up in the north
        T0 = eng.get_field("T0")
        Ti = eng.get_field("Ti")
        cnt = eng.counters()
    x, ct = st.x, pkg.cos2pit(float(st.t[700]))
    S = par["S0"] - par["S1"] * x * ct - par["S2"] * x * x
    hp = np.where(h == 0.0, par["hmin"], h)
    want = par["Tm"] + (par["ai"] * S - par["A"] + fcol[:, None]) / (par["k"] / hp + par["B"])
    err = float(np.max(np.abs(T0 - want) / np.maximum(1.0, np.abs(want))))
    record_error(f"T0 closed form without diffusion, {kind} {nlat}", "T0", err, 1e-13)
    assert err <= 1e-13, err
    assert (want > par["Tm"]).any() and (want < par["Tm"]).any() and cnt["cap_hits"] == 0
    np.testing.assert_array_equal(Ti[ice], np.minimum(T0, par["Tm"])[ice])    # ice_temp, zeroref!: src/miz.jl:31,65-66


@pytest.mark.parametrize("nlat", [256, 1024])
def test_classic_step_follows_the_analytic_mode_recurrence(pkg, nlat, cells):
    """The classic model on open water without insolation is linear (tests/test_analytic_solutions.py): the P_n
    amplitudes of E/cw and Tg follow e' = e + dt ((cg/tau) g - M e)/cw, g' = (g + (dt/tau) e')/(1 + dt/tau + n(n+1) dt D/cg).
    200 steps on the HIP path (forward Euler for E, implicit ghost-layer solve by partition + cyclic reduction)
    against that recurrence: error x nlat^2 = 0.167, get_diffop's second-order error — no oracle involved."""
    from test_analytic_solutions import classic_setup
    st, par, exact = classic_setup(pkg, nlat, 2000)
    amp = np.array([1.0, 3.0])
    T, G = exact(0, amp)
    with make_engine(pkg, "Classic", st, par, 2) as eng:
        eng.set_state(dict(E=par["cw"] * T, Tg=G))
        eng.set_time_table(st.t)
        eng.run(0, 200)
        got = eng.get_state(("E", "Tg", "T", "h"))
    assert (got["E"] > 0).all() and not got["h"].any()
    Tn, Gn = exact(200, amp)
    err = max(np.max(np.abs(got["E"] / par["cw"] - Tn) / amp[:, None]), np.max(np.abs(got["Tg"] - Gn) / amp[:, None])) * nlat**2
    record_error(f"classic analytic mode recurrence, {nlat} cells, 200 steps: error x nlat^2", "E/cw, Tg", float(err), 0.18)
    assert 0.15 < err < 0.18, err


@pytest.mark.parametrize("nlat", [180, 1000, 4096])
def test_compact_ice_growth_and_welding_follow_the_scalar_recurrence(pkg, nlat, cells):
    """Full ice cover, D = 0, no insolation (tests/test_analytic_solutions.py): every cell is its own scalar recurrence —
    T0 = (-A + f)/(k/h + B), h' = h - dt Fvi/Lf with Fvi = -A - B T0 + Fb + f, Ei' = -Lf h', phi' = 1,
    D' = min(Dmax, D + dt (kappa alpha/4) D^3): Stefan growth and floe welding, 100 steps on the HIP path against a
    ten-line NumPy loop of those formulas; two columns with different forcing.  No oracle involved."""
    if cells == 2 and nlat > 1536:
        pytest.skip("two cells per thread exist up to 1536-cell meridians")
    from test_analytic_solutions import compact_ice_setup, compact_ice_recurrence
    st, par, state, h0, D0 = compact_ice_setup(pkg, nlat, 2)
    fcol = np.array([0.0, -25.0])
    with make_engine(pkg, "MIZ", st, par, 2) as eng:
        eng.set_state(state)
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(0, 100, None, True)
        got = eng.get_state(ALL)
    worst = 0.0
    for c in range(2):
        h, D, T0, Dold = compact_ice_recurrence(par, st.dt, h0[c], D0[c], fcol[c], 100)
        for name, want in (("h", h), ("D", D), ("Ei", -par["Lf"] * h), ("phi", np.ones(nlat)), ("Ti", T0), ("T0", T0),
                           ("T", T0), ("E", -par["Lf"] * h), ("n", 1.0 / (par["alpha"] * Dold**2))):     # n: src/miz.jl:83-87, from the floe size before the step
            worst = max(worst, float(np.max(np.abs(got[name][c] / want - 1.0))))
        assert (got["Ew"][c] == 0.0).all()
    record_error(f"compact-ice scalar recurrence, {nlat} cells, 100 steps", "h, D, Ei, phi, Ti, T0, T, E, n", worst, 1e-12)
    assert worst < 1e-12, worst


def test_custom_grid_function(pkg, coracle, cells):
    """SpaceTime{F} with a user's own F (src/infrastructure.jl:109-141 takes any function of the uniform coordinate):
    x = (u + u^2)/2 on u in (0, 1), 250 cells.  Any F but identity goes through the non-uniform stencil
    (src/infrastructure.jl:505-526) with the x it produced: 30 steps against the oracle from a spun-up state, and the
    analytic Legendre decay (tests/test_analytic_solutions.py) on this grid too."""
    F = lambda u: (u + u * u) / 2.0
    nlat, nt = 250, 40000
    st = pkg.SpaceTime(F, nlat, nt, 1, urange=(0.0, 1.0))
    assert st.grid_kind == "nonuniform" and 0 < st.x[0] < st.x[-1] < 1
    par = pkg.default_parameters("MIZ")
    fcol = np.array([0.0, 1.5])
    ct = ctab(pkg, st)
    state = {k: np.zeros((2, nlat)) for k in PROG + ("T0",)}
    coracle.miz_run(1, st.x, dict(par), st.dt, ct[:60], np.zeros(60), fcol, state)
    with make_engine(pkg, "MIZ", st, par, 2) as eng:
        eng.set_state(state)
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(60, 30)
        got = eng.get_state(ALL)
    diag, _ = coracle.miz_run(1, st.x, dict(par), st.dt, ct[60:90], np.zeros(30), fcol, state)
    check_all(got, dict(state, **diag), 3e-12, what="custom grid x=(u+u^2)/2, 250 cells, 30 steps")   # measured 2.9e-13
    assert (got["phi"] > 0).any() and (got["phi"] == 0).any()
    # the analytic solution on the same grid
    p2 = dict(par, S0=0.0, S1=0.0, S2=0.0, A=0.0, Fb=0.0)
    x = st.x
    P2, P4 = (3 * x**2 - 1) / 2, (35 * x**4 - 30 * x**2 + 3) / 8
    lam, beta = st.dt * p2["D"] / p2["cw"], st.dt * p2["B"] / p2["cw"]
    g = [1 - m * lam - beta for m in (0, 6, 20)]
    n = 4000
    with make_engine(pkg, "MIZ", st, p2, 1) as eng:
        eng.set_field("Ew", (p2["cw"] * (10.0 + P2 + 0.5 * P4))[None])
        eng.set_time_table(st.t)
        eng.run(0, n)
        T = eng.get_field("Ew")[0] / p2["cw"]
    err = float(np.max(np.abs(T - (10.0 * g[0]**n + P2 * g[1]**n + 0.5 * P4 * g[2]**n)))) * nlat**2
    record_error("analytic Legendre decay on the custom grid, 250 cells, 4000 steps: error x nlat^2", "Ew/cw", err, 0.5)
    assert err < 0.5, err


@pytest.mark.parametrize("seed", range(10))
def test_integrate_randomized_surface(pkg, oracle, seed):
    """Seeded fuzz of integrate / savesol! (src/infrastructure.jl:549-591, 615-636): random model, grid, length, steps per
    year, duration, lastonly, winter / summer indices — including a season ON the last step of the year (then no annual
    mean is taken: the `elif` chain of savesol!) and winter == summer — and a ramping or constant Forcing; every entry of
    raw, seasonal.winter / summer / avg and ts against the oracle's integrate."""
    rng = np.random.default_rng(4242 + seed)
    model = "Classic" if seed % 4 == 3 else "MIZ"
    kind = "identity" if (model == "Classic" or seed % 2) else "sin"
    # short runs on short meridians: savesol!'s bookkeeping does not depend on the size, and over thousands of steps
    # the model itself amplifies rounding differences into ice-edge flips ("Sensitivity")
    nlat = int(rng.choice([2, 5, 9, 16]))
    nt = int(rng.choice([40, 64, 100]))                              # stable: nt >= 0.13 nlat^2
    dur = int(rng.integers(1, 4))
    lastonly = bool(rng.integers(0, 2))
    w_inx = int(rng.integers(1, nt + 1))
    s_inx = nt if seed == 1 else (w_inx if seed == 2 else int(rng.integers(1, nt + 1)))
    kw = dict(winter=(w_inx - 0.25) / nt, summer=(s_inx - 0.25) / nt)
    st, ost = pkg.SpaceTime(kind, nlat, nt, dur, **kw), oracle.SpaceTime(kind, nlat, nt, dur, **kw)
    assert (st.winter.inx, st.summer.inx) == (w_inx, s_inx) == (ost.winter_inx, ost.summer_inx)
    fargs = (0.5,) if seed % 3 == 0 else (0.0, 2.0, 0.0, (1, 0), (2.0, -2.0))
    forcing, oforcing = pkg.Forcing(*fargs), oracle.Forcing(*fargs)
    par = pkg.default_parameters(model)
    if model == "MIZ":
        init = {k: np.zeros(nlat) for k in PROG}
        names = oracle.MIZ_SOLVARS
    else:
        Ts = 30.0 - 45.0 * st.x ** 2
        init = dict(E=np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5), Tg=Ts.copy())
        names = oracle.CLASSIC_SOLVARS
    # Both forms of the run on the GPU, and the oracle's.  Values are compared with the oracle over the first 30 steps
    # only: at 40 ... 100 steps per year both models amplify rounding into regime flips within a few hundred steps
    # (the fp64 oracle is 3e-6 from its own 80-bit build after 192 classic steps of this kind) — what is fuzzed here
    # is savesol!'s bookkeeping, which is then checked EXACTLY, by identities between the outputs.
    full = pkg.integrate(model, st, forcing, par, pkg.Collection(init), lastonly=False)
    last = pkg.integrate(model, st, forcing, par, pkg.Collection(init), lastonly=True)
    with np.errstate(all="ignore"):
        ref = oracle.integrate(model, ost, oforcing, dict(par), {k: v.copy() for k, v in init.items()}, lastonly=lastonly)
    sols = last if lastonly else full
    assert np.array_equal(sols.ts, ref.ts) and len(full.ts) == nt * dur and len(last.ts) == nt
    worst = 0.0
    for v in names:
        raw = full.raw[v]
        assert raw.shape == (nt * dur, nlat) and last.raw[v].shape == (nt, nlat)
        worst = max(worst, scaled_err(raw[:30], np.stack([np.asarray(r) for r in
                    oracle.integrate(model, ost, oforcing, dict(par), {k: w.copy() for k, w in init.items()}, lastonly=False).raw[v][:30]])))
        assert np.array_equal(last.raw[v], raw[-nt:], equal_nan=True), v          # lastonly: the last year's snapshots
        for y in range(dur):
            yr = raw[y * nt:(y + 1) * nt]
            for got_l, got_f, want, inx in ((last.seasonal.winter[v], full.seasonal.winter[v], ref.winter[v], w_inx),
                                            (last.seasonal.summer[v], full.seasonal.summer[v], ref.summer[v], s_inx)):
                assert np.array_equal(got_l[y], got_f[y], equal_nan=True), (v, y)
                if want[y] is None:                                  # never written by savesol! (winter == summer: the elif)
                    assert np.isnan(got_f[y]).all() or not got_f[y].any(), (v, y)
                else:
                    assert np.array_equal(got_f[y], yr[inx - 1], equal_nan=True), (v, y)
            assert np.array_equal(last.seasonal.avg[v][y], full.seasonal.avg[v][y], equal_nan=True), (v, y)
            if ref.avg[v][y] is None:                                # a season ON the last step of the year: no mean
                assert np.isnan(full.seasonal.avg[v][y]).all() or not full.seasonal.avg[v][y].any(), (v, y)
            else:
                acc = np.zeros(nlat)
                with np.errstate(all="ignore"):
                    for row in yr:
                        acc = acc + row
                    assert np.array_equal(full.seasonal.avg[v][y], acc / nt, equal_nan=True), (v, y)
    record_error(f"integrate fuzz seed {seed}: {model} {kind} {nlat} nt={nt} dur={dur} lastonly={lastonly} w={w_inx} s={s_inx}, first 30 steps",
                 "raw", worst, 1e-9)
    # measured: <= 2.2e-13 for the MIZ seeds; 1.0e-10 for seed 3 (classic at 64 steps per year: at step 21 a cell changes
    # regime and every fp64 path jumps by x20 together — there the fp64 oracle is 3.6e-11 from its own 80-bit build and
    # the GPU 6.4e-11, tests/tools/classic_growth.py)
    assert worst <= 1e-9, worst


def test_melt_through_hands_the_surplus_to_the_water(pkg, cells):
    """Thin compact ice under 900 W/m2 of forcing melts within one step (tests/test_analytic_solutions.py): the ice
    enthalpy is clamped at zero and the surplus goes to the water (redistributeE, src/miz.jl:109-117) — afterwards
    Ei = h = D = phi = 0 and Ew = -Lf h0 + dt (-A + Fb + f): the cell's energy, conserved.  No oracle involved."""
    from test_analytic_solutions import melt_through_setup
    st, par, state, f, want_Ew = melt_through_setup(pkg, 200, 2)
    with make_engine(pkg, "MIZ", st, par, 2) as eng:
        eng.set_state(state)
        eng.set_time_table(st.t)
        eng.run(0, 1, np.full(1, f), True)
        got = eng.get_state(ALL)
    for k in ("Ei", "h", "D", "phi", "n"):
        assert not got[k].any() or k == "n", k
    err = float(np.max(np.abs(got["Ew"] / want_Ew - 1)))
    record_error("melt-through: Ew = -Lf h0 + dt (-A + Fb + f)", "Ew", err, 1e-13)
    assert err < 1e-13 and np.isnan(got["Ti"]).all()                 # no ice left: the sentinel of src/miz.jl:193


@pytest.mark.parametrize("nlat", [96, 1024])
def test_classic_ice_follows_the_cellwise_recurrence(pkg, nlat, cells):
    """The classic model with ice and without diffusion or insolation, cell by cell from WE15 eqs (A1)-(A3), (9)
    (tests/test_analytic_solutions.py): thick ice, thin ice and open water that freezes on the way, 150 steps on the HIP
    path against a fifteen-line NumPy loop of those equations.  No oracle involved."""
    from test_analytic_solutions import classic_ice_setup, classic_ice_recurrence
    st, par, E0, Tg0 = classic_ice_setup(pkg, nlat, 3)
    n, f = 150, 40.0
    with make_engine(pkg, "Classic", st, par, 3) as eng:
        eng.set_state(dict(E=E0, Tg=Tg0))
        eng.set_time_table(st.t)
        eng.run(0, n, np.full(n, f), True)
        got = eng.get_state(("E", "Tg", "T", "h"))
    E, Tg, T, h = classic_ice_recurrence(par, st.dt, E0, Tg0, f, n)
    assert ((E0 > 0) & (E < 0)).any() and (E0 < 0).any()
    worst = max(float(np.max(np.abs(got[k] - w) / np.maximum(1.0, np.abs(w)))) for k, w in (("E", E), ("Tg", Tg), ("T", T), ("h", h)))
    record_error(f"classic ice cell-wise recurrence, {nlat} cells, 150 steps", "E, Tg, T, h", worst, 1e-12)
    assert worst < 1e-12, worst


@pytest.mark.parametrize("kind,nlat,limit", [("identity", 256, 0.52), ("sin", 1024, 2.8), ("sin", 4096, 3.2)])
def test_t0_system_couples_ice_and_water_through_the_diffusion(pkg, kind, nlat, limit, cells):
    """The T0 system with its diffusion terms (tests/test_analytic_solutions.py): half ice cover, uniform thickness, water
    temperature c + a P_2(x): T0 - Tm = (-A + f)/(k/h + B) - 6 D (1 - phi) a/(k/h + B + 6 D phi) P_2(x) up to the stencil's
    second-order error (x nlat^2: 0.509 identity, 2.76 sin; 3.06 at 4096 cells where rounding shows) — assembly of the rows,
    concentration weighting of the active set, the water's diffusion on the right-hand side and the solve, against
    mathematics.  Three columns with different forcing.  No oracle involved."""
    if cells == 2 and nlat > 1536:
        pytest.skip("two cells per thread exist up to 1536-cell meridians")
    from test_analytic_solutions import t0_mode_setup
    st, par, state, exact = t0_mode_setup(pkg, kind, nlat, 3)
    fcol = np.array([0.0, -30.0, 25.0])
    with make_engine(pkg, "MIZ", st, par, 3) as eng:
        eng.set_state(state)
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(0, 1, None, True)
        T0 = eng.get_field("T0")
        cnt = eng.counters()
    assert (T0 < par["Tm"]).all() and cnt["cap_hits"] == 0
    err = max(float(np.max(np.abs(T0[c] - exact(fcol[c])))) for c in range(3)) * nlat**2
    record_error(f"T0 system with diffusion, Legendre right-hand side, {kind} {nlat}: error x nlat^2", "T0", err, limit)
    assert 0.9 * limit / 1.1 < err < limit or nlat == 4096 and err < limit, err


def test_lateral_melt_step_matches_its_closed_form(pkg, cells):
    """Partial cover over warm water, one step, no diffusion (tests/test_analytic_solutions.py): lateral melt rate, the
    energy it moves from ice to water, thickness, floe size (lateral melt + welding), concentration and the diagnostics
    n, E, T, Ti, Tw — every output of the step against a dozen lines of NumPy written from src/miz.jl:71-146.  No oracle."""
    from test_analytic_solutions import lateral_melt_setup, lateral_melt_step
    st, par, state = lateral_melt_setup(pkg, 300, 2)
    fcol = np.array([-20.0, 10.0])
    with make_engine(pkg, "MIZ", st, par, 2) as eng:
        eng.set_state(state)
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(0, 1, None, True)
        got = eng.get_state(ALL)
    worst = 0.0
    for c in range(2):
        want = lateral_melt_step(par, st.dt, {k: v[c] for k, v in state.items()}, fcol[c])
        for k, w in want.items():
            worst = max(worst, float(np.max(np.abs(got[k][c] / w - 1))))
    record_error("lateral-melt step closed form, 300 cells", "all eleven outputs", worst, 1e-12)
    assert worst < 1e-12, worst


def test_device_schedule_follows_the_docstring_forcing(pkg):
    """`Forcing(0.0, 5.0, -5.0, (10, 10), (0.5, -0.5))` as the reference's docstring prints it (src/infrastructure.jl:193-205):
    0 on [0,10), 0.5 (t - 10) on [10,20), 5 on [20,30), 5 - 0.5 (t - 30) on [30,50), -5 afterwards — evaluated ON THE DEVICE
    per column at the model time of each step (ebm_set_column_schedule).  With D = 0 and S = 0 the forcing a step saw can
    be read back from its T0: f = (T0 - Tm)(k/h + B) + A.  Probed in all five segments of a 55-year run, next to a
    constant-forcing column."""
    nt, nlat = 20, 8
    st = pkg.SpaceTime("identity", nlat, nt, 55)
    par = dict(pkg.default_parameters("MIZ"))
    par.update(S0=0.0, S1=0.0, S2=0.0, D=0.0, kappa=0.0)
    h0 = np.full((2, nlat), 2.0)
    state = {"h": h0, "phi": np.ones((2, nlat)), "D": np.full((2, nlat), 50.0), "Ei": -par["Lf"] * h0, "Ew": np.zeros((2, nlat)),
             "T0": np.zeros((2, nlat))}
    doc = lambda t: 0.0 if t < 10 else 0.5 * (t - 10) if t < 20 else 5.0 if t < 30 else 5.0 - 0.5 * (t - 30) if t < 50 else -5.0
    with make_engine(pkg, "MIZ", st, par, 2) as eng:
        eng.set_state(state)
        eng.set_time_table(st.t)
        eng.set_column_schedules([pkg.Forcing(0.0, 5.0, -5.0, (10, 10), (0.5, -0.5)), pkg.Forcing(1.25)])
        done, worst = 0, 0.0
        for t_probe in (3.3, 9.98, 10.02, 17.57, 19.99, 20.0, 25.0, 30.01, 41.3, 49.99, 50.0, 54.9):
            step = int(t_probe * nt)                                 # the step whose model time is st.T[step]
            eng.run(done, step - done, None, False)
            h = eng.get_field("h")
            eng.run(step, 1, None, True)
            done = step + 1
            f_seen = (eng.get_field("T0") - par["Tm"]) * (par["k"] / h + par["B"]) + par["A"]
            worst = max(worst, float(np.max(np.abs(f_seen[0] - doc(float(st.T[step]))))), float(np.max(np.abs(f_seen[1] - 1.25))))
    record_error("device-side Forcing schedule read back through T0 against the docstring's formula", "f", worst, 1e-11)
    assert worst < 1e-11, worst
    assert abs(pkg.Forcing(0.0, 5.0, -5.0, (10, 10), (0.5, -0.5))(17.57) - 3.785) < 1e-12       # the docstring's own sample


def test_classic_ice_with_insolation_follows_the_cellwise_recurrence(pkg, cells):
    """The classic cell-wise recurrence with the sun on (tests/test_analytic_solutions.py): albedo by the sign of E, S at this
    time level in the surface balance and at the NEXT one in the ghost layer (src/classic.jl:47-60); 700 steps from
    mid-winter, cells freezing and melting on the way; run in two calls across the year's wrap of the time table."""
    from test_analytic_solutions import classic_ice_setup, classic_ice_recurrence
    st, par, E0, Tg0 = classic_ice_setup(pkg, 200, 2)
    par.update({k: pkg.default_parameters("Classic")[k] for k in ("S0", "S1", "S2")})
    n, first = 700, 1600                                             # steps 1600 ... 2299: wraps at 2000
    with make_engine(pkg, "Classic", st, par, 2) as eng:
        eng.set_state(dict(E=E0, Tg=Tg0))
        eng.set_time_table(st.t)
        eng.run(first, 450, None, False)
        eng.run(first + 450, n - 450, None, True)
        got = eng.get_state(("E", "Tg", "T", "h"))
    E, Tg, T, h = classic_ice_recurrence(par, st.dt, E0, Tg0, 0.0, n, st.x, np.roll(st.t, -first))
    assert ((E0 > 0) & (E < 0)).any() or ((E0 < 0) & (E > 0)).any()
    worst = max(float(np.max(np.abs(got[k] - w) / np.maximum(1.0, np.abs(w)))) for k, w in (("E", E), ("Tg", Tg), ("T", T), ("h", h)))
    record_error("classic ice cell-wise recurrence with insolation, 200 cells, 700 steps", "E, Tg, T, h", worst, 1e-11)
    assert worst < 1e-11, worst


@pytest.mark.parametrize("nlat", [2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 127, 129])
def test_tiny_and_odd_sizes_through_every_entry_point(pkg, coracle, nlat):
    """Meridians of 2 ... 129 cells — one thread, one ragged chunk, exactly one wave, one wave and one cell — through ebm_run,
    ebm_run_fused, ebm_integrate (sums + snapshots), the hemispheric mean and the extension, for the MIZ model on both grids
    and the classic model: everything against the oracle after 25 steps from a state with ice and open water."""
    nt, nsteps, ncol = 4000, 25, 3
    fcol = np.array([-1.0, 0.0, 2.5])
    for model, kind in (("MIZ", "sin"), ("MIZ", "identity"), ("MIZ_IMEX", "sin"), ("Classic", "identity")):
        st = pkg.SpaceTime(kind, nlat, nt, 1)
        par = pkg.default_parameters("Classic" if model == "Classic" else "MIZ")
        ct = ctab(pkg, st)
        if model == "Classic":
            Ts = 30.0 - 45.0 * st.x ** 2
            state = dict(E=np.tile(np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5), (ncol, 1)), Tg=np.tile(Ts, (ncol, 1)))
            idx = np.arange(nsteps)
            ref = {k: v.copy() for k, v in state.items()}
            ref.update(coracle.classic_run(st.x, dict(par), st.dt, ct[idx], ct[(idx + 1) % nt], np.zeros(nsteps), fcol, ref))
            names = ("E", "Tg", "T", "h")
        else:
            state = {k: np.zeros((ncol, nlat)) for k in PROG + ("T0",)}
            coracle.miz_run(0 if kind == "identity" else 1, st.x, dict(par), st.dt, ct[:40], np.zeros(40), fcol, state, imex=(model == "MIZ_IMEX"))
            ref = {k: v.copy() for k, v in state.items()}
            diag, _ = coracle.miz_run(0 if kind == "identity" else 1, st.x, dict(par), st.dt, ct[40:40 + nsteps], np.zeros(nsteps), fcol, ref,
                                      imex=(model == "MIZ_IMEX"))
            ref.update(diag)
            names = ALL
        first = 0 if model == "Classic" else 40
        got = {}
        # (`integrate` needs a time table of exactly its nt entries: a window of the year's.  The classic step also reads the
        #  NEXT entry, which at the window's last step wraps inside the window — so integrate is compared with a run over
        #  the same window, and the oracle with the runs over the year's table)
        for how in ("run", "fused", "integrate", "run_window"):
            with pkg.Engine(model, st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol, device=0) as eng:
                eng.set_state(state)
                eng.set_column_forcing(fcol)
                if how == "integrate":
                    eng.set_time_table(np.take(st.t, np.arange(first, first + nsteps)))
                    out = eng.integrate(nsteps, 1, None, True, 7, 19, names[:4])
                    got[how] = eng.get_state(names)
                    for vi, v in enumerate(names[:4]):               # raw[-1] is the final state, the snapshots are raw's rows
                        assert np.array_equal(out["raw"][vi, -1], got[how][v], equal_nan=True), (model, v)
                        assert np.array_equal(out["winter"][vi, 0], out["raw"][vi, 6], equal_nan=True)
                        acc = np.zeros((ncol, nlat))
                        with np.errstate(all="ignore"):
                            for row in out["raw"][vi]:
                                acc = acc + row
                            assert np.array_equal(out["avg"][vi, 0], acc / nsteps, equal_nan=True), (model, v)
                elif how == "run_window":
                    eng.set_time_table(np.take(st.t, np.arange(first, first + nsteps)))
                    eng.run(0, nsteps, None, True)
                    got[how] = eng.get_state(names)
                else:
                    eng.set_time_table(st.t)
                    eng.run(first, nsteps, None, True, steps_per_launch=(4 if how == "fused" else 1))
                    got[how] = eng.get_state(names)
                    hm = eng.hemispheric_mean("T")
                    assert np.array_equal(hm, pkg.hemispheric_mean(got[how]["T"], st.x), equal_nan=True)
        for k in names:
            assert np.array_equal(got["run"][k], got["fused"][k], equal_nan=True), (model, kind, k)
            assert np.array_equal(got["run_window"][k], got["integrate"][k], equal_nan=True), (model, kind, k)
            if model != "Classic":
                assert np.array_equal(got["run"][k], got["run_window"][k], equal_nan=True), (model, kind, k)
        worst = max(scaled_err(got["run"][k], ref[k]) for k in names)
        # Bar: 1e-10 — unless the model itself amplifies rounding beyond that over these 25 steps, which the oracle's own
        # 80-bit build measures (same source, same fp64 inputs, ~2000x less rounding in between: how far the fp64 ORACLE is
        # from the exactly evaluated discrete model).  On the identity grid at 127 / 129 cells the freeze-up front does:
        # both fp64 paths sit ~1e-10 from the 80-bit result, GPU-vs-oracle 8e-11 (two cells per thread) / 1.2e-10 (four).
        bar = 1e-10
        if model != "Classic":
            import __graft_entry__ as graft
            ld = graft.load_oracle()[1].COracle(extended=True)
            ext = {k: v.copy() for k, v in state.items()}
            dl, _ = ld.miz_run(0 if kind == "identity" else 1, st.x, dict(par), st.dt, ct[40:40 + nsteps], np.zeros(nsteps), fcol, ext,
                               imex=(model == "MIZ_IMEX"))
            ext.update(dl)
            amp = max(scaled_err(ref[k], ext[k]) for k in names)
            record_error(f"tiny sizes: {model} {kind} {nlat} cells, 25 steps: oracle vs 80-bit", "all fields", amp, float("nan"))
            bar = max(bar, 3.0 * amp)
        record_error(f"tiny sizes: {model} {kind} {nlat} cells, 25 steps", "all fields", worst, bar)
        assert worst <= bar, (model, kind, worst, bar)


@pytest.mark.parametrize("seed", range(8))
def test_random_call_sequences_track_the_oracle(pkg, coracle, seed):
    """Stateful fuzz of the handle: random sequences of the calls a driver makes — runs of 1 ... 200 steps with one launch
    per step (long ones replay the captured graph), fused runs, per-step forcing series, new column offsets in between
    (the graph must be rebuilt), per-column schedules switched on and off, prognostic fields overwritten from the host, the
    T0 warm start overwritten with garbage (the solution must not care), single `ebm_step` calls, the step clock moved —
    mirrored on the oracle operation by operation, 600 steps in all, every field compared after every run.  What is fuzzed
    is the handle's bookkeeping (clock, time-table wrap, tables behind a captured graph, forcing plumbing), so the model runs
    in a benign regime — open water without insolation, where it is linear (tests/test_analytic_solutions.py) — and the bar
    can stay at 1e-10 (measured: <= 1.7e-11, mostly T0 of the decoupled open-water rows): from the zero state the same sequences make the MODEL amplify rounding until the fp64 oracle is 1e-5
    from its own 80-bit build, and no bar means anything."""
    rng = np.random.default_rng(900 + seed)
    kind = "identity" if seed % 2 else "sin"
    nlat, ncol, nt = int(rng.choice([64, 180, 333])), 3, 16000
    st = pkg.SpaceTime(kind, nlat, nt, 3)
    par = dict(pkg.default_parameters("MIZ"))
    par.update(S0=0.0, S1=0.0, S2=0.0, A=0.0, Fb=0.0)
    kid = 0 if kind == "identity" else 1
    ct = ctab(pkg, st)
    x = st.x
    state = {k: np.zeros((ncol, nlat)) for k in PROG + ("T0",)}
    state["Ew"] = par["cw"] * np.outer(1.0 + 0.1 * np.arange(ncol), 10.0 + (3 * x**2 - 1) / 2 + 0.5 * (35 * x**4 - 30 * x**2 + 3) / 8)
    fcol = np.zeros(ncol)
    scheds = None
    clock, total, worst, compared = int(rng.integers(0, 2 * nt)), 0, 0.0, 0

    def oracle_advance(first, n, f_steps):
        idx = (first + np.arange(n)) % nt
        base = np.zeros(n) if f_steps is None else f_steps
        if scheds is None:
            return coracle.miz_run(kid, st.x, dict(par), st.dt, ct[idx], base, fcol, state)[0]
        diag = {}
        for c in range(ncol):
            extra = np.array([scheds[c](float(st.T[(first + i) % len(st.T)])) for i in range(n)])
            sub = {k: v[c:c + 1].copy() for k, v in state.items()}
            d, _ = coracle.miz_run(kid, st.x, dict(par), st.dt, ct[idx], base + extra, fcol[c:c + 1], sub)
            for k, v in sub.items():
                state[k][c] = v[0]
            for k, v in d.items():
                diag.setdefault(k, np.zeros((ncol, nlat)))[c] = v[0]
        return diag

    with make_engine(pkg, "MIZ", st, par, ncol) as eng:
        eng.set_state(state)
        eng.set_time_table(st.t)
        eng.set_step_clock(clock)
        while total < 600:
            op = rng.choice(["run", "run", "long_run", "fused", "fcol", "sched", "poke", "t0", "step", "jump"])
            if op in ("run", "long_run", "fused"):
                n = int(rng.integers(130, 200)) if op == "long_run" else int(rng.integers(1, 40))
                n = min(n, 600 - total)
                f_steps = None if rng.random() < 0.5 else rng.normal(0.0, 1.0, n)
                eng.run(clock, n, f_steps, True, steps_per_launch=(int(rng.integers(2, 9)) if op == "fused" else 1))
                diag = oracle_advance(clock, n, f_steps)
                clock, total = clock + n, total + n
                got = eng.get_state(ALL)
                e = max(scaled_err(got[k], dict(state, **diag)[k]) for k in ALL)
                worst, compared = max(worst, e), compared + 1
                assert e <= 1e-10, (op, n, total, e)
            elif op == "fcol":
                fcol = rng.uniform(-3.0, 3.0, ncol)
                eng.set_column_forcing(fcol)
            elif op == "sched":
                if (st.T[-1] - clock / nt) < 1.5:                    # schedules are defined over the run's duration
                    continue
                scheds = None if scheds is not None else [pkg.Forcing(0.0, 2.0 * (c + 1), -1.0, (0, 1), (2.0 * (c + 1), -1.0 - 2.0 * (c + 1)))
                                                         for c in range(ncol)]
                eng.set_column_schedules(scheds)
            elif op == "poke":
                state["Ew"] = state["Ew"] * float(rng.uniform(0.9, 1.1))
                eng.set_field("Ew", state["Ew"])
            elif op == "t0":
                eng.set_field("T0", rng.normal(0.0, 5.0, (ncol, nlat)))        # garbage warm start: only its signs are used
            elif op == "jump":
                clock = int(rng.integers(0, 2 * nt))                 # a driver restarting elsewhere in the schedule
                eng.set_step_clock(clock)
            else:
                f = float(rng.normal())
                eng.step(ct[clock % nt], ct[(clock + 1) % nt], f, True)
                oracle_advance(clock, 1, np.array([f]))
                clock, total = clock + 1, total + 1
                eng.set_step_clock(clock)
        cnt = eng.counters()
    record_error(f"random call sequence, seed {seed}: {kind} {nlat}, 600 steps, {compared} comparisons", "all fields", worst, 1e-10)
    assert compared >= 5 and cnt["cap_hits"] == 0, compared


@pytest.mark.parametrize("T", list(range(64, 1025, 64)))
def test_every_workgroup_size(pkg, coracle, T, monkeypatch):
    """Every workgroup size is its own set of kernel instantiations (424 of the library's 443): for each T = 64 ... 1024, both grid
    kinds, four cells per thread (nlat = 4T - 1, ragged) and two where that geometry exists (nlat = 2T - 1; T <= 512 and 768),
    the state-only, diagnostic and savesol! kernels, the fused-K kernel of the shape (state in registers, or resident in LDS
    beyond 512 threads and for the extension) and its savesol! variant, the extension — 12 steps from a state with ice and
    open water against the oracle, and the identities between the paths bitwise."""
    nsteps, ncol = 12, 2
    fcol = np.array([-1.0, 1.5])
    for cells in (4, 2):
        if cells == 2 and not (T <= 512 or T == 768):
            continue
        monkeypatch.setenv("EBM_CELLS_PER_THREAD", str(cells))
        nlat = cells * T - 1
        dt = 1.0 / max(2000.0, 0.7 * nlat * nlat)                   # stable; only the first 64 steps of such a year are used,
        t64 = (np.arange(64) + 0.5) * dt                             # so the (long) time axis itself is never built
        for kind in ("sin", "identity"):
            st = pkg.SpaceTime(kind, nlat, 2000, 1)                  # for x only
            par = pkg.default_parameters("MIZ")
            kid = 0 if kind == "identity" else 1
            ct = np.array([pkg.cos2pit(float(t)) for t in t64])
            for model in ("MIZ", "MIZ_IMEX") if cells == 4 else ("MIZ",):
                imex = model == "MIZ_IMEX"
                state = {k: np.zeros((ncol, nlat)) for k in PROG + ("T0",)}
                coracle.miz_run(kid, st.x, dict(par), dt, ct[:30], np.zeros(30), fcol, state, imex=imex)
                ref = {k: v.copy() for k, v in state.items()}
                diag, ocnt = coracle.miz_run(kid, st.x, dict(par), dt, ct[30:30 + nsteps], np.zeros(nsteps), fcol, ref, imex=imex)
                ref.update(diag)
                got = {}
                saved = {}
                for how in ("run", "fused", "fused_lds", "integrate", "integrate_fused"):
                    # fused-K with the state in registers where that kernel exists, and in LDS at EVERY size (same bits)
                    with pkg.Engine(model, st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), dt, ncol, device=0,
                                    fused_state_in_lds=(how == "fused_lds")) as eng:
                        info = eng.launch_info()
                        assert info["threads"] == T and info["cells_per_thread"] == cells, (info, T, cells)
                        eng.set_state(state)
                        eng.set_column_forcing(fcol)
                        if how.startswith("integrate"):
                            # with the raw snapshots every step is its own launch; without them the stretches between the
                            # seasonal steps are fused (the savesol! variant of the resident kernel; of the register kernel with two
                            # cells per thread, which does not exist at 768 threads)
                            eng.set_time_table(t64[30:30 + nsteps])
                            out = eng.integrate(nsteps, 1, None, True, 5, 9, ("E", "T", "phi", "h"), want_raw=(how == "integrate"))
                            got[how] = eng.get_state(ALL)
                            saved[how] = out
                            assert eng.counters()["launches"] == (nsteps if (how == "integrate" or (cells == 2 and T == 768)) else 6), (how, eng.counters())
                            for vi, v in enumerate(("E", "T", "phi", "h")):
                                if how == "integrate":
                                    assert np.array_equal(out["raw"][vi, -1], got[how][v], equal_nan=True), (T, cells, kind, model, v)
                                    assert np.array_equal(out["summer"][vi, 0], out["raw"][vi, 8], equal_nan=True)
                        else:
                            eng.set_time_table(t64)
                            eng.run(30, nsteps, None, True, steps_per_launch=(5 if how.startswith("fused") else 1))
                            got[how] = eng.get_state(ALL)
                            cnt = eng.counters()
                            assert cnt["launches"] == (nsteps if how == "run" else 3), (how, cnt)   # every shape has a fused-K kernel
                for k in ALL:
                    assert np.array_equal(got["run"][k], got["fused"][k], equal_nan=True), (T, cells, kind, model, k)
                    assert np.array_equal(got["run"][k], got["fused_lds"][k], equal_nan=True), (T, cells, kind, model, k)
                    assert np.array_equal(got["run"][k], got["integrate"][k], equal_nan=True), (T, cells, kind, model, k)
                    assert np.array_equal(got["run"][k], got["integrate_fused"][k], equal_nan=True), (T, cells, kind, model, k)
                for k in ("winter", "summer", "avg"):
                    assert np.array_equal(saved["integrate"][k], saved["integrate_fused"][k], equal_nan=True), (T, cells, kind, model, k)
                worst = max(scaled_err(got["run"][k], ref[k]) for k in ALL)
                record_error(f"workgroup size {T}, {cells} cells per thread, {kind}, {model}: {nlat} cells, 12 steps", "all fields", worst, 1e-11)
                assert worst <= 1e-11, (T, cells, kind, model, worst)         # measured: <= 6.5e-13 over all 82 combinations
