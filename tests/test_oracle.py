"""CPU tests of the oracle itself: it must be right before it may judge the HIP path.

The reference cannot run here and its golden file is absent ("parity unpinned"), so the oracle
is pinned by (i) the known answers printed in the reference's docstrings, (ii) closed forms
that follow from the reference text, (iii) invariants of the reference's expressions, and
(iv) agreement between its two independent restatements (NumPy and C), bit for bit."""
import math

import numpy as np
import pytest

import os

from conftest import ROOT, load_golden, scaled_err

PROG = ("Ei", "Ew", "h", "D", "phi")
DIAG = ("Tw", "Ti", "n", "E", "T")


# ---- (i) known answers from the reference's docstrings --------------------------------------
def test_grid_matches_reference_docstring(oracle):
    # src/EnergyBalanceModel.jl:18-22 / src/infrastructure.jl:101-106
    st = oracle.SpaceTime("sin", 180, 2000, 30)
    assert [f"{v:.6g}" for v in st.x[:2]] == ["0.00436331", "0.0130896"]
    assert [f"{v:.6g}" for v in st.x[-2:]] == ["0.999914", "0.99999"]
    assert list(st.t[:3]) == [0.00025, 0.00075, 0.00125]
    assert list(st.t[-2:]) == [0.99925, 0.99975]
    # src/infrastructure.jl:94-99
    st = oracle.SpaceTime("identity", 100, 2000, 30)
    assert list(st.x[:3]) == [0.005, 0.015, 0.025] and list(st.x[-3:]) == [0.975, 0.985, 0.995]
    assert (st.winter_inx, st.summer_inx) == (522, 1548)      # round(522.5) is half-to-even
    assert len(st.T) == 60000 and st.T[-1] == pytest.approx(30 - 0.00025, abs=1e-12)


def test_forcing_matches_reference_docstring(oracle):
    # src/infrastructure.jl:193-205
    f = oracle.Forcing(0.0, 5.0, -5.0, (10, 10), (0.5, -0.5))
    assert f.domain == (0, 10, 20, 30, 50)
    assert f(17.57) == pytest.approx(3.785, abs=1e-12)
    assert (f(5.0), f(25.0), f(40.0), f(60.0)) == (0.0, 5.0, 0.0, -5.0)
    assert oracle.Forcing(1.25)(123.0) == 1.25
    with pytest.raises(ValueError):
        oracle.Forcing(0.0, 5.0, -5.0, (10, 10), (0.3, -0.5))   # warming time not an integer
    with pytest.raises(ValueError):
        oracle.Forcing(0.0, 5.0, -5.0, (10, 10), (0.5, 0.5))    # cooling rate must be negative


def test_parameter_sets(oracle):
    # src/EnergyBalanceModel.jl:30 (22 entries), src/infrastructure.jl:459 (16 entries)
    assert len(oracle.default_parameters("MIZ")) == 22
    assert len(oracle.default_parameters("Classic")) == 16
    p = oracle.default_parameters("MIZ")
    assert p["m1"] == 1.6e-6 * 31536000 and f"{p['m1']:.6g}" == "50.4576"   # docstring :35
    assert p["kappa"] == 0.01 * 31536000 and p["Dmax"] == 156.0


# ---- (ii) closed forms that follow from the reference text ------------------------------------
def test_miz_step1_closed_form(oracle):
    """From the all-zero state (test/runtests.jl:24-31) step 1 has closed forms
    (SURVEY §8c): Tw = Ti = n = Flat = 0, Fvw = (a0 - a2 x^2) S - A + Fb, and cells with
    rEw < 0 freeze: Ei = rEw, Ew = 0, D = Dmin, h = hmin, phi = min(1, -Ei/(Lf hmin))."""
    st = oracle.SpaceTime("sin", 180, 2000, 1)
    par = oracle.default_parameters("MIZ")
    geom = oracle.DiffusionGeometry("sin", st.x, par["D"])
    z = {k: np.zeros(180) for k in PROG}
    ct = oracle.cos2pit(float(st.t[0]))
    out, T0, nit, ok = oracle.step_miz(ct, 0.0, z, np.zeros(180), st.x, st.dt, geom, par)
    x = st.x
    S = par["S0"] - par["S1"] * x * ct - par["S2"] * x * x
    Fvw = (par["a0"] - par["a2"] * x * x) * S - par["A"] + par["Fb"]
    rEw = st.dt * Fvw
    frz = rEw < 0
    assert frz.any() and (~frz).any()
    assert np.all(out["n"] == 0.0)
    np.testing.assert_allclose(out["Ei"][frz], rEw[frz], rtol=1e-13)
    assert np.all(out["Ew"][frz] == 0.0) and np.all(out["Ei"][~frz] == 0.0)
    np.testing.assert_allclose(out["Ew"][~frz], rEw[~frz], rtol=1e-13)
    assert np.all(out["D"][frz] == par["Dmin"]) and np.all(out["D"][~frz] == 0.0)
    np.testing.assert_allclose(out["h"][frz], par["hmin"], rtol=1e-13)
    np.testing.assert_allclose(out["phi"][frz], np.minimum(1.0, -rEw[frz] / (par["Lf"] * par["hmin"])), rtol=1e-12)
    # sentinels: Ti is NaN where there is no ice, Tw where phi > 0.99 (src/miz.jl:193-194)
    assert np.all(np.isnan(out["Ti"][~frz])) and not np.isnan(out["Ti"][frz]).any()


def test_redistributeE_conserves(oracle):
    rng = np.random.default_rng(0)
    rEi, rEw = rng.normal(size=1000), rng.normal(size=1000)
    Ei, Ew, psi_i, psi_w = oracle.redistributeE(rEi, rEw)
    np.testing.assert_allclose(Ei + Ew, rEi + rEw, rtol=0, atol=1e-15)
    assert np.all(Ei <= 0) and np.all(Ew >= 0) is not None
    assert np.all(psi_i >= 0) and np.all(psi_w <= 0)


@pytest.mark.parametrize("kind", ["sin", "identity"])
def test_diffusion_is_conservative_and_kills_constants(oracle, kind):
    """Flux form with zero-flux ends: sum_k w_k (D lap T)_k = 0 and D lap(const) = 0
    (src/infrastructure.jl:485-488, 524)."""
    st = oracle.SpaceTime(kind, 180, 2000, 1)
    g = oracle.DiffusionGeometry(kind, st.x, 0.6)
    rng = np.random.default_rng(1)
    T = rng.normal(size=180) * 10
    d = g.add(np.zeros(180), T)
    w = g.w if kind == "sin" else np.full(180, 1.0 / 180)
    assert abs(np.sum(w * d)) < 1e-9 * np.sum(np.abs(w * d))
    assert np.max(np.abs(g.add(np.zeros(180), np.full(180, 3.7)))) < 1e-9
    # tridiagonal coefficients describe the same operator
    Tm = np.concatenate(([0.0], T[:-1])); Tp = np.concatenate((T[1:], [0.0]))
    np.testing.assert_allclose(g.lo * Tm + g.di * T + g.up * Tp, d, rtol=1e-9, atol=1e-7)


def test_T0_root_meets_reference_solver_tolerance(oracle, coracle):
    """The reference accepts a T0 with residual <= abstol = 1e-8 (src/miz.jl:58-59).  The
    oracle's exact root must satisfy the reference's own residual function far below that."""
    g = load_golden("miz_sin_180_2000.npz")
    par = oracle.default_parameters("MIZ")
    x, t = g["x"], g["t"]
    for s in (10, 522, 1548):
        # state after step s; T0 of step s+1 solves T0eq built from state s
        st = {k: g[f"s{s}_{k}"] for k in PROG}
        T0 = g[f"s{s+1}_T0"]
        ct = oracle.cos2pit(float(t[s]))
        res = coracle.T0eq(1, x, par, ct, 0.0, st["h"], st["Ew"], st["phi"], T0)
        assert np.max(np.abs(res)) < 1e-9, (s, np.max(np.abs(res)))


def test_thomas_solves(oracle, coracle):
    rng = np.random.default_rng(3)
    n = 257
    a, c = -rng.random(n), -rng.random(n)
    a[0] = c[-1] = 0.0
    b = 2.5 + rng.random(n)
    x = rng.normal(size=n)
    d = b * x + a * np.concatenate(([0.0], x[:-1])) + c * np.concatenate((x[1:], [0.0]))
    np.testing.assert_allclose(oracle.thomas(a, b, c, d), x, rtol=1e-12)
    assert np.array_equal(coracle.thomas(a, b, c, d), oracle.thomas(a, b, c, d))


# ---- (iii)/(iv) the two restatements agree bit for bit; golden files are reproducible --------
@pytest.mark.parametrize("kind,kid", [("sin", 1), ("identity", 0)])
def test_c_port_equals_numpy_and_golden(oracle, coracle, kind, kid):
    g = load_golden(f"miz_{kind}_180_2000.npz")
    st = oracle.SpaceTime(kind, 180, 2000, 1)
    assert np.array_equal(st.x, g["x"])
    par = oracle.default_parameters("MIZ")
    geo = coracle.geometry(kid, st.x, par["D"])
    gn = oracle.DiffusionGeometry(kind, st.x, par["D"])
    for i, nm in enumerate(("lo", "di", "up")):
        assert np.array_equal(geo[5 + i], getattr(gn, nm))
    state = {k: np.zeros((1, 180)) for k in PROG + ("T0",)}
    ct = np.array([oracle.cos2pit(float(t)) for t in st.t])
    done = 0
    for s in (1, 2, 10, 11, 522, 523, 1548, 2000):
        diag, cnt = coracle.miz_run(kid, st.x, par, st.dt, ct[done:s], np.zeros(s - done), None, state)
        done = s
        assert cnt[1] == 0
        for k in PROG + ("T0",):
            assert np.array_equal(state[k][0], g[f"s{s}_{k}"], equal_nan=True), (s, k)
        for k in DIAG:
            assert np.array_equal(diag[k][0], g[f"s{s}_{k}"], equal_nan=True), (s, k)


def test_classic_c_port_equals_golden(oracle, coracle):
    g = load_golden("classic_identity_180_2000.npz")
    st = oracle.SpaceTime("identity", 180, 2000, 1)
    par = oracle.default_parameters("Classic")
    state = dict(E=g["s0_E"][None].copy(), Tg=g["s0_Tg"][None].copy())
    ct = np.array([oracle.cos2pit(float(t)) for t in st.t])
    done = 0
    for s in (1, 2, 10, 522, 2000):
        idx = np.arange(done, s)
        out = coracle.classic_run(st.x, par, st.dt, ct[idx], ct[(idx + 1) % 2000], np.zeros(len(idx)), None, state)
        done = s
        assert np.array_equal(state["E"][0], g[f"s{s}_E"]) and np.array_equal(state["Tg"][0], g[f"s{s}_Tg"])
        assert np.array_equal(out["T"][0], g[f"s{s}_T"]) and np.array_equal(out["h"][0], g[f"s{s}_h"])


def test_classic_physics_sanity(oracle):
    """WE15 default climate: ice-free tropics, seasonal ice at the pole, h = -E/Lf where E<0."""
    g = load_golden("classic_identity_180_2000.npz")
    E, T, h = g["s2000_E"], g["s2000_T"], g["s2000_h"]
    assert T[0] > 25 and T[-1] < 0 and np.all(np.diff(T) < 1e-9)
    assert np.array_equal(h > 0, E < 0)
    np.testing.assert_allclose(h[E < 0], -E[E < 0] / 9.5, rtol=1e-14)


def test_oracle_integrate_savesol_semantics(oracle):
    """savesol!: winter/summer snapshots are raw[522-1]/raw[1548-1]; avg is the mean over
    the year's nt snapshots and NaN wherever a sentinel occurred (src/infrastructure.jl:549-591)."""
    st = oracle.SpaceTime("identity", 60, 400, 2)
    par = oracle.default_parameters("MIZ")
    init = {k: np.zeros(60) for k in PROG}
    sols = oracle.integrate("MIZ", st, oracle.Forcing(0.0), par, init, lastonly=False)
    assert len(sols.raw["E"]) == 800 and len(sols.ts) == 800
    wi, si = st.winter_inx, st.summer_inx
    for y in range(2):
        assert np.array_equal(sols.winter["h"][y], sols.raw["h"][y * 400 + wi - 1])
        assert np.array_equal(sols.summer["phi"][y], sols.raw["phi"][y * 400 + si - 1])
        m = np.mean(np.stack(sols.raw["E"][y * 400:(y + 1) * 400]), axis=0)
        np.testing.assert_allclose(sols.avg["E"][y], m, rtol=1e-12)
    last = oracle.integrate("MIZ", st, oracle.Forcing(0.0), par, init, lastonly=True)
    assert len(last.raw["E"]) == 400
    assert np.array_equal(last.raw["T"][399], sols.raw["T"][799], equal_nan=True)
    assert last.ts[0] == pytest.approx(1.0 + st.dt / 2)


@pytest.mark.parametrize("seed", range(10))
def test_c_port_equals_numpy_on_random_states(oracle, coracle, seed):
    """The two restatements (NumPy, C) must agree bit for bit not only along the golden
    trajectories but from arbitrary states: seeded random grids, perturbed parameters, states mixing
    open water / ice / saturated cells, random warm starts and forcing — one step each."""
    rng = np.random.default_rng(500 + seed)
    nx = int(rng.choice([2, 3, 7, 33, 64, 100, 181]))
    kind, kid = ("sin", 1) if rng.random() < 0.6 else ("identity", 0)
    st = oracle.SpaceTime(kind, nx, int(max(2000, 0.7 * nx * nx)), 1)
    par = dict(oracle.default_parameters("MIZ"))
    for k in ("D", "A", "B", "S1", "a0", "ai", "Fb", "k", "m1", "rl", "kappa"):
        par[k] = par[k] * float(rng.uniform(0.8, 1.25))
    ice = rng.random(nx) < 0.6
    h = np.where(ice, rng.choice([par["hmin"], 0.3, 1.0, 3.0], size=nx) * rng.uniform(0.5, 1.5, nx), 0.0)
    phi = np.where(ice, rng.choice([0.0, 0.05, 0.5, 0.995, 1.0], size=nx), 0.0)
    D = np.where(ice, rng.choice([0.0, par["Dmin"], 10.0, par["Dmax"]], size=nx), 0.0)
    Ei = -par["Lf"] * h * phi * np.where(rng.random(nx) < 0.8, 1.0, rng.uniform(0.0, 2.0, nx))
    Ew = par["cw"] * rng.uniform(-0.5, 12.0, nx) * (1.0 - 0.9 * phi)
    T0 = rng.uniform(-20.0, 5.0, nx)
    f = float(rng.uniform(-3.0, 3.0))
    ct = oracle.cos2pit(float(st.t[int(rng.integers(0, st.nt))]))
    geom = oracle.DiffusionGeometry(kind, st.x, par["D"])
    new, T0n, nit, ok = oracle.step_miz(ct, f, dict(Ei=Ei, Ew=Ew, h=h, D=D, phi=phi), T0.copy(), st.x, st.dt, geom, par)
    state = {k: np.ascontiguousarray(v[None], dtype=np.float64) for k, v in dict(Ei=Ei, Ew=Ew, h=h, D=D, phi=phi, T0=T0).items()}
    diag, cnt = coracle.miz_run(kid, st.x, par, st.dt, np.array([ct]), np.array([f]), None, state)
    assert ok and cnt[0] == nit and cnt[1] == 0
    for k in PROG:
        assert np.array_equal(state[k][0], new[k], equal_nan=True), k
    assert np.array_equal(state["T0"][0], T0n, equal_nan=True)
    for k in DIAG:
        assert np.array_equal(diag[k][0], new[k], equal_nan=True), k


@pytest.mark.parametrize("seed", range(6))
def test_classic_c_port_equals_numpy_on_random_states(oracle, coracle, seed):
    """Classic step from random states (enthalpies around zero, exact zeros included, so every
    Bool mask flips inside a column): NumPy and C restatements agree bit for bit."""
    rng = np.random.default_rng(900 + seed)
    nx = int(rng.choice([2, 5, 64, 129, 180]))
    st = oracle.SpaceTime("identity", nx, 2000, 1)
    par = dict(oracle.default_parameters("Classic"))
    for k in ("D", "A", "B", "S1", "a0", "ai", "Fb", "k", "cg"):
        par[k] = par[k] * float(rng.uniform(0.85, 1.2))
    E = par["cw"] * rng.uniform(-8.0, 25.0, nx)
    E[rng.random(nx) < 0.08] = 0.0
    Tg = rng.uniform(-25.0, 30.0, nx)
    f = float(rng.uniform(-3.0, 3.0))
    ti = int(rng.integers(0, st.nt))
    ct_i, ct_n = oracle.cos2pit(float(st.t[ti])), oracle.cos2pit(float(st.t[(ti + 1) % st.nt]))
    stat = oracle.ClassicStatics(st.x, nx, st.dt, par)
    with np.errstate(all="ignore"):
        new = oracle.step_classic(ct_i, ct_n, f, dict(E=E.copy(), Tg=Tg.copy()), st.x, st.dt, stat, par)
    state = dict(E=np.ascontiguousarray(E[None]), Tg=np.ascontiguousarray(Tg[None]))
    out = coracle.classic_run(st.x, par, st.dt, np.array([ct_i]), np.array([ct_n]), np.array([f]), None, state)
    for k, got in (("E", state["E"][0]), ("Tg", state["Tg"][0]), ("T", out["T"][0]), ("h", out["h"][0])):
        assert np.array_equal(got, new[k], equal_nan=True), k


def test_c_oracle_under_address_and_ub_sanitizers(tmp_path):
    """The checker itself, built with -fsanitize=address,undefined (CPU only; GPU sanitizers are not
    available on this pool): a MIZ run on both grids, a ragged multi-column case and a classic run must
    finish without a sanitizer report."""
    import subprocess
    import textwrap
    src = os.path.join(ROOT, "oracle", "ebm_oracle.c")
    exe = tmp_path / "oracle_asan"
    main = tmp_path / "main.c"
    main.write_text(textwrap.dedent("""
        #include <stdio.h>
        #include <stdlib.h>
        #include <math.h>
        int ebmo_miz_run(int, int, int, const double *, const double *, double, int, const double *, const double *,
                         const double *, double *, double *, double *, double *, double *, double *, double *, double *,
                         double *, double *, double *, long long *, int, int);
        int ebmo_classic_run(int, int, const double *, const double *, double, int, const double *, const double *,
                             const double *, const double *, double *, double *, double *, double *, int);
        static const double par[25] = {0.6, 193.0, 2.1, 9.8, 420.0, 338.0, 240.0, 0.7, 0.1, 0.4, 4.0, 2.0, 9.5, 0.0, 0.098,
                                       1e-5, 0.0, 50.4576, 1.36, 0.66, 0.5, 1.0, 156.0, 0.1, 315360.0};
        int main(void) {
            for (int kind = 0; kind < 2; ++kind) {
                const int nx = 61, ncol = 3, nsteps = 40;
                double *x = malloc(sizeof(double) * nx), *ct = malloc(sizeof(double) * nsteps), *ft = calloc(nsteps, sizeof(double));
                double fcol[3] = {-1.0, 0.0, 2.0};
                for (int k = 0; k < nx; ++k) x[k] = kind ? sin((k + 0.5) * M_PI / 2.0 / nx) : (k + 0.5) / nx;
                for (int s = 0; s < nsteps; ++s) ct[s] = cos(2.0 * M_PI * (s + 0.5) / 2000.0);
                double *f[11];
                for (int i = 0; i < 11; ++i) f[i] = calloc((size_t)nx * ncol, sizeof(double));
                long long cnt[2] = {0, 0};
                for (int imex = 0; imex < 2; ++imex)           /* the reference's scheme, then the implicit-diffusion extension */
                    ebmo_miz_run(kind, nx, ncol, x, par, 1.0 / 2000.0, nsteps, ct, ft, fcol, f[0], f[1], f[2], f[3], f[4], f[5],
                                 f[6], f[7], f[8], f[9], f[10], cnt, 1, imex);
                printf("miz kind %d: solves %lld\\n", kind, cnt[0]);
                for (int i = 0; i < 11; ++i) free(f[i]);
                free(x); free(ct); free(ft);
            }
            {
                const int nx = 37, ncol = 2, nsteps = 25;
                double *x = malloc(sizeof(double) * nx), *c0 = malloc(sizeof(double) * nsteps), *c1 = malloc(sizeof(double) * nsteps);
                double *ft = calloc(nsteps, sizeof(double)), *E = malloc(sizeof(double) * nx * ncol), *Tg = malloc(sizeof(double) * nx * ncol);
                double *T = malloc(sizeof(double) * nx * ncol), *h = malloc(sizeof(double) * nx * ncol);
                for (int k = 0; k < nx; ++k) x[k] = (k + 0.5) / nx;
                for (int s = 0; s < nsteps; ++s) { c0[s] = cos(2.0 * M_PI * (s + 0.5) / 2000.0); c1[s] = cos(2.0 * M_PI * (s + 1.5) / 2000.0); }
                for (int i = 0; i < nx * ncol; ++i) { double ts = 30.0 - 45.0 * x[i % nx] * x[i % nx]; Tg[i] = ts; E[i] = ts >= 0 ? 9.8 * ts : 9.5 * ts / 7.5; }
                ebmo_classic_run(nx, ncol, x, par, 1.0 / 2000.0, nsteps, c0, c1, ft, NULL, E, Tg, T, h, 1);
                printf("classic: E[0] %.6f\\n", E[0]);
                free(x); free(c0); free(c1); free(ft); free(E); free(Tg); free(T); free(h);
            }
            return 0;
        }
    """))
    subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-ffp-contract=off", "-o", str(exe), str(main), src, "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
    assert "miz kind 1" in out.stdout and "classic" in out.stdout
