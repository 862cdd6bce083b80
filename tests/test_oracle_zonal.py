"""CPU tests of the zonal half of the EXTENSION: the zonal diffusion operator as a backward-Euler
substep, ebm_zonal_diffusion, defined in include/ebm_hip.h.  It is NOT in the reference — "parity unpinned" by
construction.  What can be tested without a GPU: the checker's two restatements (NumPy: Fourier diagonalisation of the
circulant system; C: Thomas + Sherman-Morrison) agree, both solve the defining system, the closed forms for Fourier modes
hold, the substep moves heat along latitude circles without creating any.

Second part — why the library ships the OPERATOR and not a two-dimensional model.  The checker also holds the obvious
coupling (operator splitting: Z of the previous step's output temperature added to the diffusion term of both vertical
fluxes of the implicit-diffusion extension).  On open water it is a convergent scheme: the decay of the spherical harmonics
P_l^m(x) cos(m lambda), eigenfunctions of the spherical diffusion operator with eigenvalue -l(l+1), is reproduced at second
order in the grid spacing.  Over thin new ice it is unstable: zonal differences of rounding size grow ~50-fold per step
until the ice has thickened, and a 6 W/m2 zonal forcing contrast turns into a 20 K temperature contrast around a polar
latitude circle where the surface balance allows 0.3 K.  Both are measured below."""
import numpy as np
import pytest

PROG = ("Ei", "Ew", "h", "D", "phi")


@pytest.mark.parametrize("kind,nlat,nlon,nmember,nt", [("sin", 64, 16, 2, 2000), ("identity", 90, 7, 1, 2000), ("sin", 1024, 512, 1, 2000),
                                                     ("sin", 180, 3, 3, 500), ("sin", 33, 4, 2, 100000)])
def test_restatements_agree_and_solve_the_defining_system(oracle, coracle, kind, nlat, nlon, nmember, nt):
    o = oracle
    st = o.SpaceTime(kind, nlat, nt, 1)
    par = dict(o.default_parameters("MIZ"))
    rng = np.random.default_rng(nlat + nlon)
    T = rng.normal(0.0, 12.0, (nmember * nlon, nlat))
    U1, Z1 = o.zonal_substep(T, st.x, st.dt, nlon, par)
    U2, Z2 = coracle.zonal(st.x, par, st.dt, nlon, T)
    scale = np.max(np.abs(T))
    assert np.max(np.abs(U1 - U2)) <= 2e-13 * scale                   # measured <= 1.7e-13 (1024 x 512, a up to 3e5 at the pole)
    assert np.max(np.abs(Z1 - Z2)) <= 3e-13 * scale * par["cw"] / st.dt     # Z = (U - T) cw/dt inherits U's rounding, amplified by cw/dt
    # the defining system, row by row: (1 + 2a) U_l - a (U_{l-1} + U_{l+1}) = T_l
    dl = 2.0 * np.pi / nlon
    a = (st.dt / par["cw"]) * par["D"] / (((1.0 - st.x) * (1.0 + st.x)) * dl * dl)
    for U in (U1, U2):
        Um = U.reshape(nmember, nlon, nlat)
        lhs = (1.0 + 2.0 * a) * Um - a * (np.roll(Um, 1, axis=1) + np.roll(Um, -1, axis=1))
        assert np.max(np.abs(lhs - T.reshape(nmember, nlon, nlat)) / (1.0 + 4.0 * a)) <= 4e-15 * scale
    # Z is the discrete zonal Laplacian of U times D/((1-x^2) dlambda^2) — the backward-Euler flux convergence
    c = par["D"] / (((1.0 - st.x) * (1.0 + st.x)) * dl * dl)
    Um = U2.reshape(nmember, nlon, nlat)
    lap = c * (np.roll(Um, 1, axis=1) - 2.0 * Um + np.roll(Um, -1, axis=1))
    assert np.max(np.abs(lap.reshape(-1, nlat) - Z2)) <= 1e-9 * np.max(np.abs(Z2)) + 1e-9 * np.max(c) * scale * 1e-7
    # it moves heat along the circle and creates none
    assert np.max(np.abs(Z2.reshape(nmember, nlon, nlat).sum(axis=1))) <= 1e-11 * np.max(np.abs(Z2)) * nlon


def test_fourier_modes_have_their_closed_form(oracle, coracle):
    """T = A(x) cos(m lambda): U = T / (1 + a_k 4 sin^2(m dlambda/2)) exactly — the discrete zonal operator's eigenvalue —
    in both restatements; the zonally uniform mode (m = 0) is left alone and gives Z = 0 to rounding."""
    o = oracle
    nlat, nlon = 48, 24
    st = o.SpaceTime("sin", nlat, 2000, 1)
    par = dict(o.default_parameters("MIZ"))
    dl = 2.0 * np.pi / nlon
    a = (st.dt / par["cw"]) * par["D"] / (((1.0 - st.x) * (1.0 + st.x)) * dl * dl)
    A = np.linspace(-3.0, 9.0, nlat)
    lam = np.arange(nlon) * dl
    for m in (0, 1, 2, 5, 12):
        T = np.cos(m * lam)[:, None] * A[None, :]
        want = T / (1.0 + a * 4.0 * np.sin(m * dl / 2.0) ** 2)
        for U, Z in (o.zonal_substep(T, st.x, st.dt, nlon, par), coracle.zonal(st.x, par, st.dt, nlon, T)):
            assert np.max(np.abs(U - want)) <= 1e-14 * np.max(np.abs(A)), m
            if m == 0:
                assert np.max(np.abs(Z)) <= 1e-15 * np.max(np.abs(A)) * par["cw"] / st.dt * 8


def c_state(ncol, nlat):
    s = {k: np.zeros((ncol, nlat)) for k in PROG + ("T0", "T")}
    return s


def test_coupling_experiment_zonally_uniform_on_open_water(oracle, coracle):
    """The coupling experiment on warm open water: a zonally uniform state has Z = 0 up to the rounding of U = T (about
    1e-15 |T| cw/dt = 1e-10 W/m2), and the run equals the one-dimensional extension MIZ_IMEX column by column to 1e-12
    over 60 steps."""
    o = oracle
    nlat, nlon, nt, n = 90, 6, 2000, 60
    st = o.SpaceTime("sin", nlat, nt, 1)
    par = dict(o.default_parameters("MIZ"))
    ct = np.array([o.cos2pit(float(t)) for t in st.t[:n]])
    two = c_state(nlon, nlat)
    one = {k: np.zeros((1, nlat)) for k in PROG + ("T0",)}
    Tw = 25.0 - 12.0 * st.x ** 2
    two["Ew"][:] = par["cw"] * Tw
    two["T"][:] = Tw
    one["Ew"][:] = par["cw"] * Tw
    with np.errstate(all="ignore"):
        coracle.miz2d_run(1, st.x, par, st.dt, nlon, ct, np.full(n, 0.7), None, two)
        coracle.miz_run(1, st.x, par, st.dt, ct, np.full(n, 0.7), None, one, imex=True)
    for k in PROG:
        scale = max(1.0, np.max(np.abs(one[k])))
        assert np.max(np.abs(two[k] - two[k][:1])) <= 1e-12 * scale, k           # the members of a circle agree
        assert np.max(np.abs(two[k][0] - one[k][0])) <= 1e-12 * scale, k


def test_coupling_experiment_is_unstable_over_thin_new_ice(oracle, coracle):
    """THE FINDING.  Freeze-up from the zero state, six identical longitudes, identical forcing: the only zonal differences
    are the rounding of the periodic solve (1e-14).  Over open water they would be damped; over the thin ice of the first
    steps they grow by a factor of ~50 per step (steps 4 ... 8), because the ice surface temperature answers an enthalpy
    change dE with dT0/dE = |F| k / (h^2 (k/h + B)^2 Lf) ~ 6 K per W yr m^-2 at h = hmin — sixty times the 1/cw the
    implicit substep assumes, and without the delay a heat capacity would give — until the ice has thickened and the
    growth saturates at 1e-6.  With a wave-1 forcing of +-3 W/m2 the same mechanism produces a 20 K temperature contrast
    around a polar latitude circle within 100 steps; the surface balance (k/hmin + B = 22 W/m2/K) allows 0.3 K."""
    o = oracle
    nlat, nlon, nt = 90, 6, 2000
    st = o.SpaceTime("sin", nlat, nt, 1)
    par = dict(o.default_parameters("MIZ"))
    s = c_state(nlon, nlat)
    asym = []
    with np.errstate(all="ignore"):
        for n in range(1, 13):
            coracle.miz2d_run(1, st.x, par, st.dt, nlon, np.array([o.cos2pit(float(st.t[n - 1]))]), np.full(1, 0.7), None, s)
            asym.append(float(np.max(np.abs(s["Ei"] - s["Ei"][:1]))))
    growth = [asym[i + 1] / asym[i] for i in range(3, 7)]                      # steps 4 -> 5 ... 7 -> 8
    assert asym[2] < 1e-12 and all(g > 10.0 for g in growth), (asym, growth)   # measured: 17.6, 52.1, 50.4, 50.2
    assert 1e-8 < max(asym) < 1e-4                                             # and it saturates (measured 2.3e-6)
    s = c_state(nlon, nlat)
    fcol = 3.0 * np.cos(2.0 * np.pi * np.arange(nlon) / nlon)
    with np.errstate(all="ignore"):
        ct = np.array([o.cos2pit(float(t)) for t in st.t[:100]])
        coracle.miz2d_run(1, st.x, par, st.dt, nlon, ct, np.zeros(100), fcol, s)
    contrast = float(np.ptp(s["T"][:, 80]))
    allowed = float(np.ptp(fcol)) / (par["k"] / par["hmin"] + par["B"])
    assert contrast > 20.0 * allowed, (contrast, allowed)                      # measured 20.2 K against 0.27 K


def test_numpy_and_c_restatements_of_the_coupling_experiment_agree(oracle, coracle):
    """The coupled step in NumPy (zonal_substep + step_miz(imex, zon)) against the C driver: same physics expressions
    (bit-identical given the same Z), Z from two different solvers — so agreement to rounding, not bits — over the few
    steps before the instability above amplifies that rounding."""
    o = oracle
    nlat, nlon, nt, n = 48, 8, 2000, 4
    st = o.SpaceTime("sin", nlat, nt, 1)
    par = dict(o.default_parameters("MIZ"))
    geom = o.DiffusionGeometry("sin", st.x, par["D"])
    fcol = 3.0 * np.cos(2.0 * np.pi * np.arange(nlon) / nlon)            # a wave-1 forcing: zonal gradients develop
    ct = np.array([o.cos2pit(float(t)) for t in st.t[:n]])
    cs = c_state(nlon, nlat)
    with np.errstate(all="ignore"):
        diag, _ = coracle.miz2d_run(1, st.x, par, st.dt, nlon, ct, np.zeros(n), fcol, cs)
        v = [{k: np.zeros(nlat) for k in PROG} for _ in range(nlon)]
        T0 = [np.zeros(nlat) for _ in range(nlon)]
        T = np.zeros((nlon, nlat))
        for s in range(n):
            _, Z = o.zonal_substep(T, st.x, st.dt, nlon, par)
            for c in range(nlon):
                out, T0[c], _, _ = o.step_miz(ct[s], float(fcol[c]), v[c], T0[c], st.x, st.dt, geom, par, imex=True, zon=Z[c])
                v[c] = {k: out[k] for k in PROG}
                T[c] = out["T"]
    for k in PROG:
        got = np.array([v[c][k] for c in range(nlon)])
        assert np.max(np.abs(got - cs[k])) <= 1e-10 * max(1.0, np.max(np.abs(cs[k]))), k
    assert np.max(np.abs(T - cs["T"])) <= 1e-10 * max(1.0, np.max(np.abs(T)))
    assert np.max(np.ptp(cs["Ew"], axis=0)) > 1e-4 and np.max(np.abs(cs["T"])) > 0.0      # the columns really differ


@pytest.mark.parametrize("l,m", [(2, 2), (4, 2)])
def test_coupling_experiment_converges_to_the_spherical_harmonic_decay_on_open_water(oracle, coracle, l, m):
    """Open water without insolation (S = A = Fb = f = 0): cw dT/dt = D [ d/dx((1-x^2) dT/dx) + (1-x^2)^-1 d2T/dlambda2 ] - B T.
    T = P_l^m(x) cos(m lambda) with l - m even (zero flux at the equator, regular at the pole) decays like
    exp(-(l(l+1) D + B) t / cw).  The scheme (implicit zonal substep on the lagged T, linearly implicit meridional step) is
    first order in time and second order in space: at a time step small enough for the spatial error to dominate, doubling
    both resolutions divides the error by ~4."""
    o = oracle

    def mode(x, lam):
        s2 = (1.0 - x) * (1.0 + x)
        P = 3.0 * s2 if (l, m) == (2, 2) else 7.5 * (7.0 * x * x - 1.0) * s2          # P_2^2, P_4^2
        return np.cos(m * lam)[:, None] * P[None, :]

    def run(nlat, nlon, nt, nsteps):
        st = o.SpaceTime("sin", nlat, nt, 1)
        par = dict(o.default_parameters("MIZ"))
        par.update(S0=0.0, S1=0.0, S2=0.0, A=0.0, Fb=0.0)
        lam = np.arange(nlon) * 2.0 * np.pi / nlon
        T0f = mode(st.x, lam) + 20.0                                    # + 20 K: stays open water (Tw > Tm)
        s = c_state(nlon, nlat)
        s["Ew"] = par["cw"] * T0f
        s["T"] = T0f.copy()                                            # the previous step's output temperature
        with np.errstate(all="ignore"):
            coracle.miz2d_run(1, st.x, par, st.dt, nlon, np.ones(nsteps), np.zeros(nsteps), None, s)
        assert not s["phi"].any()
        t = nsteps * st.dt
        exact = mode(st.x, lam) * np.exp(-(l * (l + 1) * par["D"] + par["B"]) * t / par["cw"]) + 20.0 * np.exp(-par["B"] * t / par["cw"])
        return float(np.max(np.abs(s["Ew"] / par["cw"] - exact)))

    nt = 200000
    e1 = run(24, 16, nt, 4000)
    e2 = run(48, 32, nt, 4000)
    assert e2 < e1 / 3.0 and e1 < 0.05, (e1, e2)


def test_zonal_operator_arguments_are_validated(pkg):
    """ebm_zonal_diffusion without a GPU: the handle cannot even be made (no CPU path); with one, bad arguments are refused
    (tests/test_gpu_zonal.py)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    st = pkg.SpaceTime("sin", 32, 2000, 1)
    pv = pkg.engine.param_vector(pkg.default_parameters("MIZ"), pkg.default_parval)
    with pytest.raises(pkg.EBMError, match="no HIP device"):
        pkg.Engine("MIZ", st.grid_kind, st.x, pv, st.dt, 8, device=0)
