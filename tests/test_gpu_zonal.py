"""GPU tests (-m gpu) of ebm_zonal_diffusion — the zonal partner of the meridional diffusion operator as a backward-Euler
substep (an EXTENSION defined in include/ebm_hip.h; not in the reference: "parity unpinned" by
construction).  The HIP kernels (one lane per latitude walking the longitudes, periodic Thomas carrying the last unknown;
circles of 256 longitudes and more cut into 4 ... 32 segments with a reduced periodic system of the segment ends)
are held to the checker's two restatements — which use two other algorithms (Fourier diagonalisation; Thomas +
Sherman-Morrison) — and to closed forms that involve no restatement at all."""
import numpy as np
import pytest

from conftest import record_error

pytestmark = pytest.mark.gpu


def make_engine(pkg, st, ncol, **kw):
    par = pkg.default_parameters("MIZ")
    return pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol, device=0, **kw), par


@pytest.mark.parametrize("kind,nlat,nlon,nmember,nt,cells", [
    ("sin", 180, 3, 2, 2000, 4), ("sin", 180, 7, 1, 2000, 2),          # the shortest circles; both launch geometries
    ("identity", 255, 16, 3, 2000, 4),                                 # ragged meridian (pitch 256)
    ("sin", 1000, 64, 2, 60000, 2), ("sin", 1024, 512, 2, 2000, 4),    # one rank's grid of BASELINE configs[4]: a up to 3e5 at the pole; 8 segments
    ("sin", 4096, 64, 1, 2000, 4), ("sin", 2, 5, 1, 2000, 4),
    ("sin", 180, 256, 2, 2000, 2), ("identity", 300, 320, 1, 2000, 4),   # circles cut into 4 segments (of 64 and 80 unknowns)
    ("sin", 512, 2048, 1, 2000, 4),                                     # 32 segments of 64
    ("sin", 96, 255, 1, 2000, 4),                                       # 255 longitudes: not divisible, one segment
])
def test_zonal_substep_matches_both_restatements(pkg, oracle, coracle, kind, nlat, nlon, nmember, nt, cells):
    """U and Z against the NumPy (Fourier) and C (Sherman-Morrison) restatements of the header's definition, on random
    fields with NaN-free data: three different algorithms for the same periodic systems, so agreement is to the rounding
    the conditioning of the systems allows (their condition number is 1 + 4 a_k: up to 1e6 at the polar circle of 1024
    latitudes x 512 longitudes), not to bits.  Also: the defining equations row by row, and zero net convergence per circle."""
    st = pkg.SpaceTime(kind, nlat, nt, 1)
    ncol = nlon * nmember
    rng = np.random.default_rng(nlat * 7 + nlon)
    T = rng.normal(0.0, 12.0, (ncol, nlat))
    eng, par = make_engine(pkg, st, ncol, cells_per_thread=cells)
    with eng:
        assert eng.launch_info()["cells_per_thread"] == cells
        U, Z = eng.zonal_diffusion(T, nlon)
        U2, Z2 = eng.zonal_diffusion(T, nlon)                          # tables cached: same bits
        assert np.array_equal(U, U2) and np.array_equal(Z, Z2)
    par = dict(par)
    Un, Zn = oracle.zonal_substep(T, st.x, st.dt, nlon, par)
    Uc, Zc = coracle.zonal(st.x, par, st.dt, nlon, T)
    scale = float(np.max(np.abs(T)))
    eU = max(float(np.max(np.abs(U - Un))), float(np.max(np.abs(U - Uc)))) / scale
    eZ = max(float(np.max(np.abs(Z - Zn))), float(np.max(np.abs(Z - Zc)))) / (scale * par["cw"] / st.dt)
    between = float(np.max(np.abs(Un - Uc))) / scale
    record_error(f"zonal substep {kind} {nlat} x {nlon} x {nmember}: U vs both restatements (they differ by {between:.1e})", "U", eU, 1e-12)
    record_error(f"zonal substep {kind} {nlat} x {nlon} x {nmember}: Z vs both restatements, in units of |T| cw/dt", "Z", eZ, 1e-12)
    assert eU <= 1e-12 and eZ <= 1e-12, (eU, eZ)
    dl = 2.0 * np.pi / nlon
    a = (st.dt / par["cw"]) * par["D"] / (((1.0 - st.x) * (1.0 + st.x)) * dl * dl)
    Um = U.reshape(nmember, nlon, nlat)
    lhs = (1.0 + 2.0 * a) * Um - a * (np.roll(Um, 1, axis=1) + np.roll(Um, -1, axis=1))
    assert np.max(np.abs(lhs - T.reshape(nmember, nlon, nlat)) / (1.0 + 4.0 * a)) <= 8e-15 * scale
    assert np.max(np.abs(Z.reshape(nmember, nlon, nlat).sum(axis=1))) <= 1e-11 * np.max(np.abs(Z)) * nlon


@pytest.mark.parametrize("nlat,nlon", [(48, 24), (1024, 512)])
def test_zonal_substep_has_the_closed_form_of_fourier_modes(pkg, nlat, nlon):
    """No restatement involved: T = A(x) cos(m lambda) is an eigenvector of every circle's system, so
    U = T / (1 + a_k 4 sin^2(m dlambda / 2)) and Z = -(D / ((1-x^2) dlambda^2)) 4 sin^2(m dlambda/2) U, for the zonally
    uniform mode (left alone: Z = 0 to rounding), long and short waves and the two-grid-point wave; one member per mode."""
    st = pkg.SpaceTime("sin", nlat, 2000, 1)
    modes = (0, 1, 2, 5, nlon // 2)
    eng, par = make_engine(pkg, st, nlon * len(modes))
    dl = 2.0 * np.pi / nlon
    c = par["D"] / (((1.0 - st.x) * (1.0 + st.x)) * dl * dl)
    a = (st.dt / par["cw"]) * c
    A = np.linspace(-3.0, 9.0, nlat)
    lam = np.arange(nlon) * dl
    T = np.concatenate([np.cos(m * lam)[:, None] * A[None, :] for m in modes])
    with eng:
        U, Z = eng.zonal_diffusion(T, nlon)
    for i, m in enumerate(modes):
        sl = slice(i * nlon, (i + 1) * nlon)
        mu = 4.0 * np.sin(m * dl / 2.0) ** 2
        want = T[sl] / (1.0 + a * mu)
        eU = float(np.max(np.abs(U[sl] - want))) / np.max(np.abs(A))
        Zexact = -(c * mu) * want
        # the circles' systems have condition number 1 + 4 a_k — 1.2e6 at the polar circle of 1024 x 512 (forward error of
        # any stable solve: up to 1e-10) — so the bar scales with it: 2e-17 x max a_k = 6e-12 there, ten times what is measured
        # (5.8e-13), and 1e-13 on the small grid
        bar = max(1e-13, 2e-17 * float(np.max(a)))
        record_error(f"zonal substep {nlat} x {nlon}: Fourier mode m = {m} against its closed form (max a_k = {float(np.max(a)):.1e})", "U", eU, bar)
        assert eU <= bar, (m, eU, bar)
        # Z = (U - T) cw/dt carries U's rounding times cw/dt; for the zonally uniform mode that is all there is (Z = 0)
        assert np.max(np.abs(Z[sl] - Zexact)) <= 2.0 * bar * np.max(np.abs(A)) * par["cw"] / st.dt + 1e-9 * np.max(np.abs(Zexact)), m


def test_zonal_substep_arguments(pkg):
    st = pkg.SpaceTime("sin", 64, 2000, 1)
    eng, par = make_engine(pkg, st, 12)
    T = np.zeros((12, 64))
    with eng:
        for nlon, msg in ((2, "nlon >= 3"), (5, "multiple of nlon"), (0, "nlon >= 3")):
            with pytest.raises(pkg.EBMError, match=msg):
                eng.zonal_diffusion(T, nlon)
        U, Z = eng.zonal_diffusion(T + 3.0, 4)                        # uniform in longitude
        assert np.allclose(U, 3.0, rtol=0, atol=1e-14) and np.max(np.abs(Z)) < 1e-8
        U, Z = eng.zonal_diffusion(T + 3.0, 12)                       # a different nlon: tables rebuilt
        assert np.allclose(U, 3.0, rtol=0, atol=1e-14)
    stc = pkg.SpaceTime("identity", 64, 2000, 1)
    parc = pkg.default_parameters("Classic")
    with pkg.Engine("Classic", stc.grid_kind, stc.x, pkg.engine.param_vector(parc, pkg.default_parval), stc.dt, 12, device=0) as engc:
        with pytest.raises(pkg.EBMError, match="MIZ handle"):
            engc.zonal_diffusion(T, 4)
