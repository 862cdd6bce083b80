"""First-contact GPU check: run several configurations through the HIP path and print the
error of every variable against the oracle.  Not a pytest file; used via gpurun while
bringing kernels up."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
o, c_oracle = g.load_oracle()
co = c_oracle.COracle()
PROG = ("Ei", "Ew", "h", "D", "phi")
ALL = PROG + ("T0", "Tw", "Ti", "n", "E", "T")


def report(tag, got, ref):
    worst = 0.0
    for k in ref:
        a, b = np.asarray(got[k]), np.asarray(ref[k])
        nan_ok = np.array_equal(np.isnan(a), np.isnan(b))
        a0, b0 = np.nan_to_num(a), np.nan_to_num(b)
        err = np.abs(a0 - b0) / np.maximum(1.0, np.abs(b0))
        i = np.unravel_index(np.argmax(err), err.shape)
        worst = max(worst, float(err.max()))
        print(f"  {tag:28s} {k:4s} max_scaled_err={err.max():.3e} at {i} got={a[i]!r} ref={b[i]!r} nan_match={nan_ok} bitexact={np.array_equal(a, b, equal_nan=True)}")
    return worst


def miz_case(kind, nlat, ncol, nsteps, nt, spin=0, fcol_amp=0.0):
    st = pkg.SpaceTime(kind, nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    kid = 0 if kind == "identity" else 1
    rng = np.random.default_rng(1)
    state = {k: np.zeros((ncol, nlat)) for k in PROG + ("T0",)}
    fcol = fcol_amp * np.sin(2 * np.pi * np.arange(ncol) / max(ncol, 1)) if fcol_amp else None
    ct = np.array([pkg.cos2pit(float(t)) for t in st.t])
    if spin:
        co.miz_run(kid, st.x, par, st.dt, ct[:spin], np.zeros(spin), fcol, state)
    eng = pkg.Engine("MIZ", st.grid_kind, st.x, c_oracle.COracle.par_vector(par), st.dt, ncol)
    for k in PROG + ("T0",):
        eng.set_field(k, state[k])
    if fcol is not None:
        eng.set_column_forcing(fcol)
    eng.set_time_table(st.t)
    t0 = time.time()
    eng.run(spin, nsteps, None, True)
    eng.sync()
    tg = time.time() - t0
    got = eng.get_state(ALL)
    cnt = eng.counters()
    info = eng.launch_info()
    eng.close()
    idx = (spin + np.arange(nsteps)) % nt
    t0 = time.time()
    diag, ocnt = co.miz_run(kid, st.x, par, st.dt, ct[idx], np.zeros(nsteps), fcol, state)
    tc = time.time() - t0
    ref = dict(state); ref.update(diag)
    tag = f"miz {kind[:3]} {nlat}x{ncol} s{spin}+{nsteps}"
    print(f"{tag}: gpu {tg*1e3:.1f} ms, C oracle {tc*1e3:.1f} ms, gpu counters {cnt}, oracle {ocnt}, launch {info}")
    return report(tag, got, ref)


def classic_case(nlat, ncol, nsteps, nt):
    st = pkg.SpaceTime("identity", nlat, nt, 1)
    par = pkg.default_parameters("Classic")
    Ts = 30 - 45 * st.x ** 2
    E0 = np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5)
    state = dict(E=np.tile(E0, (ncol, 1)), Tg=np.tile(Ts, (ncol, 1)))
    fcol = 0.5 * np.sin(2 * np.pi * np.arange(ncol) / ncol)
    ct = np.array([pkg.cos2pit(float(t)) for t in st.t])
    eng = pkg.Engine("Classic", "identity", st.x, c_oracle.COracle.par_vector(par), st.dt, ncol)
    eng.set_state(state); eng.set_column_forcing(fcol); eng.set_time_table(st.t)
    eng.run(0, nsteps, None, True); eng.sync()
    got = eng.get_state(("E", "Tg", "T", "h")); eng.close()
    idx = np.arange(nsteps) % nt
    out = co.classic_run(st.x, par, st.dt, ct[idx], ct[(idx + 1) % nt], np.zeros(nsteps), fcol, state)
    ref = dict(state); ref.update(out)
    tag = f"classic {nlat}x{ncol} {nsteps}"
    print(tag)
    return report(tag, got, ref)


cases = [
    lambda: miz_case("sin", 180, 1, 1, 2000),
    lambda: miz_case("sin", 180, 1, 10, 2000),
    lambda: miz_case("identity", 180, 1, 10, 2000),
    lambda: miz_case("sin", 180, 3, 200, 2000, fcol_amp=2.0),
    lambda: miz_case("sin", 180, 1, 100, 2000, spin=1500),
    lambda: miz_case("sin", 255, 2, 50, 4000),
    lambda: miz_case("sin", 1000, 2, 20, 60000),
    lambda: miz_case("sin", 1440, 1, 20, 131072, spin=200),
    lambda: miz_case("sin", 4096, 4, 10, 1048576, spin=50),
    lambda: miz_case("identity", 1024, 4, 10, 131072, spin=50),
    lambda: classic_case(180, 2, 100, 2000),
    lambda: classic_case(1024, 8, 20, 2000),
]
if __name__ == "__main__":
    for i, c in enumerate(cases):
        try:
            w = c()
            print(f"case {i}: worst {w:.3e}\n", flush=True)
        except Exception:
            traceback.print_exc()
            print(f"case {i}: FAILED\n", flush=True)
