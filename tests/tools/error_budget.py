#!/usr/bin/env python
"""Measured rounding-error budget per configuration (GPU box; writes the table the parity tests'
tolerances are derived from — copy the output to profiles/rNN_error_budget.txt).

For every configuration: spin up with the fp64 oracle, hand the SAME fp64 state (and warm start) to
(a) the GPU, (b) the fp64 C oracle, (c) the same C source evaluated in 80-bit extended precision
(oracle/libebm_oracle_ld.so: same formulas, same fp64 inputs, ~2000x less rounding in between), advance
all three N steps, and print the maximum scaled error |a-b|/max(1,|b|) over all cells of all ten
variables (+T0) of
    GPU vs extended      — the GPU path's own rounding error
    oracle vs extended   — the fp64 oracle's rounding error (Thomas solve, same physics)
    GPU vs oracle        — what the parity tests measure
The tridiagonal T0 / Tg solves carry cond(J)*eps ~ nlat^2 * 1e-16 of forward error whichever
backward-stable algorithm is used; everything else is bit-identical arithmetic.

    python tests/tools/error_budget.py > gpurun_out/error_budget.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

PROG = ("Ei", "Ew", "h", "D", "phi")
DIAG = ("Tw", "Ti", "n", "E", "T")
ALL = PROG + ("T0",) + DIAG


def scaled(a, b):
    same_nan = np.array_equal(np.isnan(a), np.isnan(b))
    a, b = np.nan_to_num(a), np.nan_to_num(b)
    e = float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0
    return e, same_nan


def worst(x, y, names):
    errs = {k: scaled(x[k], y[k]) for k in names}
    k = max(errs, key=lambda n: errs[n][0])
    return errs[k][0], k, all(v[1] for v in errs.values())


def miz_case(pkg, co, cl, kind, nlat, ncol, nt, spin, checkpoints, fcol_amp=2.0, cells=4):
    """cells: launch geometry of the GPU run (4 = what every throughput-sized workload uses; 2 = what
    the library picks for a few short meridians)."""
    os.environ["EBM_CELLS_PER_THREAD"] = str(cells)
    st = pkg.SpaceTime(kind, nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    kid = 0 if kind == "identity" else 1
    fcol = fcol_amp * np.sin(2 * np.pi * (np.arange(ncol) + 0.3) / ncol) if ncol > 1 else np.zeros(1)
    ct = np.array([pkg.cos2pit(float(t)) for t in st.t[: spin + max(checkpoints)]])
    base = {k: np.zeros((ncol, nlat)) for k in PROG + ("T0",)}
    if spin:
        co.miz_run(kid, st.x, dict(par), st.dt, ct[:spin], np.zeros(spin), fcol, base)
    ref = {k: v.copy() for k, v in base.items()}
    ext = {k: v.copy() for k, v in base.items()}
    eng = pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol, device=0)
    eng.set_state(base)
    eng.set_column_forcing(fcol)
    eng.set_time_table(st.t)
    done = 0
    rows = []
    for n in checkpoints:
        seg = ct[spin + done: spin + n]
        d_ref, _ = co.miz_run(kid, st.x, dict(par), st.dt, seg, np.zeros(n - done), fcol, ref)
        d_ext, _ = cl.miz_run(kid, st.x, dict(par), st.dt, seg, np.zeros(n - done), fcol, ext)
        # the extended run is handed back fp64-rounded states between checkpoints: re-widened exactly
        eng.run(spin + done, n - done, None, True)
        got = eng.get_state(ALL)
        r, e = dict(ref, **d_ref), dict(ext, **d_ext)
        rows.append((n, worst(got, e, ALL), worst(r, e, ALL), worst(got, r, ALL), worst(got, r, PROG)))
        done = n
    eng.close()
    return rows


def classic_case(pkg, co, cl, nlat, ncol, checkpoints, cells=4):
    os.environ["EBM_CELLS_PER_THREAD"] = str(cells)
    st = pkg.SpaceTime("identity", nlat, 2000, 1)
    par = pkg.default_parameters("Classic")
    Ts = 30.0 - 45.0 * st.x ** 2
    E0 = np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5)
    fcol = 0.5 * np.sin(2 * np.pi * np.arange(ncol) / ncol)
    ct = np.array([pkg.cos2pit(float(t)) for t in st.t])
    ref = dict(E=np.tile(E0, (ncol, 1)), Tg=np.tile(Ts, (ncol, 1)))
    ext = {k: v.copy() for k, v in ref.items()}
    eng = pkg.Engine("Classic", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol, device=0)
    eng.set_state(ref)
    eng.set_column_forcing(fcol)
    eng.set_time_table(st.t)
    names = ("E", "Tg", "T", "h")
    done, rows = 0, []
    for n in checkpoints:
        idx = np.arange(done, n)
        o_ref = co.classic_run(st.x, dict(par), st.dt, ct[idx], ct[(idx + 1) % st.nt], np.zeros(n - done), fcol, ref)
        o_ext = cl.classic_run(st.x, dict(par), st.dt, ct[idx], ct[(idx + 1) % st.nt], np.zeros(n - done), fcol, ext)
        eng.run(done, n - done, None, True)
        got = eng.get_state(names)
        r, e = dict(ref, **o_ref), dict(ext, **o_ext)
        rows.append((n, worst(got, e, names), worst(r, e, names), worst(got, r, names), worst(got, r, ("E", "Tg"))))
        done = n
    eng.close()
    return rows


def main():
    pkg = graft.load_package()
    _, c_oracle = graft.load_oracle()
    co, cl = c_oracle.COracle(), c_oracle.COracle(extended=True)
    print("# error budget: max scaled error |a-b|/max(1,|b|) over all cells and variables (worst variable named)")
    print("# ext = oracle/ebm_oracle.c in 80-bit extended precision between fp64 inputs and outputs")
    print(f"{'configuration':52s} {'steps':>6s} {'GPU vs ext':>22s} {'oracle vs ext':>22s} {'GPU vs oracle':>22s} "
          f"{'GPU vs oracle, prognostics':>28s} NaNs")
    cases = [
        ("MIZ sin 180 x1 nt=2000 from zero (reference test)", lambda: miz_case(pkg, co, cl, "sin", 180, 1, 2000, 0, (1, 2, 10, 50))),
        ("MIZ sin 180 x1 nt=2000 from zero, 2 cells per thread", lambda: miz_case(pkg, co, cl, "sin", 180, 1, 2000, 0, (1, 2, 10, 50), cells=2)),
        ("MIZ identity 180 x1 nt=2000 from zero", lambda: miz_case(pkg, co, cl, "identity", 180, 1, 2000, 0, (1, 10, 50))),
        ("MIZ sin 1000 x5 nt=60000 spin 20", lambda: miz_case(pkg, co, cl, "sin", 1000, 5, 60000, 20, (1, 20))),
        ("MIZ identity 1024 x8 nt=262144 spin 50", lambda: miz_case(pkg, co, cl, "identity", 1024, 8, 262144, 50, (1, 20))),
        ("MIZ sin 1024 x8 nt=65536 from zero (cfg5 columns)", lambda: miz_case(pkg, co, cl, "sin", 1024, 8, 65536, 0, (1, 24))),
        ("MIZ sin 1440 x2 nt=131072 spin 50 (cfg2)", lambda: miz_case(pkg, co, cl, "sin", 1440, 2, 131072, 50, (1, 20, 60))),
        ("MIZ sin 2048 x3 nt=262144 spin 50", lambda: miz_case(pkg, co, cl, "sin", 2048, 3, 262144, 50, (1, 20))),
        ("MIZ sin 4096 x6 nt=1048576 spin 50 (cfg4 columns)", lambda: miz_case(pkg, co, cl, "sin", 4096, 6, 1048576, 50, (1, 10, 40))),
        ("MIZ sin 4096 x8 nt=1048576 from zero, f 0.5 (bench)", lambda: miz_case(pkg, co, cl, "sin", 4096, 8, 1048576, 0, (10, 20, 30, 40, 50, 100, 200), 0.5)),
        ("Classic identity 180 x2", lambda: classic_case(pkg, co, cl, 180, 2, (1, 10, 522))),
        ("Classic identity 1024 x16 (cfg3 columns)", lambda: classic_case(pkg, co, cl, 1024, 16, (1, 40, 120))),
    ]
    for name, fn in cases:
        for n, g_e, r_e, g_r, g_p in fn():
            fmt = lambda t: f"{t[0]:10.3e} ({t[1]:>3s})"                      # noqa: E731
            print(f"{name:52s} {n:6d} {fmt(g_e):>22s} {fmt(r_e):>22s} {fmt(g_r):>22s} {fmt(g_p):>28s} "
                  f"{'same' if (g_e[2] and r_e[2] and g_r[2]) else 'DIFFER'}", flush=True)


if __name__ == "__main__":
    main()
