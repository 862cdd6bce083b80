"""Time of a step that also writes T0 and the five diagnostic fields (OUT_DIAG: what `ebm_step`, the per-call step! form,
launches) against the state-only step, headline shape (GPU box): python tests/tools/diag_step_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
nlat, ncol, nt = 4096, 2048, 1048576
st = pkg.SpaceTime("sin", nlat, nt, 1)
par = pkg.default_parameters("MIZ")
with pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol, device=0) as eng:
    eng.set_column_forcing(0.5 * np.sin(2 * np.pi * np.arange(ncol) / ncol))
    eng.set_time_table(st.t)
    eng.run(0, 2000, None, False)
    eng.sync()
    for name, diag in (("state only", False), ("with diagnostics", True), ("state only", False), ("with diagnostics", True)):
        t0 = time.perf_counter()
        for i in range(200):
            eng.run(2000 + i, 1, None, diag)
        eng.sync()
        print(f"{name}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per step")
