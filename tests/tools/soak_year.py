"""One full model year (nt = 2^20 steps) of the headline configuration, in chunks of 2^16 steps:
per chunk ms/step, solves per column-step, cap hits, ice fraction, non-finite prognostics.
About 3 minutes of GPU time."""
import time, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
nlat, ncol, nt = 4096, 2048, 1048576
st = pkg.SpaceTime("sin", nlat, nt, 1)
par = pkg.default_parameters("MIZ")
eng = pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol)
eng.set_column_forcing(0.5 * np.sin(2 * np.pi * np.arange(ncol) / ncol))
eng.set_time_table(st.t)
chunk, done, prev = 65536, 0, dict(steps=0, solves=0, cap_hits=0)
while done < nt:
    t0 = time.perf_counter(); eng.run(done, chunk, None, True); eng.sync(); dt = time.perf_counter() - t0
    done += chunk
    c = eng.counters()
    s = eng.get_state(("Ei", "Ew", "h", "D", "phi", "T"))
    bad = sum(int((~np.isfinite(s[k])).sum()) for k in ("Ei", "Ew", "h", "D", "phi"))
    print(f"t = {done / nt:5.3f} yr: {dt / chunk * 1e3:.4f} ms/step, solves/col-step "
          f"{(c['solves'] - prev['solves']) / (chunk * ncol):.5f}, cap hits {c['cap_hits'] - prev['cap_hits']}, "
          f"ice fraction {np.mean(s['phi'] > 0):.3f}, max h {s['h'].max():.2f} m, T [{np.nanmin(s['T']):.1f}, {np.nanmax(s['T']):.1f}], "
          f"non-finite prognostics {bad}", flush=True)
    prev = c
