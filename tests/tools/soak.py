import time, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
nlat, ncol, nt = 4096, 2048, 1048576
st = pkg.SpaceTime("sin", nlat, nt, 1)
par = pkg.default_parameters("MIZ")
eng = pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol)
eng.set_column_forcing(0.5 * np.sin(2 * np.pi * np.arange(ncol) / ncol))
eng.set_time_table(st.t)
done = 0
for chunk in (2000, 8000, 10000):
    t0 = time.perf_counter(); eng.run(done, chunk, None, True); eng.sync(); dt = time.perf_counter() - t0
    done += chunk
    s = eng.get_state(("Ei", "Ew", "h", "D", "phi", "T", "E"))
    c = eng.counters()
    print(f"after {done} steps: {dt/chunk*1e3:.4f} ms/step, solves/col-step {c['solves']/(c['steps']*ncol):.4f}, cap_hits {c['cap_hits']}, "
          f"ice fraction {np.mean(s['phi']>0):.3f}, phi==1 {np.mean(s['phi']==1):.3f}, max h {s['h'].max():.3f}, T range [{np.nanmin(s['T']):.2f},{np.nanmax(s['T']):.2f}], "
          f"non-finite prognostics {sum(int((~np.isfinite(s[k])).sum()) for k in ('Ei','Ew','h','D','phi'))}")
