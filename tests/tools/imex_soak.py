import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
nlat, ncol, nt, years = 4096, 256, 2000, 12
st = pkg.SpaceTime("sin", nlat, nt, years)
par = pkg.default_parameters("MIZ")
run = pkg.EnsembleRun("MIZ_IMEX", st, par, {"Ew": par["cw"] * np.maximum(30.0 - 45.0 * st.x ** 2, 0.0), "Ei": np.zeros(nlat),
                      "h": np.zeros(nlat), "D": np.zeros(nlat), "phi": np.zeros(nlat)}, fcol=np.linspace(-2.0, 2.0, ncol), device=0)
t0 = time.perf_counter()
for y in range(years):
    q = []
    for part in range(4):
        run.run(nt // 4)
        q.append((run.engine.hemispheric_mean("T"), run.engine.hemispheric_mean("phi")))
dt = time.perf_counter() - t0
mid = ncol // 2
print(f"{years} years x {nt} steps of {nlat} x {ncol} (extension): {dt:.2f} s = {dt / (years * nt) * 1e3:.3f} ms/step; counters {run.engine.counters()}")
print("year 12 quarters, member with f ~ 0: <T> " + " ".join(f"{a[mid]:7.3f}" for a, b in q) + "  <phi> " + " ".join(f"{b[mid]:6.4f}" for a, b in q))
print("                  f = -2:            <T> " + " ".join(f"{a[0]:7.3f}" for a, b in q) + "  <phi> " + " ".join(f"{b[0]:6.4f}" for a, b in q))
print("                  f = +2:            <T> " + " ".join(f"{a[-1]:7.3f}" for a, b in q) + "  <phi> " + " ".join(f"{b[-1]:6.4f}" for a, b in q))
s = run.state(("Ei", "Ew", "h", "D", "phi"))
print("finite:", all(np.isfinite(v).all() for v in s.values()))
