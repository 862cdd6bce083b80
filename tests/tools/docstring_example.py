"""The reference's own docstring example (src/EnergyBalanceModel.jl:17-61): integrate(:MIZ,
SpaceTime{sin}(180, 2000, 30), Forcing(0.0), default_parameters(:MIZ), zeros) — 60,000 steps, 1:57 there."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
st = pkg.SpaceTime("sin", 180, 2000, 30)
par = pkg.default_parameters("MIZ")
init = pkg.Collection({k: np.zeros(st.nx) for k in ("Ei", "Ew", "h", "D", "phi")})
for rep in range(3):
    t0 = time.perf_counter()
    sols = pkg.integrate("MIZ", st, pkg.Forcing(0.0), par, init)
    dt = time.perf_counter() - t0
    print(f"integrate(:MIZ, SpaceTime{{sin}}(180, 2000, 30), ...): {dt:.3f} s = {60000 / dt:.0f} steps/s "
          f"(reference docstring: 1:57, 511.24 steps/s); {sols}", flush=True)
print("annual-mean hemispheric T of year 30:", float(pkg.hemispheric_mean(sols.seasonal.avg.T[29], st.x)))
