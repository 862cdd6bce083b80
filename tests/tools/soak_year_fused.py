"""One full model year (nt = 2^20 steps) of the headline configuration TWICE, in lockstep: one launch per step (ebm_run) and
64 steps per launch with the state resident in LDS (ebm_run_fused -> miz_resident_kernel).  After every chunk of 2^16 steps
all twelve fields of the two runs are compared bit for bit, and so are the solve counters.  About 6 minutes of GPU time."""
import time, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
nlat, ncol, nt = 4096, 2048, 1048576
ALL = ("Ei", "Ew", "h", "D", "phi", "T0", "Tw", "Ti", "n", "E", "T")
st = pkg.SpaceTime("sin", nlat, nt, 1)
par = pkg.default_parameters("MIZ")
engs = {}
for name in ("single", "fused"):
    e = pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol)
    e.set_column_forcing(0.5 * np.sin(2 * np.pi * np.arange(ncol) / ncol))
    e.set_time_table(st.t)
    engs[name] = e
chunk, done, worst = 65536, 0, 0
while done < nt:
    ms = {}
    for name, e in engs.items():
        t0 = time.perf_counter()
        e.run(done, chunk, None, True, steps_per_launch=(64 if name == "fused" else 1))
        e.sync()
        ms[name] = (time.perf_counter() - t0) / chunk * 1e3
    done += chunk
    a, b = engs["single"].get_state(ALL), engs["fused"].get_state(ALL)
    diff = [k for k in ALL if not np.array_equal(a[k], b[k], equal_nan=True)]
    ca, cb = engs["single"].counters(), engs["fused"].counters()
    same_counts = ca["solves"] == cb["solves"] and ca["cap_hits"] == cb["cap_hits"] == 0
    worst += len(diff) + (0 if same_counts else 1)
    print(f"t = {done / nt:5.3f} yr: {ms['single']:.4f} / {ms['fused']:.4f} ms per step (single / fused), ice fraction "
          f"{np.mean(a['phi'] > 0):.3f}, solves per column-step so far {ca['solves'] / (done * ncol):.5f} (same in both: {same_counts}), "
          f"fields that differ: {diff if diff else 'none'}", flush=True)
print("launches:", engs["single"].counters()["launches"], "vs", engs["fused"].counters()["launches"])
print("BIT-IDENTICAL over the whole year" if worst == 0 else f"DIFFERENCES in {worst} checks")
sys.exit(0 if worst == 0 else 1)
