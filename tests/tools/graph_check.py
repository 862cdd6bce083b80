import time, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
for nlat, nt in ((180, 2000), (1440, 131072)):
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    res = {}
    for mode in ("0", "1"):
        os.environ["EBM_GRAPH"] = mode
        eng = pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, 1)
        eng.set_time_table(st.t)
        eng.run(0, 200, None, True); eng.sync()
        t0 = time.perf_counter(); eng.run(200, 6000, None, True); eng.sync(); dt = time.perf_counter() - t0
        res[mode] = eng.get_state(("Ei", "Ew", "h", "D", "phi", "T0", "T"))
        print(f"nlat={nlat} graph={mode}: {dt/6000*1e6:.2f} us/step, counters {eng.counters()}")
        eng.close()
    same = all(np.array_equal(res["0"][k], res["1"][k], equal_nan=True) for k in res["0"])
    print("  graph replay bit-identical to direct launches:", same)
