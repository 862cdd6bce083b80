"""The headline model at the card's memory scale (GPU box): 4096 latitudes x 524,288 meridians — 16 GiB per field,
176 GiB of state of the 288 GB — stepped from the zero state; per-step time, and the per-column hemispheric means of
E and phi (reduced on the device, 4 MiB each) of columns that share a forcing value compared bitwise across the slab.
python tests/tools/capacity_run.py [ncol]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
nlat, nt = 4096, 1048576
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
st = pkg.SpaceTime("sin", nlat, nt, 1)
par = pkg.default_parameters("MIZ")
fcol = -2.0 + 4.0 * (np.arange(ncol) % 64) / 63.0
t0 = time.perf_counter()
with pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol, device=0) as eng:
    eng.set_column_forcing(fcol)
    eng.set_time_table(st.t)
    print(f"state: {11 * ncol * nlat * 8 / 2**30:.0f} GiB in {time.perf_counter() - t0:.1f} s", flush=True)
    eng.run(0, 20, None, False)
    eng.sync()
    for blk in range(3):
        t0 = time.perf_counter()
        eng.run(20 + 10 * blk, 10, None, blk == 2)
        eng.sync()
        dt = (time.perf_counter() - t0) / 10
        print(f"block {blk}: {dt * 1e3:.2f} ms per step of {ncol * nlat / 1e9:.2f} G cells = {ncol * nlat / dt / 1e9:.1f} G cell-steps/s "
              f"= {96 * ncol * nlat / dt / 8e12 * 100:.1f} % of 8 TB/s", flush=True)
    cnt = eng.counters()
    print("solves per column-step %.4f, cap hits %d" % (cnt["solves"] / (cnt["steps"] * ncol), cnt["cap_hits"]))
    for name in ("E", "phi"):
        hm = eng.hemispheric_mean(name).reshape(ncol // 64, 64)
        same = all(np.array_equal(hm[0], hm[i]) for i in (1, ncol // 128, ncol // 64 - 1))
        print(f"hemispheric mean of {name}: replicas bitwise equal across the slab: {same}; member 0: {hm[0, 0]:.12g}, member 63: {hm[0, 63]:.12g}")
        assert same and np.isfinite(hm).all()
