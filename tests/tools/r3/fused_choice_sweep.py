"""Where is the crossover between the two fused-K kernels?  ms per step of ebm_run_fused (K = 64) with the state in registers
and in LDS, over the column count, for three meridian lengths (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
par = pkg.default_parameters("MIZ")
for nlat, nt in ((180, 2000), (1024, 65536), (2048, 262144)):
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    for ncol in (64, 128, 256, 384, 512, 768, 1024, 2048, 4096):
        ms = {}
        for name, lds in (("registers", False), ("lds", True)):
            with pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol, device=0,
                            cells_per_thread=4, fused_state_in_lds=lds, use_graph=False) as eng:
                eng.set_column_forcing(0.5 * np.sin(2 * np.pi * np.arange(ncol) / ncol))
                eng.set_time_table(st.t)
                eng.run(0, 512, None, False, steps_per_launch=64)
                eng.sync()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    eng.run(512, 1024, None, False, steps_per_launch=64)
                    eng.sync()
                    best = min(best, (time.perf_counter() - t0) / 1024 * 1e3)
                ms[name] = best
        print(f"{nlat:5d} x {ncol:5d}: registers {ms['registers']*1e3:8.2f} us/step, lds {ms['lds']*1e3:8.2f} us/step  -> {'lds' if ms['lds'] < ms['registers'] else 'registers'}", flush=True)
