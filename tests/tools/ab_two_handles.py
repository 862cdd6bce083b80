#!/usr/bin/env python
"""Experiment: columns are independent, so the headline grid's 2048 meridians can be split over
several handles — each with its own stream, driven from its own host thread — to let one group's workgroups fill the CUs
another group's store tail and launch boundary leave idle.  Aggregate cell-steps/s of 1, 2 and 4 handles over the same
4096 x 2048 grid, same spin-up, same steps.

    python tests/tools/ab_two_handles.py [steps] [repeats]
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nlat, ncol, nt, spinup = 4096, 2048, 1048576, 2000
pkg = graft.load_package()
st = pkg.SpaceTime("sin", nlat, nt, 1)
par = pkg.default_parameters("MIZ")
pv = pkg.engine.param_vector(par, pkg.default_parval)
fcol = 0.5 * np.sin(2.0 * np.pi * np.arange(ncol) / ncol)

for nh in (1, 2, 4, 1):
    per = ncol // nh
    engs = []
    for i in range(nh):
        e = pkg.Engine("MIZ", st.grid_kind, st.x, pv, st.dt, per, device=0)
        e.set_column_forcing(fcol[i * per:(i + 1) * per])
        e.set_time_table(st.t)
        e.run(0, spinup, None, False)
        engs.append(e)
    for e in engs:
        e.sync()
    clock = spinup

    def drive(e, first, n):
        e.run(first, n, None, False)
        e.sync()

    def block(n):
        global clock
        ts = [threading.Thread(target=drive, args=(e, clock, n)) for e in engs]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        clock += n
        return dt

    block(1500)                      # pre-roll
    times = sorted(block(steps) for _ in range(repeats))
    med = times[len(times) // 2]
    info = engs[0].launch_info()
    print(f"{nh} handle(s) x {per} columns: {med * 1e3 / steps:.4f} ms per step of the whole grid "
          f"({nlat * ncol * steps / med / 1e9:.2f} G cell-steps/s; blocks {[round(t * 1e3 / steps, 4) for t in times]}; "
          f"prefetch-ahead workgroups per CU {info})", flush=True)
    for e in engs:
        e.close()
