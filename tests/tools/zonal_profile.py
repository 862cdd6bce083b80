#!/usr/bin/env python
"""Repeated ebm_zonal_diffusion calls on one rank's grid of BASELINE configs[4] (32 members of 1024 x 512), for
`rocprofv3 --kernel-trace --stats` (the kernel's time) and `--pmc FETCH_SIZE` / `WRITE_SIZE` passes:

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tests/tools/zonal_profile.py [nmember] [reps] [nlat] [nlon]

Prints the shape and the algorithmic bytes of one sweep (read temp, write the forward sweep's dp into Z's array, read it back,
read temp again, write Z: 40 B per cell; the two coefficient tables, [nlon][nlat] each and shared by all members, come from
the caches)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

nmember = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nlat = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
nlon = int(sys.argv[4]) if len(sys.argv) > 4 else 512
pkg = graft.load_package()
st = pkg.SpaceTime("sin", nlat, 2000, 1)
par = pkg.default_parameters("MIZ")
eng = pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, nlon * nmember, device=0)
T = np.random.default_rng(0).normal(0.0, 10.0, (nlon * nmember, nlat))
for _ in range(reps):
    U, Z = eng.zonal_diffusion(T, nlon)
cells = nlat * nlon * nmember
print(f"zonal sweep: {nlat} lat x {nlon} lon x {nmember} members = {cells} cells, {reps} calls; algorithmic bytes per sweep "
      f"(40 B/cell, + 8 with U written as well) = {48 * cells / 1e6:.1f} MB; max|U| {np.max(np.abs(U)):.3f}")
eng.close()
