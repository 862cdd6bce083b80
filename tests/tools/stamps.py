"""Diagnostic: per-phase s_memtime timeline of the MIZ kernel (needs the -DEBM_STAMPS build of
the library passed as argv[1]).  Reads SHARES of the timeline, not absolute kernel time."""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from energybalancemodel_jl_amd import _lib
_lib.LIB_PATH = sys.argv[1]
lib = _lib.load()
nlat, ncol, nt = 4096, 2048, 1048576
st = pkg.SpaceTime("sin", nlat, nt, 1)
par = pkg.default_parameters("MIZ")
eng = pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol)
eng.set_column_forcing(0.5 * np.sin(2 * np.pi * np.arange(ncol) / ncol))
eng.set_time_table(st.t)
eng.run(0, 300, None, False); eng.sync()
lib.ebm_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
lib.ebm_debug_stamps(eng._h, None)
eng.run(300, 3, None, False); eng.sync()
nwg = eng.launch_info()["workgroups"]
buf = np.zeros(ncol * (16 + 128), dtype=np.uint64)     # per-workgroup stamps, then per-wave ones (wave_stamps.py)
lib.ebm_debug_stamps(eng._h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
s = buf[:ncol * 16].reshape(ncol, 16).astype(np.int64)[:nwg]
names = ["start", "A:loads+Tw", "A:r halo", "A:rhs+sync", "B:g halo+rows", "B:partition solve", "B:check+sync",
         "D:tb+halo", "D:pair0", "D:pair1", "-", "-", "-", "-", "-", "end"]
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15]
d = np.diff(s[:, order], axis=1)
tot = (s[:, 15] - s[:, 0])
print("median total cycles per workgroup (s_memtime ticks):", np.median(tot))
for i, k in enumerate(order[1:]):
    print(f"  {names[k]:22s} median {np.median(d[:, i]):9.0f}  ({100*np.median(d[:, i])/np.median(tot):5.1f} %)")
t0 = s[:, 0] - s[:, 0].min()
print("wg start spread (ticks): p50", np.median(t0), "max", t0.max(), " end max", (s[:, 15] - s[:, 0].min()).max())
