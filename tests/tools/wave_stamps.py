"""Diagnostic: per-WAVE s_memtime timeline of the MIZ kernel (needs the -DEBM_STAMPS build of the
library as argv[1]; argv[2] = number of columns, default 2048).  Shows how the 16 waves of a
workgroup move through phase D and when each has issued its stores: min / median / max over the
waves, median over the workgroups, in ticks since the workgroup's first instruction."""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from energybalancemodel_jl_amd import _lib
_lib.LIB_PATH = sys.argv[1]
lib = _lib.load()
nlat, ncol, nt = 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 2048, 1048576
st = pkg.SpaceTime("sin", nlat, nt, 1)
par = pkg.default_parameters("MIZ")
eng = pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol)
eng.set_column_forcing(0.5 * np.sin(2 * np.pi * np.arange(ncol) / ncol))
eng.set_time_table(st.t)
eng.run(0, 300, None, False); eng.sync()
lib.ebm_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
lib.ebm_debug_stamps(eng._h, None)
eng.run(300, 3, None, False); eng.sync()
buf = np.zeros(ncol * (16 + 128), dtype=np.uint64)
lib.ebm_debug_stamps(eng._h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
s = buf[ncol * 16:].reshape(ncol, 16, 8).astype(np.int64)
names = ["first instruction", "inputs arrived", "phase D starts", "Tbar halo done", "first pair done",
         "arithmetic done", "stores issued"]
ref = s[:, :, 0].min(axis=1)[:, None]
for k, name in enumerate(names):
    rel = s[:, :, k] - ref
    print(f"{name:18s} first wave {np.median(rel.min(axis=1)):8.0f}   median wave {np.median(np.median(rel, axis=1)):8.0f}"
          f"   last wave {np.median(rel.max(axis=1)):8.0f}")
wg = ncol // 2
print(f"workgroup {wg}, rows = waves 0..15, columns as above:")
print(s[wg][:, :7] - s[wg][:, 0].min())
