"""Per-step error growth of the classic model at 64 steps per year (GPU box): GPU vs fp64 oracle, GPU vs the 80-bit
build of the oracle, fp64 oracle vs 80-bit — shows that the jump at step 21 of tests/test_gpu_parity.py::
test_integrate_randomized_surface[3] is the model amplifying rounding in every fp64 path alike.  python tests/tools/classic_growth.py"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'oracle')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import ebm_oracle as o
from c_oracle import COracle
c64, c80 = COracle(), COracle(extended=True)
nlat, nt = 16, 64
st = pkg.SpaceTime("identity", nlat, nt, 3)
par = pkg.default_parameters("Classic")
Ts = 30.0 - 45.0 * st.x ** 2
E0 = np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5)
ct = np.array([pkg.cos2pit(float(t)) for t in st.t])
N = 40
idx = [o.classic_time_index(float(st.t[i % nt]), st.dt, nt) for i in range(N)]
cti = np.array([ct[i - 1] for i in idx]); ctp = np.array([ct[i % nt] for i in idx])
f = np.full(N, 0.5)
eng = pkg.Engine("Classic", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, 1, device=0)
eng.set_state(dict(E=E0[None].copy(), Tg=Ts[None].copy())); eng.set_time_table(st.t)
s64 = dict(E=E0[None].copy(), Tg=Ts[None].copy()); s80 = dict(E=E0[None].copy(), Tg=Ts[None].copy())
for n in range(N):
    eng.run(n, 1, f[n:n+1], True)
    got = eng.get_state(("E", "Tg"))
    c64.classic_run(st.x, dict(par), st.dt, cti[n:n+1], ctp[n:n+1], f[n:n+1], None, s64)
    c80.classic_run(st.x, dict(par), st.dt, cti[n:n+1], ctp[n:n+1], f[n:n+1], None, s80)
    e = lambda a, b: max(np.max(np.abs(a[k] - b[k]) / np.maximum(1, np.abs(b[k]))) for k in ("E", "Tg"))
    print(n + 1, "gpu-o64 %.2e  gpu-o80 %.2e  o64-o80 %.2e" % (e(got, s64), e(got, s80), e(s64, s80)))
