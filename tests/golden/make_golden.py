"""Generates tests/golden/*.npz from the NumPy oracle (oracle/ebm_oracle.py).

The reference's own golden file (test/solution_1year.jld2) is absent from the mount and Julia
cannot run here, so these vectors are ORACLE outputs, not reference outputs ("parity
unpinned", see oracle/ebm_oracle.py).  The configuration is the reference test's
(test/runtests.jl:22-32: SpaceTime{sin}(180, 2000, 1), Forcing(0.0), default MIZ parameters,
all-zero initial state) plus identity-grid and classic variants.

For every saved step k the full state AFTER step k is stored (prognostics, warm start T0 and
diagnostics), and also after step k+1, so a test can load state k, take one step on the GPU
and compare with k+1 without trajectory drift.

    python tests/golden/make_golden.py
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ebm_oracle as o

MIZ_STEPS = (1, 2, 10, 100, 522, 1000, 1548, 1999)      # 522 / 1548 = winter / summer index


def miz(kind):
    st = o.SpaceTime(kind, 180, 2000, 1)
    par = o.default_parameters("MIZ")
    geom = o.DiffusionGeometry(kind, st.x, par["D"])
    vars_ = {k: np.zeros(180) for k in ("Ei", "Ew", "h", "D", "phi")}
    T0 = np.zeros(180)
    out = {"x": st.x, "t": st.t}
    want = set(MIZ_STEPS) | {k + 1 for k in MIZ_STEPS}
    for s in range(1, 2001):
        vars_, T0, nit, ok = o.step_miz(o.cos2pit(float(st.t[s - 1])), 0.0, vars_, T0, st.x, st.dt, geom, par)
        assert ok
        if s in want:
            for k, v in vars_.items():
                out[f"s{s}_{k}"] = v
            out[f"s{s}_T0"] = T0
    np.savez_compressed(os.path.join(HERE, f"miz_{kind}_180_2000.npz"), **out)


def classic():
    st = o.SpaceTime("identity", 180, 2000, 1)
    par = o.default_parameters("Classic")
    stat = o.ClassicStatics(st.x, st.nx, st.dt, par)
    Ts = 30.0 - 45.0 * st.x ** 2                       # SURVEY §8(d) cfg3 warm start
    vars_ = dict(E=np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5), Tg=Ts.copy())
    out = {"x": st.x, "t": st.t, "s0_E": vars_["E"], "s0_Tg": vars_["Tg"]}
    ct = [o.cos2pit(float(t)) for t in st.t]
    for s in range(1, 2001):
        i = o.classic_time_index(float(st.t[s - 1]), st.dt, st.nt)
        vars_ = o.step_classic(ct[i - 1], ct[i % st.nt], 0.0, vars_, st.x, st.dt, stat, par)
        if s in (1, 2, 10, 11, 522, 523, 1999, 2000):
            for k, v in vars_.items():
                out[f"s{s}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "classic_identity_180_2000.npz"), **out)


if __name__ == "__main__":
    miz("sin")
    miz("identity")
    classic()
    print("golden written")
