"""Seeded fuzz of the fused paths against one launch per step, GPU only (no oracle: this is about identity, and the per-step path
is what the oracle is compared with elsewhere): random meridian length 2 ... 4096, grid kind, model (MIZ / MIZ_IMEX), cells per
thread, the loose random states of test_randomized_states_one_step (open water, thin and thick ice, phi = 0 / 1 / between,
floes at Dmin / Dmax / 0, inconsistent Ei, random warm starts; every sixth seed keeps Inf / NaN in), random K and run length,
random choice of the fused-K kernel where both exist (ebm_options.fused_state_in_lds).
Compared bit for bit: ebm_run vs ebm_run_fused (state and diagnostics), ebm_integrate with one launch per step vs fused
stretches (final state, winter, summer, avg), and the solve counters.
    python tests/tools/fuzz_fused.py [first] [count]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
PROG = ("Ei", "Ew", "h", "D", "phi")
ALL = PROG + ("T0", "Tw", "Ti", "n", "E", "T")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 600
bad, sizes = [], {}
for seed in range(first, first + count):
    rng = np.random.default_rng(77000 + seed)
    nlat = int(rng.choice([2, 3, 5, 17, 64, 65, 127, 180, 256, 300, 511, 777, 1024, 1441, 1537, 2049, 2500, 3072, 3333, 4095, 4096]))
    ncol = int(rng.integers(1, 4))
    kind = "sin" if rng.random() < 0.6 else "identity"
    model = "MIZ_IMEX" if rng.random() < 0.3 else "MIZ"
    cells = 2 if (seed % 2 and nlat <= 1536 and model == "MIZ") else 4
    nt = int(max(2000, 0.7 * nlat * nlat)) if model == "MIZ" else 2000
    st = pkg.SpaceTime(kind, nlat, 2000, 1)                                  # for x only
    dt = 1.0 / nt
    par = pkg.default_parameters("MIZ")
    for k in ("D", "A", "B", "S1", "a0", "ai", "Fb", "k", "m1", "rl", "kappa"):
        par[k] = par[k] * float(rng.uniform(0.8, 1.25))
    shape = (ncol, nlat)
    ice = rng.random(shape) < 0.6
    h = np.where(ice, rng.choice([par["hmin"], 0.3, 1.0, 3.0], size=shape) * rng.uniform(0.5, 1.5, shape), 0.0)
    phi = np.where(ice, rng.choice([0.0, 0.05, 0.5, 0.995, 1.0], size=shape), 0.0)
    D = np.where(ice, rng.choice([0.0, par["Dmin"], 10.0, par["Dmax"]], size=shape), 0.0)
    Ei = -par["Lf"] * h * phi * np.where(rng.random(shape) < 0.8, 1.0, rng.uniform(0.0, 2.0, shape))
    Ew = par["cw"] * rng.uniform(-0.5, 12.0, shape) * (1.0 - 0.9 * phi)
    if seed % 6 != 5:
        Ew = np.where(phi == 1.0, 0.0, Ew)
    state = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in dict(Ei=Ei, Ew=Ew, h=h, D=D, phi=phi, T0=rng.uniform(-20.0, 5.0, shape)).items()}
    fcol = rng.uniform(-5.0, 5.0, ncol)
    K = int(rng.choice([2, 3, 5, 8, 13, 64]))
    nsteps = int(rng.integers(5, 31))
    t0 = int(rng.integers(0, 1000))
    tt = np.array([pkg.cos2pit((i + 0.5) * dt) for i in range(t0, t0 + 64)])
    f_steps = rng.uniform(-2.0, 2.0, nsteps)
    wi, si = 2, max(4, nsteps // 2)
    in_lds = [None, False, True][int(rng.integers(0, 3))]                   # which fused-K kernel where both exist
    out = {}
    for how in ("run", "run_fused", "integrate", "integrate_fused"):
        with pkg.Engine(model, st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), dt, ncol, device=0,
                        cells_per_thread=cells, integrate_steps_per_launch=(1 if how == "integrate" else K),
                        fused_state_in_lds=in_lds) as eng:
            eng.set_state(state)
            eng.set_column_forcing(fcol)
            if how.startswith("run"):
                eng.set_time_table(tt)
                eng.run(0, nsteps, f_steps, True, steps_per_launch=(K if how == "run_fused" else 1))
                saved = {}
            else:
                eng.set_time_table(tt[:nsteps])
                saved = eng.integrate(nsteps, 1, f_steps, True, wi, si, ("E", "T", "phi", "Ti"), want_raw=False)
            out[how] = (eng.get_state(ALL), saved, eng.counters())
    what = f"seed {seed}: {model} {kind} {nlat}x{ncol}, {cells} cells per thread, K = {K}, {nsteps} steps, fused state in LDS: {in_lds}"
    fails = []
    for a, b in (("run", "run_fused"), ("run", "integrate"), ("integrate", "integrate_fused")):
        fails += [f"{b}:{k}" for k in ALL if not np.array_equal(out[a][0][k], out[b][0][k], equal_nan=True)]
        if out[a][2]["solves"] != out[b][2]["solves"] or out[a][2]["cap_hits"] != out[b][2]["cap_hits"]:
            fails.append(f"{b}:counters")
    fails += [f"saved:{k}" for k in ("winter", "summer", "avg")
              if not np.array_equal(out["integrate"][1][k], out["integrate_fused"][1][k], equal_nan=True)]
    if out["run_fused"][2]["launches"] != -(-nsteps // K):
        fails.append("run_fused:launches")
    sizes[nlat] = sizes.get(nlat, 0) + 1
    if fails:
        bad.append(seed)
        print(what, "FAILED:", fails[:6], flush=True)
    elif seed % 50 == 0:
        print(what, "ok", flush=True)
print(f"{count} seeds from {first}: {len(bad)} failures {bad[:20]}; meridian lengths drawn: {dict(sorted(sizes.items()))}")
sys.exit(1 if bad else 0)
