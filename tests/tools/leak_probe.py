"""Free device memory after each create / use / destroy cycle (GPU box): python tests/tools/leak_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
torch.cuda.synchronize()
base = torch.cuda.mem_get_info()[0]
for rep in range(24):
    model = "Classic" if rep % 4 == 3 else "MIZ"
    small = rep % 5 == 0
    nlat, ncol = (180, 4) if small else (2048, 4096)
    st = pkg.SpaceTime("identity" if model == "Classic" else "sin", nlat, 2000 if small else 262144, 1)
    par = pkg.default_parameters(model)
    with pkg.Engine(model, st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol, device=0) as eng:
        eng.set_time_table(st.t)
        eng.run(0, 130 if small else 3, None, True, steps_per_launch=(8 if rep % 2 else 1))
        eng.hemispheric_mean("T")
        if small:
            names = ("E", "T", "h") if model == "Classic" else ("E", "T", "phi")
            eng.integrate(st.nt, 1, None, True, st.winter.inx, st.summer.inx, names)
            try:
                eng.integrate(st.nt, 1, None, True, st.winter.inx, st.summer.inx, ("E", "E"))
            except pkg.EBMError:
                pass
    torch.cuda.synchronize()
    print(rep, model, "small" if small else "big", "fused" if rep % 2 else "k1", (base - torch.cuda.mem_get_info()[0]) / 2**20, "MiB below start")
