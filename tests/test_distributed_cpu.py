"""World-size-2 test of the multi-GPU path's plumbing on CPU (gloo): block-sharding of the
columns, the I/O broadcast of the inputs from rank 0, global column indexing of the per-column
forcing, and the I/O gather.  The per-rank
compute is done by the oracle here (there is no GPU), which is exactly what the HIP path is
checked against in the -m gpu tests; columns are independent, so no collective is involved
in the time loop itself."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import __graft_entry__ as graft
    import torch.distributed as dist
    pkg = graft.load_package()
    o, c_oracle = graft.load_oracle()
    dist.init_process_group("gloo")
    rank, ws = dist.get_rank(), dist.get_world_size()
    ncol, nlat, nsteps = 11, 96, 25
    st = pkg.SpaceTime("sin", nlat, 4000, 1)
    par = pkg.default_parameters("MIZ")
    # rank 0 owns the inputs; every other rank receives them (I/O broadcast, SURVEY 8(e))
    mine = None
    if rank == 0:
        mine = dict(x=st.x, fcol=-2.0 + 4.0 * np.arange(ncol) / (ncol - 1),    # cfg5-style member forcing
                    params=np.array([par[k] for k in sorted(par)]), ftab=np.zeros((2, nsteps)))
    got = pkg.broadcast_inputs(mine, dist)
    assert np.array_equal(got["x"], st.x) and got["ftab"].shape == (2, nsteps)
    assert np.array_equal(got["params"], np.array([par[k] for k in sorted(par)]))
    fcol_all = got["fcol"]
    sl = pkg.shard_columns(ncol, ws, rank)
    n_loc = sl.stop - sl.start
    state = {{k: np.zeros((n_loc, nlat)) for k in ("Ei", "Ew", "h", "D", "phi", "T0")}}
    ct = np.array([pkg.cos2pit(float(t)) for t in st.t[:nsteps]])
    co = c_oracle.COracle()
    diag, _ = co.miz_run(1, st.x, dict(par), st.dt, ct, np.zeros(nsteps), fcol_all[sl], state)
    hm = pkg.hemispheric_mean(diag["T"], st.x)                     # per-member diagnostic
    full_T = pkg.gather_columns(diag["T"], ncol, dist)
    full_hm = pkg.gather_columns(hm[:, None], ncol, dist)
    if rank == 0:
        np.savez({out!r}, T=full_T, hm=full_hm[:, 0])
    else:
        assert full_T is None
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_sharded_ensemble_matches_single_process(tmp_path, pkg, coracle):
    out = str(tmp_path / "gathered.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
        env=env, timeout=240)
    got = np.load(out)
    ncol, nlat, nsteps = 11, 96, 25
    st = pkg.SpaceTime("sin", nlat, 4000, 1)
    par = pkg.default_parameters("MIZ")
    fcol = -2.0 + 4.0 * np.arange(ncol) / (ncol - 1)
    state = {k: np.zeros((ncol, nlat)) for k in ("Ei", "Ew", "h", "D", "phi", "T0")}
    ct = np.array([pkg.cos2pit(float(t)) for t in st.t[:nsteps]])
    diag, _ = coracle.miz_run(1, st.x, dict(par), st.dt, ct, np.zeros(nsteps), fcol, state)
    assert np.array_equal(got["T"], diag["T"], equal_nan=True)
    assert np.array_equal(got["hm"], pkg.hemispheric_mean(diag["T"], st.x), equal_nan=True)
