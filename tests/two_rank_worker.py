"""Worker of the 2-process Engine test (tests/test_gpu_configs.py): launched by
``python -m torch.distributed.run --nproc-per-node 2`` BEFORE the pytest process has touched the GPU
(tests/conftest.py starts it at session start).  Both ranks share GPU 0 (gloo backend: RCCL refuses
two ranks on one device); each rank drives the HIP library on its own block of columns
(shard_columns), reduces its per-column diagnostics on the device, and rank 0 gathers.

    two_rank_worker.py OUT.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

# the configuration both the workers and the single-process reference run (test_gpu_configs.py)
NLAT, NLON, NMEMBER, NT, NSTEPS = 256, 8, 5, 8192, 48


def member_forcing(ncol):
    member = np.arange(ncol) // NLON
    return -2.0 + 4.0 * member / max(NMEMBER - 1, 1)


def run_block(pkg, columns, device=0):
    """Integrate the given global columns on one GPU; returns (T field, hemispheric means T / phi)."""
    st = pkg.SpaceTime("sin", NLAT, NT, 1)
    par = pkg.default_parameters("MIZ")
    fcol = member_forcing(NLON * NMEMBER)[columns]
    init = {k: np.zeros(NLAT) for k in ("Ei", "Ew", "h", "D", "phi")}
    run = pkg.EnsembleRun("MIZ", st, par, init, fcol=fcol, device=device)
    run.run(NSTEPS)
    out = (run.field_tensor("T"), run.hemispheric_mean_tensor("T"), run.hemispheric_mean_tensor("phi"))
    run.close()
    return out


def main():
    import torch.distributed as dist
    out_path = sys.argv[1]
    pkg = graft.load_package()
    dist.init_process_group("gloo")
    rank, ws = dist.get_rank(), dist.get_world_size()
    ncol = NLON * NMEMBER
    # rank 0 owns the inputs; everybody else receives them (I/O broadcast)
    got = pkg.broadcast_inputs(dict(fcol=member_forcing(ncol)) if rank == 0 else None, dist)
    assert np.array_equal(got["fcol"], member_forcing(ncol))
    sl = pkg.shard_columns(ncol, ws, rank)
    T, hmT, hmphi = run_block(pkg, np.arange(ncol)[sl])
    full_T = pkg.gather_columns(T, ncol, dist)
    full_hmT = pkg.gather_columns(hmT, ncol, dist)
    full_hmphi = pkg.gather_columns(hmphi, ncol, dist)
    print(f"rank {rank}/{ws}: columns {sl.start}:{sl.stop} on cuda:0, library {pkg.LIB_PATH}", flush=True)
    if rank == 0:
        np.savez(out_path, T=full_T, hmT=full_hmT, hmphi=full_hmphi)
        print(f"rank 0 gathered {full_T.shape[0]} columns x {full_T.shape[1]} latitudes", flush=True)
    else:
        assert full_T is None and full_hmT is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
