#!/usr/bin/env python
"""Hysteresis ensemble: every member of the ensemble ramps the forcing up and down at its own
rate (the experiment the reference's `Forcing{false}` exists for, src/infrastructure.jl:208-241),
all members advance in one kernel launch per step, and only per-member scalars — hemispheric
means of temperature and ice concentration — leave the GPU once per year.

    python examples/hysteresis_ensemble.py [--members 64] [--nlat 180] [--nt 2000]

Multi-GPU: launch with torch.distributed.run; the members are sharded across the ranks and the
yearly diagnostics gathered to rank 0 (RCCL for I/O only, no collective in the time loop).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=64)
    ap.add_argument("--nlat", type=int, default=180)
    ap.add_argument("--nt", type=int, default=2000)
    args = ap.parse_args()
    pkg = graft.load_package()

    dist = None
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        dist.init_process_group("nccl")

    # member m: hold 2 years, warm by 8 W/m2 at rate r_m, hold 2 years, cool back at -r_m
    rates = [8.0 / k for k in (2, 4, 8, 16)]
    forcings = [pkg.Forcing(0.0, 8.0, 0.0, (2, 2), (rates[m % 4], -rates[m % 4])) for m in range(args.members)]
    years = max(f.domain[4] for f in forcings) + 2
    st = pkg.SpaceTime("sin", args.nlat, args.nt, years)
    par = pkg.default_parameters("MIZ")
    mine = pkg.shard_columns(args.members, world, rank)
    init = {k: np.zeros(st.nx) for k in ("Ei", "Ew", "h", "D", "phi")}
    run = pkg.EnsembleRun("MIZ", st, par, init, forcings=forcings[mine],
                          device=int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0)
    if rank == 0:
        print(f"{args.members} members x {args.nlat} latitudes, {years} years of {args.nt} steps on {world} GPU(s)")
    for year in range(years):
        # forcing comes from the device schedules; nobody looks at the state inside a year: 64 steps per launch
        run.run(args.nt, diag_last=True, steps_per_launch=64)
        # per-member means reduced on the device, gathered to rank 0 as device tensors (RCCL), one host copy there
        T = pkg.gather_columns(run.hemispheric_mean_tensor("T")[:, None], args.members, dist)
        phi = pkg.gather_columns(run.hemispheric_mean_tensor("phi")[:, None], args.members, dist)
        if rank == 0:
            f_now = [f(year + 1.0 - 0.5 / args.nt) for f in forcings[:4]]
            print(f"year {year + 1:3d}  f = " + " ".join(f"{v:5.2f}" for v in f_now) +
                  "   <T> = " + " ".join(f"{v:6.2f}" for v in T[:4, 0]) +
                  "   <phi> = " + " ".join(f"{v:5.3f}" for v in phi[:4, 0]))
    run.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
