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
    # One call for the whole experiment: savesol! runs inside the step kernel (annual-mean sums of T and phi,
    # winter / summer snapshots), and only the per-member hemispheric means of those seasonal outputs — the
    # numbers the reference's plot_seasonal draws (src/plot.jl:173-225) — leave the device.
    hm = run.seasonal_means(years, ("T", "phi"))
    # [variable, year, member] -> members first, gathered to rank 0 (device tensors over RCCL when sharded)
    cols = {k: np.ascontiguousarray(np.moveaxis(v, 2, 0)) for k, v in hm.items()}
    full = {k: pkg.gather_columns(v, args.members, dist) for k, v in cols.items()}
    if rank == 0:
        print("annual-mean hemispheric temperature and ice-covered area 2*pi*<phi> (annual mean / winter / summer), first 4 members")
        for year in range(years):
            f_now = [f(year + 0.5) for f in forcings[:4]]
            Tm = full["avg"][:4, 0, year]
            area = [2.0 * np.pi * full[k][:4, 1, year] for k in ("avg", "winter", "summer")]
            print(f"year {year + 1:3d}  f = " + " ".join(f"{v:5.2f}" for v in f_now) +
                  "   <T> = " + " ".join(f"{v:6.2f}" for v in Tm) +
                  "   A_i = " + " ".join(f"{a:5.2f}/{w:5.2f}/{s_:5.2f}" for a, w, s_ in zip(*area)))
    run.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
