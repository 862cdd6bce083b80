#!/usr/bin/env python
"""bench.py — grid-cell-steps/s and achieved HBM GB/s of the MIZ step on MI355X.

Workload (BASELINE.json configs[3]): 2-D 4096 x 2048 MIZ model on the sin
grid — 2048 independent meridians of 4096 latitudes per GPU — nt = 1,048,576 steps/year
(explicit stability), all-zero initial prognostics as in the reference test, per-column
forcing f[lon] = 0.5*sin(2*pi*lon/nlon), `--spinup` untimed spin-up steps so ice and open
water and the T0 solve are all live.  A "step" is one time step of the whole grid = one kernel
launch (K = 1 step per launch).  With N GPUs every rank integrates its own 4096 x 2048 block
(columns are independent: weak scaling, no collective in the time loop).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline.achieved` = 96 B per cell-step (read + write of
Ei, Ew, h, D, phi and the T0 warm start) x cells per launch / average launch
duration measured with HIP events on the stream the kernels run on.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

BYTES_PER_CELL_STEP = 96.0       # MIZ, state only
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (model, grid kind, nlat, ncol, nt)
    "miz_4096x2048": ("MIZ", "sin", 4096, 2048, 1048576),
    "miz_1024x512x32": ("MIZ", "sin", 1024, 512 * 32, 65536),
    "miz_1440x1": ("MIZ", "sin", 1440, 1, 131072),
    "miz_2048x4096": ("MIZ", "sin", 2048, 4096, 262144),     # same cells and bytes as the headline, half-length meridians
    "miz_8192x1024": ("MIZ", "sin", 8192, 1024, 4194304),    # longest supported meridians (16 cells per thread, no LDS stash)
    "classic_1024x512": ("Classic", "identity", 1024, 512, 2000),
}


def cpu_baseline(pkg, wl, st, par, state, fcol, first_step, budget_s):
    """Time the oracle's C port (OpenMP over columns) on a bounded sample of the same workload:
    the first `ncols` columns of the spun-up state for `nsteps` steps."""
    o, c_oracle = graft.load_oracle()
    co = c_oracle.COracle(openmp=True)
    cores = co.max_threads()
    model, kind, nlat, ncol, nt = wl
    kid = 0 if kind == "identity" else 1
    ncols = min(ncol, max(cores * 8, 64))
    sub = {k: np.ascontiguousarray(v[:ncols]) for k, v in state.items()}
    fc = None if fcol is None else np.ascontiguousarray(fcol[:ncols])
    i0 = first_step % nt

    def table(n):
        return np.array([pkg.cos2pit(float(st.t[(i0 + i) % nt])) for i in range(n)])

    # probe (after a one-step warm-up of the thread pool), then size the sample to the budget
    co.miz_run(kid, st.x, dict(par), st.dt, table(1), np.zeros(1), fc, sub, nthreads=cores)
    probe = 4
    t0 = time.perf_counter()
    co.miz_run(kid, st.x, dict(par), st.dt, table(probe), np.zeros(probe), fc, sub, nthreads=cores)
    rate = ncols * nlat * probe / (time.perf_counter() - t0)
    nsteps = int(max(8, min(20000, budget_s * rate / (ncols * nlat))))
    ct = table(nsteps)
    t0 = time.perf_counter()
    co.miz_run(kid, st.x, dict(par), st.dt, ct, np.zeros(nsteps), fc, sub, nthreads=cores)
    dt = time.perf_counter() - t0
    # the same port on ONE core, a short sample (single-thread figure beside the OpenMP one)
    one = {k: np.ascontiguousarray(v[:8]) for k, v in sub.items()}
    n1 = max(2, int(2.0 * rate / cores / (8 * nlat)))
    t0 = time.perf_counter()
    co.miz_run(kid, st.x, dict(par), st.dt, table(n1), np.zeros(n1), None if fc is None else fc[:8], one, nthreads=1)
    rate1 = 8 * nlat * n1 / (time.perf_counter() - t0)
    return {
        "value": ncols * nlat * nsteps / dt, "unit": "grid-cell-steps/s", "cores": cores,
        "value_single_core": rate1,
        "kind": "port",
        "sample": f"{nsteps} steps x {ncols} columns x {nlat} latitudes of the spun-up state "
                  f"(oracle/ebm_oracle.c, gcc -O2 -fopenmp, {dt:.1f} s)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup", type=int, default=2000)
    ap.add_argument("--workload", default="miz_4096x2048", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-budget", type=float, default=30.0, help="seconds of CPU baseline work (0 = skip)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; EBM_BENCH_BACKEND=gloo lets several ranks share a device to rehearse the
    # N > 1 path on a one-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("EBM_BENCH_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    pkg = graft.load_package()
    wl = WORKLOADS[args.workload]
    model, kind, nlat, ncol, nt = wl
    st = pkg.SpaceTime(kind, nlat, nt, 1)
    par = pkg.default_parameters(model)
    lon = np.arange(ncol) + rank * ncol                      # this rank's block of columns
    if args.workload == "miz_1024x512x32":
        # BASELINE configs[4] / SURVEY 8(d) cfg5: 32 members of 512 meridians per GPU, member m of 256
        # forced by the constant f_m = -2 + 4 m/255 W/m2
        fcol = -2.0 + 4.0 * ((lon // 512) % 256) / 255.0
    else:
        fcol = 0.5 * np.sin(2.0 * np.pi * lon / ncol)
    eng = pkg.Engine(model, st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval),
                     st.dt, ncol, device=device)
    if model == "Classic":
        Ts = 30.0 - 45.0 * st.x ** 2
        E0 = np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5)
        eng.set_field("E", np.tile(E0, (ncol, 1)))
        eng.set_field("Tg", np.tile(Ts, (ncol, 1)))
    eng.set_column_forcing(fcol)
    eng.set_time_table(st.t)
    step = 0
    eng.run(step, args.spinup, None, False); step += args.spinup
    eng.sync()
    # state after spin-up (for the CPU baseline sample and the ice fraction)
    cpu = None
    ice_fraction = None
    if model == "MIZ":
        state = {k: eng.get_field(k) for k in ("Ei", "Ew", "h", "D", "phi", "T0")}
        ice_fraction = float(np.mean(state["phi"] > 0))
        if rank == 0 and world == 1 and args.cpu_budget > 0:
            cpu = cpu_baseline(pkg, wl, st, par, state, fcol, step, args.cpu_budget)
        del state
    eng.run(step, args.warmup, None, False); step += args.warmup
    eng.sync()
    eng.reset_counters()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    eng.run(step, args.steps, None, False)
    ev_ms = eng.timer_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    cnt = eng.counters()
    info = eng.launch_info()
    eng.close()

    cells = nlat * ncol
    bpc = BYTES_PER_CELL_STEP if model == "MIZ" else 32.0
    launch_s = ev_ms * 1e-3 / args.steps
    achieved = bpc * cells / launch_s / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        with open(pmc) as fh:
            traffic = json.load(fh).get(args.workload)
    out = {
        "metric": "grid-cell-steps/sec (2D 4096x2048 MIZ model)" if args.workload == "miz_4096x2048"
                  else f"grid-cell-steps/sec ({args.workload})",
        "value": cells * world * args.steps / elapsed,
        "unit": "grid-cell-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {model} model, {nlat} lat x {ncol} meridians per GPU, "
                        f"{kind} grid, nt={nt}, {args.spinup} spin-up steps from zero state, "
                        + ("f[member]=-2+4*member/255" if args.workload == "miz_1024x512x32"
                           else "f[lon]=0.5*sin(2*pi*lon/nlon)"),
            "steps_per_launch": 1,
            "ice_covered_fraction": ice_fraction,
            "mean_tridiagonal_solves_per_column_step": (cnt["solves"] / (args.steps * ncol)) if model == "MIZ" else 1.0,
            "t0_cap_hits": cnt["cap_hits"],
            "threads_per_workgroup": info["threads"], "cells_per_thread": info["cells_per_thread"],
            "lds_bytes_per_workgroup": info["lds_bytes"],
            "parallelism": f"columns sharded over {world} GPU(s), no in-loop collective",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": "miz_step_kernel" if model == "MIZ" else "classic_step_kernel",
            "algorithmic_bytes_per_launch": bpc * cells,
            "avg_launch_ms": launch_s * 1e3,
            # transparency: the kernel carries the T0 warm start as a bit mask, so it moves fewer bytes
            # than the contract's 96 B per cell-step; this is the rate of the counter-measured traffic
            "achieved_traffic_gbs": (traffic / launch_s / 1e9) if traffic else None,
            "frac_traffic": (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
        },
        "cpu_baseline": cpu,
    }
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
