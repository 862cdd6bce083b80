#!/usr/bin/env python
"""bench.py — grid-cell-steps/s and achieved HBM GB/s of the MIZ step on MI355X.

Headline workload (BASELINE.json configs[3]): 2-D 4096 x 2048 MIZ model on the
sin grid — 2048 independent meridians of 4096 latitudes per GPU — nt = 1,048,576 steps/year
(explicit stability), all-zero initial prognostics as in the reference test, per-column forcing
f[lon] = 0.5*sin(2*pi*lon/nlon), `--spinup` untimed spin-up steps so ice and open water and the T0
solve are all live.  A "step" is one time step of the whole grid = one kernel launch (K = 1 step per
launch).  With N GPUs every rank integrates its own 4096 x 2048 block (columns are independent: weak
scaling, no collective in the time loop).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process — before it has made any GPU call —
starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...  bench.py
<same arguments>` as a CHILD, relays rank 0's JSON line and exits with the child's return code.  With fewer
than N devices visible it exits non-zero and says so: an N-GPU number is never reported from fewer GPUs
(EBM_BENCH_BACKEND=gloo lets ranks share a device to REHEARSE the N > 1 path on a one-GPU box; the line then
carries `"backend": "gloo"` and `devices_used` < n_gpus).  Started by a launcher (WORLD_SIZE set), `--gpus` must
equal the world size.  The line reports `n_gpus`, the backend, the world size the process group saw and every
rank's own block timings (`ranks`).

Order of events: spin-up -> pre-roll (>= 0.25 s of untimed steps, so that the GPU is at its working
clocks whatever `--warmup` says) -> W warm-up steps -> `--repeats` blocks of EXACTLY K steps, each
bracketed by barrier + synchronize on both sides and timed with the host clock (max over ranks) and
with HIP events on the stream the kernels run on -> only then the state download, the ice fraction
and the CPU baseline.  `ms_per_step` / `value` are the MEDIAN block; every block is printed.

Other workloads (never the headline; `metric` names them):
  --workload miz_180x1 / miz_1440x1 [--steps-per-launch K]   1-D shapes of configs[0] / [1]; with K > 1
        the fused-K path (ebm_run_fused: K steps per launch, state in registers or LDS), reported separately
  --workload miz_1024x512x32_integrate   ebm_integrate with the annual-mean sums of all 10 solution
        variables taken from the step kernel's registers (savesol! fused; avg on, raw off)
  --workload miz_1024x512x32   the per-GPU share of BASELINE configs[4] (256 members of 1024 x 512 over 8 GPUs:
        32 members per GPU, member forcing by global member index): the ensemble weak-scaling leg
  --workload miz_4096x2048_step   the headline grid with step! semantics: EVERY step writes T0 and the five
        diagnostic fields as well (ebm_step(write_diag = 1), src/miz.jl:150-196 returns all ten variables
        every call): 144 B per cell-step

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per cell-step (MIZ: 96 B =
read + write of Ei, Ew, h, D, phi and the T0 warm start; integrate: + 16 B per saved
variable for the read-modify-write of its running sum) x cells per launch / median launch duration.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

BYTES_PER_CELL_STEP = 96.0       # MIZ, state only
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
PREROLL_S = 0.25

WORKLOADS = {
    # name: (model, grid kind, nlat, ncol, nt)
    "miz_4096x2048": ("MIZ", "sin", 4096, 2048, 1048576),
    "miz_1024x512x32": ("MIZ", "sin", 1024, 512 * 32, 65536),
    "miz_1024x512x32_integrate": ("MIZ", "sin", 1024, 512 * 32, 65536),
    "miz_180x1": ("MIZ", "sin", 180, 1, 2000),               # the reference's own test / docstring shape
    "miz_180x8192": ("MIZ", "sin", 180, 8192, 2000),         # an ensemble of it: 8192 members of the reference's resolution
    "miz_1440x1": ("MIZ", "sin", 1440, 1, 131072),
    "miz_2048x4096": ("MIZ", "sin", 2048, 4096, 262144),     # same cells and bytes as the headline, half-length meridians
    "classic_1024x512": ("Classic", "identity", 1024, 512, 2000),
    # the implicit-diffusion EXTENSION (not in the reference): the headline grid at the reference test's 2000
    # steps per year, 520x beyond the explicit limit of the reference's own step
    "miz_imex_4096x2048": ("MIZ_IMEX", "sin", 4096, 2048, 2000),
    # step! semantics on the headline grid: every step also writes T0 and the five diagnostic fields
    "miz_4096x2048_step": ("MIZ", "sin", 4096, 2048, 1048576),
}
MIZ_VARS = ("E", "T", "h", "Ei", "Ew", "Ti", "Tw", "D", "phi", "n")


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
    nc1 = min(8, ncols)
    one = {k: np.ascontiguousarray(v[:nc1]) for k, v in sub.items()}
    n1 = max(2, int(2.0 * rate / cores / (nc1 * nlat)))
    t0 = time.perf_counter()
    co.miz_run(kid, st.x, dict(par), st.dt, table(n1), np.zeros(n1), None if fc is None else fc[:nc1], one, nthreads=1)
    rate1 = nc1 * nlat * n1 / (time.perf_counter() - t0)
    return {
        "value": ncols * nlat * nsteps / dt, "unit": "grid-cell-steps/s", "cores": cores,
        "value_single_core": rate1,
        "kind": "port",
        "sample": f"{nsteps} steps x {ncols} columns x {nlat} latitudes of the spun-up state "
                  f"(oracle/ebm_oracle.c, gcc -O2 -fopenmp, {dt:.1f} s)",
    }


def cells_bytes(nlat, ncol):
    return 8.0 * nlat * ncol


def spawn_ranks(ngpus: int, argv: list) -> int:
    """`python bench.py --gpus N` (N > 1, no launcher): start N ranks as a CHILD process group and relay
    rank 0's line.  This process must not have touched the GPU (devices are counted from the driver's
    topology, not through HIP): on this pool a process that has initialised HIP may not start a launcher."""
    import subprocess
    pkg = graft.load_package()
    backend = os.environ.get("EBM_BENCH_BACKEND", "nccl")
    ndev = pkg.visible_gpu_count()
    if ndev < 1:
        print("bench.py needs a GPU: the HIP path has no CPU fallback", file=sys.stderr)
        return 2
    if backend == "nccl" and ndev < ngpus:
        print(f"bench.py --gpus {ngpus}: only {ndev} GPU(s) visible on this node — refusing to report an "
              f"{ngpus}-GPU number from fewer devices (RCCL needs one device per rank; EBM_BENCH_BACKEND=gloo "
              f"rehearses the N > 1 path with ranks sharing a device and says so in its line)", file=sys.stderr)
        return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(pkg.free_port()),
           os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)       # stderr passes through
    relayed = 0
    for line in proc.stdout.splitlines():
        is_line = False
        if line.startswith("{"):
            try:
                is_line = "metric" in json.loads(line)
            except ValueError:
                pass
        if is_line and not relayed:
            print(line, flush=True)
            relayed += 1
        elif line.strip():
            print(line, file=sys.stderr)
    if proc.returncode == 0 and relayed != 1:
        print("bench.py: the ranks finished without printing the result line", file=sys.stderr)
        return 3
    return proc.returncode


def kernel_name(model, K, info, ncol=1, ncu=256, fused_state=None):
    """The kernel a workload's launches run, derived the way the library decides (csrc/ebm_kernels.hip: fused_state_in_lds;
    csrc/ebm_runtime.hip: ebm_create_ex)."""
    if not model.startswith("MIZ"):
        return "classic_step_kernel"
    if K <= 1:
        return "miz_step_kernel"
    if info["cells_per_thread"] == 2:
        return "miz_fused_kernel"
    # four cells per thread: state in registers where that kernel exists and the launch has few columns; resident in LDS for
    # longer meridians, for the extension, and for more columns than the register kernel runs in one round (unless told otherwise)
    in_lds = fused_state if fused_state is not None else ncol > ncu * max(1, 256 // info["threads"])
    return "miz_fused_kernel" if (model == "MIZ" and info["threads"] <= 512 and not in_lds) else "miz_resident_kernel"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node to run on (one rank each); default: the launcher's world size, else 1")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; the median is reported")
    ap.add_argument("--spinup", type=int, default=2000)
    ap.add_argument("--preroll", type=float, default=PREROLL_S,
                    help="seconds of untimed steps right before the warm-up (0 under a counter-collecting profiler)")
    ap.add_argument("--workload", default="miz_4096x2048", choices=sorted(WORKLOADS))
    ap.add_argument("--fused-state", choices=("auto", "registers", "lds"), default="auto",
                    help="ebm_options.fused_state_in_lds: where fused-K launches keep the state when both kernels exist")
    ap.add_argument("--integrate-steps-per-launch", type=int, default=None,
                    help="ebm_options.integrate_steps_per_launch of the integrate workload (default: the library's 64; 1 = one "
                         "launch per step)")
    ap.add_argument("--steps-per-launch", type=int, default=1,
                    help="K > 1: fused-K stepping (ebm_run_fused), reported as its own metric")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--launch-chains", type=int, default=1, choices=(1, 2),
                    help="2: ebm_options.launch_chains = 2 — the two halves of the columns stepped by two independent chains of "
                         "launches on two streams (bit-identical; reported as its own metric, never the headline)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if not launched and (args.gpus or 1) > 1:
        # no launcher: become one — before anything in this process touches the GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is None:
        args.gpus = world
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} under a launcher with WORLD_SIZE={world}: the two must agree "
                         "(n_gpus in the result line is the number of ranks that really ran)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; EBM_BENCH_BACKEND=gloo lets several ranks share a device to rehearse the
    # N > 1 path on a one-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("EBM_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} GPU(s) visible — RCCL needs one device per rank; "
                         "refusing to report an N-GPU number from fewer devices")
    device = local_rank % ndev
    torch.cuda.set_device(device)
    dist = None
    # (EBM_BENCH_FORCE_DIST=1: initialise the process group even for ONE rank, so that the RCCL code path — communicator
    #  set-up on the device, barrier, all_reduce of a device tensor, object gather — runs on a one-GPU box as well)
    if world > 1 or (launched and os.environ.get("EBM_BENCH_FORCE_DIST") == "1"):
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    pkg = graft.load_package()
    wl = WORKLOADS[args.workload]
    model, kind, nlat, ncol, nt = wl
    integrate = args.workload.endswith("_integrate")
    every_step_diag = args.workload.endswith("_step")        # step! semantics: all ten variables written every step
    K = max(1, args.steps_per_launch)
    if every_step_diag and K != 1:
        raise SystemExit("--workload miz_4096x2048_step is one ebm_step per step: --steps-per-launch must be 1")
    st = pkg.SpaceTime(kind, nlat, nt, 1)
    par = pkg.default_parameters("MIZ" if model.startswith("MIZ") else model)
    lon = np.arange(ncol) + rank * ncol                      # this rank's block of columns
    if args.workload.startswith("miz_1024x512x32"):
        # BASELINE configs[4] / SURVEY 8(d) cfg5: 32 members of 512 meridians per GPU, member m of 256
        # forced by the constant f_m = -2 + 4 m/255 W/m2
        fcol = -2.0 + 4.0 * ((lon // 512) % 256) / 255.0
    elif ncol == 1:
        fcol = np.zeros(1)                                   # the reference's Forcing(0.0)
    else:
        fcol = 0.5 * np.sin(2.0 * np.pi * lon / ncol)
    chains = args.launch_chains
    # (the reference's own shapes are one meridian: latency-bound, two latitudes per thread — an explicit choice of the
    #  caller, never derived by the library from the column count: energybalancemodel.jl_amd/infrastructure.py)
    cells_opt = 2 if (ncol == 1 and nlat <= 1536 and model == "MIZ" and os.environ.get("EBM_CELLS_PER_THREAD") is None) else None
    eng = pkg.Engine(model, st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval),
                     st.dt, ncol, device=device, cells_per_thread=cells_opt,
                     launch_chains=(chains if chains > 1 else None), use_graph=(False if chains > 1 else None),
                     integrate_steps_per_launch=args.integrate_steps_per_launch,
                     fused_state_in_lds={"auto": None, "registers": False, "lds": True}[args.fused_state])
    if model == "Classic":
        Ts = 30.0 - 45.0 * st.x ** 2
        E0 = np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5)
        eng.set_field("E", np.tile(E0, (ncol, 1)))
        eng.set_field("Tg", np.tile(Ts, (ncol, 1)))
    eng.set_column_forcing(fcol)
    eng.set_time_table(st.t)
    clock = {"step": 0}

    def advance(n):
        """n steps on the stream (asynchronous), the way this workload takes them."""
        if integrate:
            # a `year` of n steps of the same dt: annual-mean sums of all ten variables on every step,
            # the means taken (and copied out) at its end; no raw output, no seasonal snapshots
            i0 = clock["step"] % nt
            eng.set_time_table(np.take(st.t, np.arange(i0, i0 + n) % nt))
            # (the ten mean fields land in the same host arrays every time: allocating 1.3 GB of fresh pages per call in
            #  Python would be timed as part of the "year end")
            res = eng.integrate(n, 1, None, True, 0, 0, MIZ_VARS, want_raw=False, want_seasonal=False, want_avg=True,
                                out=clock.get("out"))
            clock["out"] = res
        elif every_step_diag:
            # Infrastructure.step! once per step (ebm_step with write_diag = 1): the reference's operator returns
            # Tw, Ti, n, E, T with the prognostics on every call (src/miz.jl:150-196)
            eng.set_step_clock(clock["step"])
            tab = eng.ttab
            for i in range(clock["step"], clock["step"] + n):
                eng.step(float(tab[i % nt]), float(tab[(i + 1) % nt]), 0.0, True)
        else:
            eng.run(clock["step"], n, None, False, steps_per_launch=K)
        clock["step"] += n

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()                                           # the handle's own (non-blocking) stream

    eng.run(0, args.spinup, None, False)
    clock["step"] = args.spinup
    eng.sync()
    # pre-roll: a fixed amount of untimed work right before the timed region
    t0 = time.perf_counter()
    probe = max(1, min(args.steps, 64))
    advance(probe)
    eng.sync()
    per_step = (time.perf_counter() - t0) / probe
    preroll = 0 if integrate else int(min(200000, max(0, args.preroll / max(per_step, 1e-7))))
    if preroll > nt:
        # short years (the 180-band workloads: 2000 steps): whole years only, so that every run is timed at the same point of
        # the seasonal cycle — the T0 iteration count, hence the time per step, depends on it (+-3 % otherwise)
        preroll = (preroll // nt) * nt
    if preroll:
        advance(preroll)
    advance(args.warmup) if args.warmup else None
    eng.sync()
    eng.reset_counters()

    blocks_wall, blocks_ev, own_wall = [], [], []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        eng.timer_start()
        advance(args.steps)
        ev_ms = eng.timer_stop()
        barrier()
        elapsed = time.perf_counter() - t0
        own_wall.append(elapsed)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        blocks_wall.append(elapsed)
        blocks_ev.append(ev_ms)
    # every rank's own view of the timed blocks (rank 0 prints them: the MAX above hides who was slow)
    mine = {"rank": rank, "device": device, "blocks_ms_per_step": [b * 1e3 / args.steps for b in own_wall],
            "blocks_event_ms_per_step": [b / args.steps for b in blocks_ev]}
    ranks = [mine]
    world_seen = 1
    if dist is not None:
        world_seen = dist.get_world_size()
        ranks = [None] * world_seen
        dist.all_gather_object(ranks, mine)
    cnt = eng.counters()
    info = eng.launch_info()
    elapsed = statistics.median(blocks_wall)
    ev_ms = statistics.median(blocks_ev)
    year_end_ms = None
    if integrate:
        # Every timed block is one ebm_integrate call that ends a `year`: besides its --steps step launches it
        # allocates and clears the sum buffers, runs ten finish-mean launches and copies ten mean fields to the
        # host.  In the workload's real year (nt = 65,536 steps) that happens once per 65,536 steps; here once per
        # --steps steps.  Separated by a second measurement: blocks of 2 x --steps steps cost one more set of step
        # launches and the same per-call work, so the difference is the steps alone.
        two = []
        for _ in range(3):
            eng.sync()
            t0 = time.perf_counter()
            advance(2 * args.steps)
            eng.sync()
            two.append((time.perf_counter() - t0) * 1e3)
        per_step_ms = max((statistics.median(two) - elapsed * 1e3) / args.steps, 1e-9)
        year_end_ms = max(0.0, elapsed * 1e3 - per_step_ms * args.steps)

    # ---- after the timed region: diagnostics of the state that was timed, CPU baseline --------------
    cpu = None
    ice_fraction = None
    host_transfer = None
    if model.startswith("MIZ"):
        # the fp64 T0 (the CPU baseline's warm start) exists only after a step that writes the diagnostics: the library
        # refuses to hand out a stale one (EBM_ERR_STALE) — one more step, with diagnostics, after the timed region
        eng.run(clock["step"], 1, None, True)
        clock["step"] += 1
        eng.sync()
        eng.get_field("phi")       # untimed: the handle's pinned staging ring is created by its first host transfer
        t0 = time.perf_counter()
        state = {k: eng.get_field(k) for k in ("Ei", "Ew", "h", "D", "phi", "T0")}
        dl = time.perf_counter() - t0
        t0 = time.perf_counter()
        for k in ("Ei", "Ew", "h", "D", "phi"):
            eng.set_field(k, state[k])                       # the same values back: the state is unchanged
        ul = time.perf_counter() - t0
        # what crossing the C ABI with HOST buffers costs (ebm_get_field / ebm_set_field, pageable memory):
        # never part of `value`, which is measured with the state resident in HBM
        host_transfer = {"download_GBps": 6 * cells_bytes(nlat, ncol) / dl / 1e9,
                         "upload_GBps": 5 * cells_bytes(nlat, ncol) / ul / 1e9,
                         "state_round_trip_ms": (dl * 5 / 6 + ul) * 1e3,
                         "note": "6 fields down into fresh (never touched) host arrays, 5 prognostic fields up, through "
                                 "ebm_get_field/ebm_set_field (pinned staging ring + host threads, after one untimed transfer); "
                                 "state_round_trip_ms / ms_per_step = steps a resident state must take per round trip "
                                 "for PCIe to cost as much as the stepping"}
        ice_fraction = float(np.mean(state["phi"] > 0))
        if rank == 0 and world == 1 and args.cpu_budget > 0 and model == "MIZ":
            cpu = cpu_baseline(pkg, wl, st, par, state, fcol, clock["step"], args.cpu_budget)
        del state
    eng.close()

    cells = nlat * ncol
    nsaved = len(MIZ_VARS) if integrate else 0
    bpc = (BYTES_PER_CELL_STEP if model.startswith("MIZ") else 32.0) + 16.0 * nsaved
    if every_step_diag:
        bpc += 48.0                                          # T0 and Tw, Ti, n, E, T written as well (144 B, SURVEY 8(d): 136 + T0)
    spl = (cnt["steps"] / cnt["launches"]) if cnt["launches"] else 1.0
    kname = kernel_name(model, K, info, ncol, torch.cuda.get_device_properties(device).multi_processor_count,
                        {"auto": None, "registers": False, "lds": True}[args.fused_state])
    fused_note = None
    if integrate and spl > 1.5:
        # ebm_integrate fuses the steps that need only the running sums (all but the year's last here): the state stays in
        # LDS between the steps of a launch, the sums' read-modify-write (16 B per saved variable) is the traffic left
        kname = "miz_resident_kernel"
        bpc = BYTES_PER_CELL_STEP / spl + 16.0 * nsaved
        fused_note = ("ebm_integrate, fused: the state stays in LDS between the steps of a launch (its 96 B per cell-step are "
                      "moved once per launch), the annual-mean sums are read and written every step (16 B per saved variable) — "
                      "ALGORITHMIC bytes: a workgroup adds to the same 80 KiB of sums on every step of its launch, so most of "
                      "that read-modify-write is served by L2, and achieved may exceed what HBM itself streams")
    elif spl > 1.0 and kname != "miz_step_kernel":
        # K steps per launch with the state on the chip: HBM is touched once per LAUNCH, so the algorithmic bytes per
        # cell-step are 1/K of the per-step figure; the kernel is bound by one workgroup's VALU issue and barrier chain
        bpc = bpc / spl
        where = "registers" if kname == "miz_fused_kernel" else "LDS"
        fused_note = (f"fused-K: the state stays in {where} between the steps of a launch; achieved/frac count the bytes "
                      "really moved (per-step figure / K) — this kernel is VALU-issue and barrier-latency bound, not HBM bound")
    launches = cnt["launches"] / max(1, args.repeats)        # kernel launches per timed block
    ev_kernel_ms = max(ev_ms - (year_end_ms or 0.0), 1e-9)   # the step launches alone (integrate: without the year end)
    launch_s = ev_kernel_ms * 1e-3 / max(1.0, launches)
    achieved = bpc * cells * args.steps / (ev_kernel_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    pmc_key = args.workload if K == 1 else None
    traffic_source = None
    if pmc_key and os.path.exists(pmc):
        with open(pmc) as fh:
            doc = json.load(fh)
        traffic = doc.get(pmc_key)
        if traffic:
            traffic_source = ("profiles/pmc_latest.json: " + str(doc.get("_detail", {}).get("source", "rocprofv3 --pmc of an earlier run"))
                              + " — a tracked constant, NOT measured in this run")
    name = args.workload + (f", {K} steps per launch (fused)" if K > 1 else "") + (f", {chains} launch chains" if chains > 1 else "")
    if chains > 1:
        # two kernels in flight: a profiler's per-kernel durations overlap pairwise; the roofline figure is per STEP
        # (both chains' launches of a step together), i.e. algorithmic bytes per step / time per step
        launches = launches / chains
        launch_s = ev_kernel_ms * 1e-3 / max(1.0, launches)
        spl = spl * chains
    out = {
        "metric": "grid-cell-steps/sec (2D 4096x2048 MIZ model)" if (args.workload == "miz_4096x2048" and K == 1 and chains == 1)
                  else f"grid-cell-steps/sec ({name})",
        "value": cells * world * args.steps / elapsed,
        "unit": "grid-cell-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "backend": (backend if dist is not None else None), "world_size_seen": world_seen,
        "devices_used": len({r["device"] for r in ranks}),
        "ranks": ranks,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "repeats": len(blocks_wall),
        "blocks_ms_per_step": [b * 1e3 / args.steps for b in blocks_wall],
        "preroll_steps": preroll,
        **({"year_end_ms": year_end_ms,
            "ms_per_step_excluding_year_end": (elapsed * 1e3 - year_end_ms) / args.steps,
            "year_end_note": "each timed block is one ebm_integrate call closing a year (sums cleared, one finish-mean "
                             "launch, 10 mean fields copied to the caller's pageable arrays through the pinned ring); value / ms_per_step INCLUDE that "
                             "once per --steps steps, the workload's own year has 65,536 steps; separated by timing "
                             "blocks of 2 x --steps steps as well (the difference is --steps step launches)"} if integrate else {}),
        "config": {
            "workload": f"{name}: {model} model, {nlat} lat x {ncol} meridians per GPU, "
                        f"{kind} grid, nt={nt}, {args.spinup} spin-up steps from zero state, "
                        + ("f[member]=-2+4*member/255" if args.workload.startswith("miz_1024x512x32")
                           else ("f=0" if ncol == 1 else "f[lon]=0.5*sin(2*pi*lon/nlon)"))
                        + (", every step through ebm_step with write_diag = 1 (T0 + Tw, Ti, n, E, T written too)" if every_step_diag else "")
                        + (", ebm_integrate: annual-mean sums of 10 variables from the step kernel's registers, "
                           "means copied out at the end of every block" if integrate else ""),
            "steps_per_launch": spl if cnt["launches"] else None,
            "ice_covered_fraction": ice_fraction,
            "mean_tridiagonal_solves_per_column_step": (cnt["solves"] / (cnt["steps"] * ncol)) if model.startswith("MIZ") and cnt["steps"] else 1.0,
            "t0_cap_hits": cnt["cap_hits"],
            "threads_per_workgroup": info["threads"], "cells_per_thread": info["cells_per_thread"],
            "lds_bytes_per_workgroup": info["lds_bytes"],
            "parallelism": f"columns sharded over {world} GPU(s), no in-loop collective",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "kernel": kname, **({"note": fused_note} if fused_note else {}),
            **({"launch_chains": chains, "chains_note": "two launches per step, one per half of the columns, on two streams and in "
                "flight together: avg_launch_ms is the time per STEP; a profiler reports twice the launches, each about as long"}
               if chains > 1 else {}),
            "algorithmic_bytes_per_cell_step": bpc,
            "algorithmic_bytes_per_launch": bpc * cells * spl,
            "avg_launch_ms": launch_s * 1e3,
            "blocks_event_ms_per_step": [b / args.steps for b in blocks_ev],
            # transparency: the kernel carries the T0 warm start as a bit mask, so it moves fewer bytes
            # than the contract's 96 B per cell-step; this is the rate of the counter-measured traffic
            "achieved_traffic_gbs": (traffic / launch_s / 1e9) if traffic else None,
            "frac_traffic": (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
        },
        "cpu_baseline": cpu,
        "host_transfer": host_transfer,
    }
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
