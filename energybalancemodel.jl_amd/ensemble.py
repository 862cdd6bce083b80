"""Ensembles and 2-D (lat x lon) grids: many independent meridians per handle, sharded across
GPUs by column.

The reference has no longitude axis and no ensemble mechanism; columns never
exchange data, so the natural shard is the column.  One process per GPU owns a contiguous
block of columns; the only communication is I/O (broadcast of the grid/parameters, gather of
per-column diagnostics) over ``torch.distributed`` (backend "nccl" = RCCL on ROCm, "gloo" on
CPU for tests).  No collective sits inside the time loop.
"""
from __future__ import annotations

import numpy as np

from .engine import Engine, param_vector
from .infrastructure import default_parval


def shard_columns(ncol: int, world_size: int, rank: int) -> slice:
    """Contiguous block partition of ``ncol`` columns: the first ``ncol % world_size`` ranks
    get one extra column."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(ncol, world_size)
    start = rank * base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def hemispheric_mean(vec: np.ndarray, x: np.ndarray) -> np.ndarray:
    """Trapezoid integral over x of each column (reference src/utilities.jl:397-403), summed
    strictly left to right as the reference's loop does (np.cumsum is sequential)."""
    v = np.asarray(vec, dtype=np.float64)
    return np.cumsum((v[..., :-1] + v[..., 1:]) * (x[1:] - x[:-1]) / 2.0, axis=-1)[..., -1]


class EnsembleRun:
    """``ncol`` independent columns of one model on one GPU (this rank's shard).

    ``init`` maps prognostic names to [ncol, nlat] arrays (or [nlat], broadcast to all
    columns); ``fcol`` is the per-column forcing offset; ``forcings`` a sequence of one Forcing
    per column, evaluated on the device at every step (hysteresis ensembles: every member its own
    ramp).

    A column's results do not depend on how many columns share its handle or on how the ensemble is sharded
    over GPUs: the launch geometry is a function of the latitude count and ``cells_per_thread`` only."""

    def __init__(self, model, st, par, init, fcol=None, device=0, forcings=None, cells_per_thread=None):
        first = np.asarray(next(iter(init.values())))
        self.ncol = 1 if first.ndim == 1 else first.shape[0]
        if fcol is not None:
            self.ncol = len(fcol)
        if forcings is not None:
            self.ncol = len(forcings)
        self.st = st
        self.device = int(device)
        # cells_per_thread: launch option of ebm_create_ex (None = the library's default of 4).  Every rank of a
        # sharded run must pass the same value: the rounding of the solves depends on it and on nothing else
        self.engine = Engine(model, st.grid_kind, st.x, param_vector(par, default_parval), st.dt,
                             self.ncol, device, cells_per_thread=cells_per_thread)
        for k, v in init.items():
            a = np.asarray(v, dtype=np.float64)
            if a.ndim == 1:
                a = np.broadcast_to(a, (self.ncol, st.nx))
            self.engine.set_field(k, a)
        if fcol is not None:
            self.engine.set_column_forcing(fcol)
        self.engine.set_time_table(st.t)
        if forcings is not None:
            self.engine.set_column_schedules(forcings)
        self.step_index = 0

    def run(self, nsteps, forcing=None, diag_last=True, steps_per_launch=None):
        """Advance ``nsteps`` steps; ``forcing`` is a Forcing or None.  Nothing leaves the device between the
        steps of this call, so they are fused ``steps_per_launch`` to a launch (ebm_run_fused: the state stays in
        registers or LDS, bit-identical to one launch per step for every model and size); default 64, 1 = one
        launch per step (ebm_run)."""
        if steps_per_launch is None:
            steps_per_launch = 64
        f = None
        if forcing is not None:
            T = (np.arange(self.step_index, self.step_index + nsteps) + 0.5) * self.st.dt
            f = np.array([forcing(float(t)) for t in T])
        self.engine.run(self.step_index, nsteps, f, diag_last, steps_per_launch)
        self.step_index += nsteps

    def seasonal_means(self, years, names=("T", "phi"), forcing=None):
        """Integrate ``years`` whole years from the current state and return, per column, the
        hemispheric means (reference src/utilities.jl:397-403) of the winter snapshot, the summer
        snapshot and the annual mean of every variable in ``names`` for every year — reduced on the
        device (ebm_integrate_hemispheric): dict(winter, summer, avg), each [len(names), years, ncol].
        This is the data behind the reference's hysteresis plot (src/plot.jl:173-225: hemispheric_mean
        of seasonal.avg.T[year] against 2*pi*hemispheric_mean of seasonal.{avg,winter,summer}.phi[year])
        for every member, at O(members x years) bytes of I/O.  Model time — the scalar ``forcing`` and the
        per-column Forcing schedules — continues from the steps this run has already taken, so a ramp integrated
        in several calls (chunks of years) equals the same ramp integrated in one; the calls must start at a
        year boundary, as the reference's ``integrate`` does."""
        st = self.st
        if self.step_index % st.nt:
            raise ValueError(f"seasonal_means starts a year: {self.step_index} steps taken so far is not a multiple of nt = {st.nt}")
        f = None
        if forcing is not None:
            T = (np.arange(self.step_index, self.step_index + st.nt * years) + 0.5) * st.dt
            f = np.array([forcing(float(t)) for t in T])
        self.engine.set_step_clock(self.step_index)
        out = self.engine.integrate_hemispheric(st.nt, years, f, st.winter.inx, st.summer.inx, tuple(names))
        self.step_index += st.nt * years
        return out

    def state(self, names=None):
        return self.engine.get_state(names)

    def hemispheric_mean_tensor(self, name):
        """Per-column hemispheric mean (reference src/utilities.jl:397-403) as a torch tensor ON THE
        DEVICE — reduced there by ebm_hemispheric_mean_device, never staged through the host — ready
        to be gathered over RCCL (gather_columns)."""
        import torch
        out = torch.empty(self.ncol, dtype=torch.float64, device=torch.device("cuda", self.device))
        self.engine.hemispheric_mean_device(name, out.data_ptr())
        return out

    def field_tensor(self, name):
        """A packed [ncol, nlat] device tensor copy of a field (device-to-device)."""
        import torch
        out = torch.empty((self.ncol, self.st.nx), dtype=torch.float64, device=torch.device("cuda", self.device))
        self.engine.get_field_device(name, out.data_ptr())
        return out

    def close(self):
        self.engine.close()


def broadcast_inputs(arrays: dict | None, dist=None, device=None, src: int = 0) -> dict:
    """Broadcast the run's small inputs — ``st.x``, the parameter vector, the per-step forcing
    table, the per-column forcing offsets — from rank ``src`` to every rank (I/O only, a few KB;
    SURVEY §8(e)).  ``arrays`` maps names to fp64 arrays on ``src`` and is ignored elsewhere;
    every rank returns the same dict.  With dist=None (single process) returns ``arrays``."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in (arrays or {}).items()}
    import torch
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        arrays = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in arrays.items()}
        meta = [[(k, v.shape) for k, v in arrays.items()]]
    dist.broadcast_object_list(meta, src=src)                 # names and shapes (host side)
    total = sum(int(np.prod(shape)) for _, shape in meta[0])
    flat = np.concatenate([arrays[k].ravel() for k, _ in meta[0]]) if rank == src else np.empty(total)
    t = torch.from_numpy(flat)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)                                # one message for all payloads
    flat = t.cpu().numpy()
    out, pos = {}, 0
    for k, shape in meta[0]:
        n = int(np.prod(shape))
        out[k] = flat[pos:pos + n].reshape(shape).copy()
        pos += n
    return out


def _backend(dist) -> str:
    try:
        return str(dist.get_backend())
    except Exception:
        return ""


def gather_columns(local, ncol_total: int, dist=None, device=None, dst: int = 0):
    """Gather per-column data ([ncol_local, ...]) from all ranks to rank ``dst`` (I/O only).

    ``local`` is a NumPy array or a torch tensor.  With the "nccl" backend (= RCCL on ROCm) the
    payload stays on the device end to end: a device tensor (e.g. from
    ``EnsembleRun.hemispheric_mean_tensor`` / ``field_tensor``) is gathered with ``dist.gather`` —
    point-to-point sends to ``dst`` only, 1/world_size of an all-gather's traffic — and copied to the
    host once, on ``dst``.  With "gloo" (CPU tests) host tensors are gathered.  Returns the
    [ncol_total, ...] NumPy array on ``dst`` and None elsewhere.  With dist=None (single process)
    returns ``local`` as a NumPy array."""
    import_torch = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    if not import_torch:
        return local.cpu().numpy() if hasattr(local, "cpu") else local
    import torch
    ws, rank = dist.get_world_size(), dist.get_rank()
    t = local if isinstance(local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
    if _backend(dist) == "nccl":
        if t.device.type == "cpu":
            t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    else:
        t = t.cpu()
    width = -(-ncol_total // ws)                      # block sizes differ by at most one column: pad
    if t.shape[0] < width:
        t = torch.cat([t, torch.zeros((width - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)])
    t = t.contiguous()
    outs = [torch.empty_like(t) for _ in range(ws)] if rank == dst else None
    dist.gather(t, outs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(ws):
        sl = shard_columns(ncol_total, ws, r)
        parts.append(outs[r][: sl.stop - sl.start])
    return torch.cat(parts, dim=0).cpu().numpy()
