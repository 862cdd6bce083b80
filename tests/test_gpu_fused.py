"""GPU tests (-m gpu) of the round-2 launch paths, all through the C ABI:

* fused-K stepping (ebm_run_fused): K steps per launch, state in registers between steps —
  bit-identical to one launch per step (reference loop: src/infrastructure.jl:630-634);
* savesol! fused into the step kernel (ebm_integrate): one launch per step, the annual-mean sums
  and raw snapshots taken from the step's registers — the mean must equal, bit for bit, the
  sequential per-cell sum of the raw snapshots divided by nt (src/infrastructure.jl:549-591,
  src/utilities.jl:390-395);
* the device-pointer entry points.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import record_error, scaled_err

pytestmark = pytest.mark.gpu

PROG = ("Ei", "Ew", "h", "D", "phi")
DIAG = ("Tw", "Ti", "n", "E", "T")
ALL = PROG + ("T0",) + DIAG
MIZ_VARS = ("E", "T", "h", "Ei", "Ew", "Ti", "Tw", "D", "phi", "n")


def make_engine(pkg, model, st, par, ncol=1):
    return pkg.Engine(model, st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval),
                      st.dt, ncol, device=0)


def classic_init(pkg, st, par, ncol):
    Ts = 30.0 - 45.0 * st.x ** 2
    E0 = np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5)
    return dict(E=np.tile(E0, (ncol, 1)), Tg=np.tile(Ts, (ncol, 1)))


# ---- fused-K stepping ---------------------------------------------------------------------------
@pytest.mark.parametrize("kind,nlat,ncol,nt,K", [
    ("sin", 180, 1, 2000, 64),            # BASELINE configs[0] shape: one wave
    ("sin", 180, 3, 2000, 7),
    ("identity", 180, 2, 2000, 50),
    ("sin", 63, 2, 2000, 16),             # ragged single wave
    ("sin", 1440, 1, 131072, 32),         # BASELINE configs[1]: six waves (twelve with two cells per thread)
    ("identity", 1101, 2, 100000, 11),    # two cells per thread: 768 threads for 551 chunks
    ("sin", 1536, 2, 150000, 6),          # the longest meridian with two cells per thread
    ("identity", 1000, 2, 60000, 5),
    ("sin", 2048, 3, 262144, 9),          # the largest meridian whose state the register kernel holds (512 threads)
    ("sin", 2045, 2, 262144, 4),
    ("sin", 2049, 2, 262144, 5),          # from here on the state is resident in LDS (miz_resident_kernel): 576 threads
    ("identity", 2560, 3, 400000, 7),     # 640 threads
    ("sin", 3300, 2, 700000, 4),          # 832 threads, ragged
    ("identity", 3840, 2, 900000, 6),     # 960 threads
    ("sin", 4096, 3, 1048576, 8),         # BASELINE's largest meridian: 1024 threads, every byte of the CU's LDS
    ("identity", 4093, 2, 1048576, 3),
])
def test_fused_run_equals_single_steps(pkg, kind, nlat, ncol, nt, K, cells):
    """K steps per launch against one launch per step: same operations on the same values, so every
    field — prognostics, the T0 of the last step, the diagnostics, NaN sentinels — is bitwise equal.
    Varying per-step forcing, per-column offsets, a run length that is not a multiple of K, a start
    late in the year (time index wraps), state handed over between two fused calls."""
    if cells == 2 and nlat > 1536:
        pytest.skip("two cells per thread exist up to 1536-cell meridians")
    st = pkg.SpaceTime(kind, nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    nsteps = 3 * K + 5
    first = nt - 2 * K                                         # wraps around the year end
    f_steps = 0.3 * np.sin(np.arange(nsteps) / 5.0)
    fcol = np.linspace(-1.5, 1.5, ncol) if ncol > 1 else np.array([0.4])
    out, cnt = {}, {}
    for mode in ("single", "fused"):
        with make_engine(pkg, "MIZ", st, par, ncol) as eng:
            eng.set_column_forcing(fcol)
            eng.set_time_table(st.t)
            eng.run(0, 40, None, False)                        # freeze-up: ice edge and T0 solve live
            eng.reset_counters()
            if mode == "single":
                eng.run(first, nsteps, f_steps, True)
            else:
                half = K + 3
                eng.run(first, half, f_steps[:half], False, steps_per_launch=K)
                eng.run(first + half, nsteps - half, f_steps[half:], True, steps_per_launch=K)
            out[mode] = eng.get_state(ALL)
            cnt[mode] = eng.counters()
    for k in ALL:
        assert np.array_equal(out["single"][k], out["fused"][k], equal_nan=True), k
    assert np.any(out["single"]["phi"] > 0) and np.any(out["single"]["Ew"] != 0)
    assert cnt["fused"]["steps"] == cnt["single"]["steps"] == nsteps
    assert cnt["fused"]["solves"] == cnt["single"]["solves"] and cnt["fused"]["cap_hits"] == 0
    assert cnt["single"]["launches"] == nsteps
    assert cnt["fused"]["launches"] == -(-(K + 3) // K) + -(-(nsteps - K - 3) // K)


def test_fused_run_with_column_schedules(pkg):
    """Per-column Forcing schedules are evaluated at the model time of every fused step."""
    st = pkg.SpaceTime("sin", 180, 2000, 3)
    par = pkg.default_parameters("MIZ")
    members = [pkg.Forcing(0.5), pkg.Forcing(0.0, 2.0, 0.0, (1, 0), (2.0, -2.0)),
               pkg.Forcing(-1.0, 1.0, 0.0, (0, 1), (2.0, -1.0))]
    out = {}
    for K in (1, 37):
        with make_engine(pkg, "MIZ", st, par, 3) as eng:
            eng.set_time_table(st.t)
            eng.set_column_schedules(members)
            eng.run(1900, 2300, None, True, steps_per_launch=K)
            out[K] = eng.get_state(ALL)
    for k in ALL:
        assert np.array_equal(out[1][k], out[37][k], equal_nan=True), k
    assert np.any(out[1]["Ew"][0] != out[1]["Ew"][1])


@pytest.mark.parametrize("nlat,ncol,K", [(180, 2, 25), (1024, 3, 8), (333, 1, 100)])
def test_fused_run_classic(pkg, nlat, ncol, K, cells):
    """Classic model: E and Tg stay in registers across the K steps of a launch."""
    st = pkg.SpaceTime("identity", nlat, 2000, 1)
    par = pkg.default_parameters("Classic")
    nsteps = 2 * K + 3
    f_steps = 0.5 * np.cos(np.arange(nsteps) / 3.0)
    out = {}
    for mode in (1, K):
        with make_engine(pkg, "Classic", st, par, ncol) as eng:
            eng.set_state(classic_init(pkg, st, par, ncol))
            eng.set_column_forcing(np.linspace(-1.0, 1.0, ncol))
            eng.set_time_table(st.t)
            eng.run(1995, nsteps, f_steps, True, steps_per_launch=mode)
            out[mode] = eng.get_state(("E", "Tg", "T", "h"))
            assert eng.counters()["launches"] == (nsteps if mode == 1 else 3)
    for k in out[1]:
        assert np.array_equal(out[1][k], out[K][k], equal_nan=True), k


def test_fused_run_long_meridians_keep_their_state_in_lds(pkg):
    """Meridians of more than 2048 cells do not fit the register file: their fused-K kernel keeps the state in LDS
    (160 KiB at 4096 cells).  One launch per K steps — not one per step, as up to round 3 — and the same bits,
    also over a run long enough for the ice edge to move (active-set changes inside a launch) and with the two
    launch chains of ``launch_chains = 2``."""
    st = pkg.SpaceTime("sin", 4096, 1048576, 1)
    par = pkg.default_parameters("MIZ")
    out, cnt = {}, {}
    for K, chains in ((1, 1), (16, 1), (64, 2)):
        with pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, 5, device=0,
                        launch_chains=chains, use_graph=False) as eng:
            eng.set_column_forcing(np.linspace(-2.0, 2.0, 5))
            eng.set_time_table(st.t)
            eng.run(0, 200, None, True, steps_per_launch=K)
            out[K] = eng.get_state(ALL)
            cnt[K] = eng.counters()
    assert cnt[1]["launches"] == 200 and cnt[16]["launches"] == 13 and cnt[64]["launches"] == 2 * 4
    assert cnt[1]["solves"] > 200 * 5                          # some steps took more than one iteration
    for K in (16, 64):
        assert cnt[K]["solves"] == cnt[1]["solves"] and cnt[K]["cap_hits"] == 0
        for k in ALL:
            assert np.array_equal(out[1][k], out[K][k], equal_nan=True), (K, k)
    assert np.any(out[1]["phi"] > 0)


def test_fused_kernel_choice_follows_the_column_count_and_changes_no_bit(pkg):
    """Where both fused-K kernels exist the library picks by column count (more columns than the register kernel runs in one
    round: state in LDS, for occupancy; fewer: in registers, for latency) unless ebm_options.fused_state_in_lds says
    otherwise.  A launch of many short meridians: per step, default, forced either way — the same bits."""
    import torch
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    nlat, ncol, nt = 180, 4 * ncu + 7, 2000
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    out = {}
    for name, kw, K in (("single", {}, 1), ("auto", {}, 16), ("registers", dict(fused_state_in_lds=False), 16), ("lds", dict(fused_state_in_lds=True), 16)):
        with pkg.Engine("MIZ", st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), st.dt, ncol, device=0, **kw) as eng:
            eng.set_column_forcing(np.linspace(-3.0, 3.0, ncol))
            eng.set_time_table(st.t)
            eng.run(0, 100, None, True, steps_per_launch=K)
            out[name] = eng.get_state(ALL)
    for name in ("auto", "registers", "lds"):
        for k in ALL:
            assert np.array_equal(out["single"][k], out[name][k], equal_nan=True), (name, k)


def test_ensemble_run_fuses_by_default(pkg):
    """EnsembleRun.run lets nothing leave the device between its steps, so it fuses them (64 to a launch) unless told
    otherwise — same bits as one launch per step, a Forcing evaluated per step, for a short and a long meridian and
    for the extension."""
    for model, nlat, nt in (("MIZ", 180, 2000), ("MIZ", 3000, 600000), ("MIZ_IMEX", 1440, 2000)):
        st = pkg.SpaceTime("sin", nlat, nt, 1)
        par = pkg.default_parameters("MIZ")
        init = {k: np.zeros((3, nlat)) for k in PROG}
        forcing = pkg.Forcing(0.0, 2.0, 0.0, (0, 0), (2.0, -2.0)) if nt == 2000 else None
        out, launches = {}, {}
        for spl in (None, 1):
            run = pkg.EnsembleRun(model, st, par, init, fcol=np.array([-1.0, 0.0, 1.0]))
            run.run(150, forcing, steps_per_launch=spl)
            out[spl] = run.engine.get_state(ALL)
            launches[spl] = run.engine.counters()["launches"]
            run.close()
        assert launches[1] == 150 and launches[None] == 3, launches
        for k in ALL:
            assert np.array_equal(out[None][k], out[1][k], equal_nan=True), (model, nlat, k)


def test_fused_run_matches_oracle(pkg, coracle):
    """The fused path against the oracle directly (not only against the per-step path)."""
    nlat, nt, nsteps = 1440, 131072, 60
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    ct = np.array([pkg.cos2pit(float(t)) for t in st.t[:nsteps]])
    state = {k: np.zeros((1, nlat)) for k in PROG + ("T0",)}
    diag, ocnt = coracle.miz_run(1, st.x, dict(par), st.dt, ct, np.zeros(nsteps), None, state)
    with make_engine(pkg, "MIZ", st, par, 1) as eng:
        eng.set_time_table(st.t)
        eng.run(0, nsteps, None, True, steps_per_launch=20)
        got = eng.get_state(ALL)
        cnt = eng.counters()
    ref = dict(state, **diag)
    for k in ALL:
        e = scaled_err(got[k], ref[k])
        record_error("fused 1440 x1, 60 steps from zero", k, e, 1e-10)
        assert e <= 1e-10, k
    assert cnt["solves"] == ocnt[0] and cnt["launches"] == 3


# ---- savesol! fused into the step kernel ----------------------------------------------------------
def sequential_mean(raw, nt):
    """crossmean as the device accumulates it: per cell, sum over the year's steps in step order
    (one IEEE addition per step, starting from +0.0), then one division by nt."""
    acc = np.zeros(raw.shape[:1] + raw.shape[2:])
    for ti in range(raw.shape[1]):
        acc = acc + raw[:, ti]
    return acc / float(nt)


@pytest.mark.parametrize("model,kind,nlat,ncol,nt,dur", [
    ("MIZ", "sin", 180, 2, 400, 2),
    ("MIZ", "identity", 255, 3, 600, 1),          # ragged: padding cells of sums and snapshots
    ("MIZ", "sin", 1024, 2, 70000, 1),            # 256 threads; only the first 300 steps of the year are run
    ("MIZ", "sin", 1024, 48, 70000, 1),           # 3.9 MB per snapshot: the 256 MiB staging ring is flushed 5 times
    ("Classic", "identity", 180, 2, 400, 2),
    ("Classic", "identity", 333, 1, 500, 1),
])
def test_integrate_saves_from_registers(pkg, model, kind, nlat, ncol, nt, dur, cells):
    """ebm_integrate takes the annual-mean sums and the raw snapshots from the step kernel's
    registers (one launch per step).  (1) avg == sequential sum of the raw snapshots / nt, bit for
    bit; (2) the last raw snapshot is the final state; (3) winter / summer snapshots are the raw
    snapshots of those steps; (4) launches == steps + nothing else per step."""
    short = nt > 5000
    if short:
        nt_run = 300                                           # a 300-step "year" of the same dt
        st = pkg.SpaceTime(kind, nlat, nt, 1)
        st_run = pkg.SpaceTime(kind, nlat, nt_run, 1)
        st_run.dt = st.dt
    else:
        st_run = pkg.SpaceTime(kind, nlat, nt, dur)
        nt_run = nt
    par = pkg.default_parameters(model)
    names = MIZ_VARS if model == "MIZ" else ("E", "T", "h", "Tg")
    total = nt_run * dur
    f_steps = 0.4 * np.sin(np.arange(total) / 11.0)
    with make_engine(pkg, model, st_run, par, ncol) as eng:
        if model == "Classic":
            eng.set_state(classic_init(pkg, st_run, par, ncol))
        eng.set_column_forcing(np.linspace(-1.0, 2.0, ncol))
        eng.set_time_table(st_run.t)
        eng.reset_counters()
        out = eng.integrate(nt_run, dur, f_steps, False, st_run.winter.inx, st_run.summer.inx, names)
        final = eng.get_state(names)
        cnt = eng.counters()
    raw = out["raw"]
    assert raw.shape == (len(names), total, ncol, nlat)
    for y in range(dur):
        ref = sequential_mean(raw[:, y * nt_run:(y + 1) * nt_run], nt_run)
        assert np.array_equal(out["avg"][:, y], ref, equal_nan=True), f"year {y}"
        assert np.array_equal(out["winter"][:, y], raw[:, y * nt_run + st_run.winter.inx - 1], equal_nan=True)
        assert np.array_equal(out["summer"][:, y], raw[:, y * nt_run + st_run.summer.inx - 1], equal_nan=True)
    for vi, v in enumerate(names):
        assert np.array_equal(raw[vi, -1], final[v], equal_nan=True), v
    assert cnt["launches"] == cnt["steps"] == total
    if model == "MIZ":
        assert np.isnan(out["avg"][names.index("Ti")]).any()   # NaN sentinels propagate into the mean


@pytest.mark.parametrize("model,kind,nlat,ncol,nt,dur,K", [
    ("MIZ", "sin", 180, 3, 400, 3, 64),
    ("MIZ", "identity", 180, 2, 300, 2, 7),
    ("MIZ", "sin", 1024, 5, 250, 2, 64),
    ("MIZ_IMEX", "sin", 1440, 2, 200, 3, 16),
    ("MIZ", "sin", 2049, 2, 120, 2, 64),
    ("MIZ", "identity", 4096, 3, 100, 2, 13),
    ("MIZ_IMEX", "identity", 4096, 2, 80, 2, 64),
    ("MIZ", "sin", 1300, 1, 90, 2, 64),        # two cells per thread at 768 threads has no fused variant: one launch per step
])
def test_integrate_fuses_the_stretches_between_snapshots(pkg, model, kind, nlat, ncol, nt, dur, K, cells):
    """ebm_integrate steps through what needs nothing but the annual-mean sums (lastonly: every year but the last, between
    the seasonal snapshots) K steps to a launch with the state resident on the chip and the sums taken inside the launch
    (miz_resident_kernel<SAVE>; two cells per thread: miz_fused_kernel<2, ..., SAVE>).  Against
    integrate_steps_per_launch = 1 (one launch per step everywhere): every output —
    raw of the last year, winter, summer, the annual means of every year — and the final state bitwise equal, fewer
    launches; also with no mean asked for (plain fused stepping) and with hemispheric means reduced on the device."""
    if cells == 2 and nlat > 1536:
        pytest.skip("two cells per thread exist up to 1536-cell meridians")
    st = pkg.SpaceTime(kind, nlat, nt, dur)
    par = pkg.default_parameters("MIZ")
    dt = st.dt if model == "MIZ_IMEX" else 1.0 / max(float(nt), 0.7 * nlat * nlat)
    tt = (np.arange(nt) + 0.5) * dt
    f_steps = 0.4 * np.sin(np.arange(nt * dur) / 9.0)
    fcol = np.linspace(-1.0, 1.0, ncol)
    out, launches, state = {}, {}, {}
    for spl in (1, K):
        with pkg.Engine(model, st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval), dt, ncol, device=0,
                        integrate_steps_per_launch=spl) as eng:
            eng.set_column_forcing(fcol)
            eng.set_time_table(np.array([pkg.cos2pit(float(t)) for t in tt]))
            eng.run(0, 30, None, False)
            eng.reset_counters()
            out[spl] = [eng.integrate(nt, dur, f_steps, True, 3, nt // 2 + 1, MIZ_VARS)]
            launches[spl] = eng.counters()["launches"]
            state[spl] = eng.get_state(ALL)
            out[spl].append(eng.integrate(nt, dur, f_steps, True, 3, nt // 2 + 1, ("T", "phi"), want_avg=False))
            out[spl].append(eng.integrate_hemispheric(nt, dur, f_steps, 3, nt // 2 + 1, ("T", "phi")))
    assert launches[1] == nt * dur
    segs = [2, nt // 2 + 1 - 3 - 1, nt - (nt // 2 + 1) - 1]        # plain steps before winter, between the seasons, before year end
    per_year = sum(-(-n // K) if n >= 2 else n for n in segs) + 3
    if cells == 2 and model == "MIZ" and nlat > 1024:
        per_year = nt                                           # 768 threads, two cells each: not fused
    assert launches[K] == (dur - 1) * per_year + nt, (launches, per_year)
    for a, b in zip(out[1], out[K]):
        assert a.keys() == b.keys()
        for k in a:
            assert (a[k] is None and b[k] is None) or np.array_equal(a[k], b[k], equal_nan=True), (model, nlat, k)
    for k in ALL:
        assert np.array_equal(state[1][k], state[K][k], equal_nan=True), k
    assert np.ptp(out[1][0]["avg"][MIZ_VARS.index("T")][:, 0, 0]) > 0            # the years differ


def test_integrate_avg_only_and_lastonly(pkg):
    """avg without seasonal snapshots or raw output (Engine.integrate(want_seasonal=False)): the sums
    run on every step, nothing else is stored; lastonly keeps the last year's raw snapshots only.
    All three ways of asking give the same means, bit for bit."""
    st = pkg.SpaceTime("sin", 180, 500, 3)
    par = pkg.default_parameters("MIZ")
    res = {}
    for mode in ("avg_only", "lastonly", "full"):
        with make_engine(pkg, "MIZ", st, par, 2) as eng:
            eng.set_column_forcing(np.array([0.0, 1.0]))
            eng.set_time_table(st.t)
            res[mode] = eng.integrate(st.nt, st.dur, None, mode != "full", st.winter.inx, st.summer.inx, MIZ_VARS,
                                      want_raw=(mode != "avg_only"), want_seasonal=(mode != "avg_only"))
    assert res["avg_only"]["raw"] is None and res["avg_only"]["winter"] is None
    assert res["avg_only"]["avg"] is not None and not np.all(np.isnan(res["avg_only"]["avg"]))
    assert res["lastonly"]["raw"].shape[1] == st.nt and res["full"]["raw"].shape[1] == st.nt * st.dur
    for mode in ("avg_only", "lastonly"):
        assert np.array_equal(res[mode]["avg"], res["full"]["avg"], equal_nan=True), mode
    assert np.array_equal(res["lastonly"]["raw"], res["full"]["raw"][:, -st.nt:], equal_nan=True)


def test_integrate_rejects_the_hidden_warm_start(pkg):
    st = pkg.SpaceTime("sin", 64, 100, 1)
    with make_engine(pkg, "MIZ", st, pkg.default_parameters("MIZ")) as eng:
        eng.set_time_table(st.t)
        with pytest.raises(pkg.EBMError, match="not a solution variable"):
            eng.integrate(st.nt, 1, None, True, st.winter.inx, st.summer.inx, ("E", "T0"))
        with pytest.raises(pkg.EBMError, match="listed twice"):
            eng.integrate(st.nt, 1, None, True, st.winter.inx, st.summer.inx, ("E", "E"))


# ---- device-pointer entry points ----------------------------------------------------------------
def test_device_pointer_entry_points(pkg):
    """ebm_field_device_ptr (zero-copy view), ebm_get_field_device (packed device copy) and
    ebm_hemispheric_mean_device against the host-copy entry points, through torch tensors on the
    handle's device (no host staging)."""
    import torch
    nlat, ncol = 255, 5                                       # pitch 256 > nlat: the view is strided
    st = pkg.SpaceTime("sin", nlat, 8000, 1)
    par = pkg.default_parameters("MIZ")
    run = pkg.EnsembleRun("MIZ", st, par, {k: np.zeros(nlat) for k in PROG},
                          fcol=np.linspace(-2.0, 2.0, ncol), device=0)
    run.run(30)
    eng = run.engine
    host = eng.get_field("T")
    ptr, pitch = eng.field_device_ptr("T")
    assert ptr and pitch == 256
    # copy the strided device view [ncol][pitch] out through HIP and compare
    hip = C.CDLL("libamdhip64.so")
    buf = np.empty((ncol, pitch))
    assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(buf.nbytes), 2) == 0
    assert np.array_equal(buf[:, :nlat], host, equal_nan=True) and np.all(buf[:, nlat:] == 0.0)
    packed = run.field_tensor("T")
    assert packed.is_cuda and packed.shape == (ncol, nlat)
    assert np.array_equal(packed.cpu().numpy(), host, equal_nan=True)
    for name in ("T", "phi", "Ti"):
        dev = run.hemispheric_mean_tensor(name)
        assert dev.is_cuda
        assert np.array_equal(dev.cpu().numpy(), eng.hemispheric_mean(name), equal_nan=True), name
    with pytest.raises(pkg.EBMError, match="not part of this model"):
        eng.field_device_ptr("Tg")
    # single-process gather is the identity, tensor or array
    assert np.array_equal(pkg.gather_columns(run.hemispheric_mean_tensor("T"), ncol), eng.hemispheric_mean("T"))
    run.close()


# ---- the diffusion operator on its own ---------------------------------------------
@pytest.mark.parametrize("kind", ["identity", "sin"])
@pytest.mark.parametrize("nlat", [2, 63, 180, 1000, 4096])
def test_diffusion_operator_is_bit_exact(pkg, oracle, kind, nlat):
    """ebm_diffusion = diffusion!(base, temp, st, par) / diffusion(T, st, par), src/infrastructure.jl:495-533,
    evaluated by the device functions the step kernels fuse: bit for bit the oracle's restatement of the
    reference's operation order (CSC product on the identity grid, flux form elsewhere), zero-flux ends,
    NaN / Inf cells included, with and without a base."""
    rng = np.random.default_rng(nlat)
    st = pkg.SpaceTime(kind, nlat, 2000, 1)
    par = pkg.default_parameters("MIZ")
    ncol = 3
    temp = rng.normal(0.0, 15.0, (ncol, nlat))
    temp[1] = 30.0 - 45.0 * st.x ** 2                          # a smooth profile
    if nlat > 10:
        temp[2, 5] = np.nan
        temp[2, nlat - 2] = np.inf
    base = rng.normal(0.0, 3.0, (ncol, nlat))
    geom = oracle.DiffusionGeometry(kind if kind == "identity" else "sin", st.x, par["D"])
    with make_engine(pkg, "MIZ", st, par, ncol) as eng:
        got0 = eng.diffusion(temp)
        got1 = eng.diffusion(temp, base)
    with np.errstate(all="ignore"):
        for c in range(ncol):
            assert np.array_equal(got0[c], geom.add(np.zeros(nlat), temp[c]), equal_nan=True), (c, "no base")
            assert np.array_equal(got1[c], geom.add(base[c], temp[c]), equal_nan=True), (c, "base")
    # a constant has no gradient: exactly zero in the flux form, zero to rounding in the matrix form
    with make_engine(pkg, "MIZ", st, par, 1) as eng:
        const = eng.diffusion(np.full((1, nlat), 7.25))
        assert np.all(const == 0.0) if kind == "sin" else np.all(np.abs(const) <= 1e-9 * 7.25 * par["D"] * nlat ** 2)
    st2 = pkg.SpaceTime(kind, nlat, 2000, 1)
    with make_engine(pkg, "Classic", st2, pkg.default_parameters("Classic"), 1) as eng:
        with pytest.raises(pkg.EBMError, match="MIZ handle"):
            eng.diffusion(np.zeros((1, nlat)))


# ---- hemispheric means of the seasonal outputs, reduced on the device -----------
@pytest.mark.parametrize("model,kind,nlat,ncol,nt,dur", [
    ("MIZ", "sin", 180, 5, 2000, 2),              # the reference test's grid and time step
    ("MIZ", "identity", 127, 2, 2000, 2),         # ragged
    ("Classic", "identity", 180, 3, 400, 2),
])
def test_integrate_hemispheric_equals_host_means_of_the_fields(pkg, model, kind, nlat, ncol, nt, dur):
    """ebm_integrate_hemispheric returns, per variable, year and column, hemispheric_mean
    (src/utilities.jl:397-403) of the winter snapshot, the summer snapshot and the annual mean — what
    the reference's plot_seasonal consumes (src/plot.jl:173-225) — reduced on the device.  It must equal,
    bit for bit, the reference's sequential loop applied on the host to ebm_integrate's full fields,
    NaN sentinels included, and leave the same final state."""
    st = pkg.SpaceTime(kind, nlat, nt, dur)
    par = pkg.default_parameters(model)
    names = ("T", "phi", "E", "Ti") if model == "MIZ" else ("T", "E", "h")
    f_steps = 0.8 * np.sin(np.arange(nt * dur) / 53.0)
    fcol = np.linspace(-1.5, 1.5, ncol)
    res = {}
    for mode in ("fields", "means"):
        with make_engine(pkg, model, st, par, ncol) as eng:
            if model == "Classic":
                eng.set_state(classic_init(pkg, st, par, ncol))
            eng.set_column_forcing(fcol)
            eng.set_time_table(st.t)
            if mode == "fields":
                out = eng.integrate(nt, dur, f_steps, True, st.winter.inx, st.summer.inx, names, want_raw=False)
            else:
                out = eng.integrate_hemispheric(nt, dur, f_steps, st.winter.inx, st.summer.inx, names)
            res[mode] = (out, eng.get_state(names))
    for k in ("winter", "summer", "avg"):
        fields, means = res["fields"][0][k], res["means"][0][k]
        assert means.shape == (len(names), dur, ncol)
        with np.errstate(invalid="ignore"):
            ref = pkg.hemispheric_mean(fields, st.x)             # [nvars, dur, ncol], sequential left-to-right sum
        assert np.array_equal(means, ref, equal_nan=True), k
    for v in names:
        assert np.array_equal(res["fields"][1][v], res["means"][1][v], equal_nan=True), v
    if model == "MIZ":
        assert np.isnan(res["means"][0]["avg"][names.index("Ti")]).any()      # sentinels reach the means
        phi_w = res["means"][0]["winter"][names.index("phi")]
        assert np.isfinite(phi_w).all() and len(np.unique(phi_w[0])) == ncol     # a stable run; members differ
