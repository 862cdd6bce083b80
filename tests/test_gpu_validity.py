"""GPU tests (-m gpu) of round 3's boundary work, all through the C ABI:

* validity of the fields that only diagnostic steps write (EBM_ERR_STALE): nothing stale is returned silently,
  in particular not the T0 a caller would checkpoint as the warm start (reference src/miz.jl:47,64);
* launch options are arguments (ebm_create_ex / struct ebm_options), never the environment, and the launch
  geometry — hence the rounding of the tridiagonal solves — does not depend on the number of columns;
* the diagnostic fields' private store layout is invisible: every reader sees [ncol][nlat];
* host transfers through the pinned ring return exactly what a plain copy returns.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PROG = ("Ei", "Ew", "h", "D", "phi")
DIAG = ("Tw", "Ti", "n", "E", "T")
ALL = PROG + ("T0",) + DIAG


def make_engine(pkg, model, st, par, ncol=1, **kw):
    return pkg.Engine(model, st.grid_kind, st.x, pkg.engine.param_vector(par, pkg.default_parval),
                      st.dt, ncol, device=0, **kw)


def test_stale_fields_are_refused_not_returned(pkg):
    """run(diag_last=False) + get_field("T") fails with EBM_ERR_STALE, as do T0, the other diagnostics, the device
    copies, the zero-copy pointer and the hemispheric mean; the prognostic fields are always readable; a diagnostic
    step makes everything current again; overwriting a prognostic field makes the diagnostics stale; the explicit
    "as of step" query returns the old field when — and only when — the named step wrote it."""
    import torch
    st = pkg.SpaceTime("sin", 180, 2000, 1)
    par = pkg.default_parameters("MIZ")
    with make_engine(pkg, "MIZ", st, par, 3) as eng:
        eng.set_time_table(st.t)
        with pytest.raises(pkg.StaleFieldError, match="never been written"):
            eng.get_field("T")                                     # before any step there is no T (src/miz.jl:187)
        assert np.array_equal(eng.get_field("T0"), np.zeros((3, 180)))      # the warm start begins at zero (src/miz.jl:47)
        eng.run(0, 30, None, True)
        fs = eng.field_step("T")
        assert fs == dict(written_step=29, state_step=29, current=True)
        T29, T0_29 = eng.get_field("T"), eng.get_field("T0")
        eng.run(30, 5, None, False)
        for name in DIAG + ("T0",):
            with pytest.raises(pkg.StaleFieldError, match=r"last written by step 29; the state is at step 34"):
                eng.get_field(name)
        assert eng.field_step("Ti") == dict(written_step=29, state_step=34, current=False)
        assert eng.field_step("Ei") == dict(written_step=34, state_step=34, current=True)
        buf = torch.empty((3, 180), dtype=torch.float64, device="cuda")
        with pytest.raises(pkg.StaleFieldError):
            eng.get_field_device("T", buf.data_ptr())
        with pytest.raises(pkg.StaleFieldError):
            eng.hemispheric_mean("T")
        with pytest.raises(pkg.StaleFieldError):
            eng.field_device_ptr("n")
        with pytest.raises(pkg.StaleFieldError):
            eng.get_state()                                        # the default set includes the diagnostics
        assert set(eng.get_state(PROG)) == set(PROG)               # prognostics: always current
        # the explicit query: the field as of the step that wrote it
        assert np.array_equal(eng.get_field_as_of("T", 29), T29, equal_nan=True)
        assert np.array_equal(eng.get_field_as_of("T0", 29), T0_29)
        with pytest.raises(pkg.StaleFieldError, match="not as of step 34"):
            eng.get_field_as_of("T", 34)
        assert eng.get_field_as_of("Ei", 34).shape == (3, 180)
        # a diagnostic step: current again
        eng.step(float(eng.ttab[35]), 0.0, 0.0, True)
        assert eng.field_step("T")["current"] and eng.field_step("T")["written_step"] == 35
        eng.get_state()
        # fused run without diagnostics: stale again; with: current
        eng.run(36, 20, None, False, steps_per_launch=8)
        with pytest.raises(pkg.StaleFieldError):
            eng.get_field("E")
        eng.run(56, 20, None, True, steps_per_launch=8)
        eng.get_field("E")
        # overwriting a prognostic field: the diagnostics no longer describe the state
        eng.set_field("h", eng.get_field("h"))
        with pytest.raises(pkg.StaleFieldError, match="prognostic fields overwritten"):
            eng.get_field("T")
        # ... but a field the caller sets is the caller's statement of what it holds
        eng.set_field("T0", T0_29)
        assert np.array_equal(eng.get_field("T0"), T0_29)
    # classic: T and h are the diagnostics
    stc = pkg.SpaceTime("identity", 180, 2000, 1)
    parc = pkg.default_parameters("Classic")
    with make_engine(pkg, "Classic", stc, parc, 2) as eng:
        eng.set_time_table(stc.t)
        eng.set_field("E", np.full((2, 180), 50.0))
        eng.run(0, 4, None, False)
        with pytest.raises(pkg.StaleFieldError):
            eng.get_field("h")
        eng.get_field("Tg")
        eng.run(4, 1, None, True)
        assert np.isfinite(eng.get_field("T")).all()


@pytest.mark.parametrize("nlat,nt", [(180, 2000), (1024, 65536)])
def test_checkpoint_and_resume_through_t0(pkg, nlat, nt):
    """Checkpoint = the five prognostics + T0 read after a diagnostic step (src/miz.jl:47,64: the reference's hidden
    warm start, here part of the handle's state); a fresh handle that is given them continues bit for bit like the
    uninterrupted run.  Without T0 the resumed run starts its active-set iteration from the wrong set; with a STALE T0
    the library refuses to hand it out at all."""
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    n1, n2 = 60, 40
    fcol = np.array([0.0, 1.5])
    with make_engine(pkg, "MIZ", st, par, 2) as eng:
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(0, n1, None, True)
        ckpt = eng.get_state(PROG + ("T0",))
        eng.run(n1, n2, None, True)
        want = eng.get_state(ALL)
        cnt_want = eng.counters()["solves"]
    with make_engine(pkg, "MIZ", st, par, 2) as eng:
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.set_state(ckpt)
        eng.run(n1, n2, None, True)
        got = eng.get_state(ALL)
        cnt_got = eng.counters()["solves"]
    for k in ALL:
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    assert np.any(want["phi"] > 0)
    with make_engine(pkg, "MIZ", st, par, 2) as eng:               # the first run's solve count, for the comparison below
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(0, n1, None, False)
        first = eng.counters()["solves"]
    assert cnt_got == cnt_want - first                             # the same iterations, step for step


def test_options_are_arguments_not_environment(pkg, monkeypatch):
    """The library reads no environment variable: with every former knob set in the environment a handle made by
    plain ebm_create has the default geometry; ebm_create_ex with options has what the options say; bad options are
    refused.  (The Python mirror maps the variables to options on purpose — not used here.)"""
    import sys
    L = sys.modules[pkg.__name__ + "._lib"]
    lib = L.load()
    monkeypatch.setenv("EBM_CELLS_PER_THREAD", "2")
    monkeypatch.setenv("EBM_GRAPH", "0")
    monkeypatch.setenv("EBM_PREFETCH_COLS", "7")
    st = pkg.SpaceTime("sin", 180, 2000, 1)
    pv = pkg.engine.param_vector(pkg.default_parameters("MIZ"), pkg.default_parval)
    x = np.ascontiguousarray(st.x)

    def create(opt):
        h = C.c_void_p()
        if opt is None:
            rc = lib.ebm_create(C.byref(h), 0, 1, 180, 1, L.dptr(x), L.dptr(pv), st.dt, 0)
        else:
            rc = lib.ebm_create_ex(C.byref(h), 0, 1, 180, 1, L.dptr(x), L.dptr(pv), st.dt, 0, C.byref(opt))
        return rc, h

    def info(h):
        out = (C.c_int * 4)()
        assert lib.ebm_launch_info(h, out) == 0
        return list(out)

    rc, h = create(None)
    assert rc == 0 and info(h)[:2] == [64, 4]                     # 180 cells, four per thread: one wave
    lib.ebm_destroy(h)
    opt = L.Options()
    assert lib.ebm_options_default(C.byref(opt)) == 0
    assert (opt.struct_bytes, opt.cells_per_thread, opt.use_graph, opt.prefetch_cols, opt.launch_chains,
            opt.integrate_steps_per_launch, opt.fused_state_in_lds) == (C.sizeof(L.Options), 0, -1, -1, -1, -1, -1)
    rc, h = create(opt)
    assert rc == 0 and info(h)[:2] == [64, 4]
    lib.ebm_destroy(h)
    opt.cells_per_thread = 2
    rc, h = create(opt)
    assert rc == 0 and info(h)[:2] == [128, 2]                    # two waves
    lib.ebm_destroy(h)
    # a caller compiled against an OLDER, shorter struct: the fields it does not have keep their defaults
    short = L.Options()
    short.struct_bytes, short.cells_per_thread, short.use_graph, short.prefetch_cols, short.launch_chains = 8, 2, 99, -99, 7
    rc, h = create(short)
    assert rc == 0 and info(h)[:2] == [128, 2]
    lib.ebm_destroy(h)
    for field, bad in (("cells_per_thread", 3), ("use_graph", 2), ("prefetch_cols", -2), ("struct_bytes", 0), ("launch_chains", 3),
                       ("integrate_steps_per_launch", -2), ("fused_state_in_lds", 2)):
        o = L.Options()
        lib.ebm_options_default(C.byref(o))
        setattr(o, field, bad)
        rc, h = create(o)
        assert rc == -1 and not h.value, field


def test_results_do_not_depend_on_the_column_count(pkg):
    """The launch geometry — and with it the partition of the tridiagonal solves, i.e. their rounding — is a function of
    the latitude count and the cells_per_thread option only.  A member therefore gives the SAME BITS alone, inside a
    large ensemble, and under any sharding: checked across the column counts at which earlier versions switched the
    geometry by themselves (ncol x ceil(nlat/256) <= 128), for both explicit choices."""
    st = pkg.SpaceTime("sin", 180, 2000, 1)
    par = pkg.default_parameters("MIZ")
    ncol = 200
    fcol = np.linspace(-2.0, 2.0, ncol)
    nsteps = 80

    def run(cols, cells):
        with make_engine(pkg, "MIZ", st, par, len(cols), cells_per_thread=cells) as eng:
            assert eng.launch_info()["cells_per_thread"] == (cells or 4)
            eng.set_column_forcing(fcol[cols])
            eng.set_time_table(st.t)
            eng.run(0, nsteps, None, True)
            return eng.get_state(ALL)

    for cells in (None, 2, 4):
        whole = run(np.arange(ncol), cells)
        assert np.any(whole["phi"] > 0)
        halves = [run(np.arange(0, 100), cells), run(np.arange(100, 200), cells)]           # 2 x 100: below the old threshold
        alone = run(np.array([137]), cells)                                                   # one member on its own
        for k in ALL:
            assert np.array_equal(np.concatenate([halves[0][k], halves[1][k]]), whole[k], equal_nan=True), (cells, k)
            assert np.array_equal(alone[k][0], whole[k][137], equal_nan=True), (cells, k)


@pytest.mark.parametrize("model,kind,nlat,ncol,nt", [("MIZ", "sin", 4096, 37, 1048576), ("MIZ", "sin", 300, 9, 8000), ("MIZ_IMEX", "sin", 1024, 6, 2000),
                                                      ("Classic", "identity", 512, 5, 2000)])
def test_two_launch_chains_give_the_same_bits(pkg, model, kind, nlat, ncol, nt):
    """ebm_options.launch_chains = 2: the two halves of the columns are stepped by two independent chains of launches on two
    streams.  Columns are independent, so every result — runs, fused runs, a diagnostic step in between, integrate with
    savesol! in the step, the hemispheric means, fields set and read in between (which join the chains) — equals the
    one-chain run bit for bit; the launch counter counts one launch per chain and step."""
    st = pkg.SpaceTime(kind, nlat, nt, 1)
    par = pkg.default_parameters("Classic" if model == "Classic" else "MIZ")
    fcol = np.linspace(-2.0, 2.0, ncol)
    names = ("E", "Tg", "T", "h") if model == "Classic" else ALL
    out = {}
    for chains in (1, 2):
        with make_engine(pkg, model, st, par, ncol, launch_chains=chains, use_graph=False) as eng:
            if model == "Classic":
                Ts = 30.0 - 45.0 * st.x ** 2
                eng.set_field("E", np.tile(np.where(Ts >= 0, par["cw"] * Ts, par["Lf"] * Ts / 7.5), (ncol, 1)))
                eng.set_field("Tg", np.tile(Ts, (ncol, 1)))
            eng.set_column_forcing(fcol)
            eng.set_time_table(st.t)
            eng.run(0, 30, None, True)
            mid = eng.get_state(names)                                # joins the chains
            eng.set_field(names[0], mid[names[0]])                    # an upload between two forked runs
            eng.run(30, 21, None, False, steps_per_launch=(1 if model == "MIZ_IMEX" else 5))
            eng.step(float(eng.ttab[51]), float(eng.ttab[52]), 0.25, True)
            hm = eng.hemispheric_mean("T")
            state = eng.get_state(names)
            cnt = eng.counters()
            eng.set_time_table(st.t[:16])
            res = eng.integrate(16, 2, None, False, 4, 11, names[:3] if model == "Classic" else ("E", "T", "phi", "Ti"))
            out[chains] = (mid, state, hm, res, cnt)
    for a, b in zip(out[1][:2], out[2][:2]):
        for k in names:
            assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.array_equal(out[1][2], out[2][2], equal_nan=True)
    for k in ("raw", "winter", "summer", "avg"):
        assert np.array_equal(out[1][3][k], out[2][3][k], equal_nan=True), k
    assert out[1][4]["steps"] == out[2][4]["steps"] and out[2][4]["launches"] == 2 * out[1][4]["launches"]
    assert out[1][4]["solves"] == out[2][4]["solves"]


def test_two_launch_chains_wait_for_what_the_handle_did_before(pkg):
    """The second chain runs on a stream of its own; before its first launch after anything else the handle did, it has to
    wait for that work.  The case with teeth: the year end of ebm_integrate on a grid large enough that its finish-mean launch
    (which reads, averages and ZEROES the running sums of every saved variable) is still running when the next year's first
    step is launched — a second chain that did not wait would add into sums that are then zeroed under it.  Annual means of
    three 8-step years, one chain against two, bit for bit."""
    nlat, ncol, nt, years = 1024, 4096, 65536, 3
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    fcol = np.linspace(-2.0, 2.0, ncol)
    out = {}
    # (with one launch per step everywhere, and with the default: the stretches between the year ends fused.  Up to a late
    #  commit of round 3 every fused stretch began with a stream synchronisation, which hid a missing wait from this test:
    #  profiles/r03_mutants_final.log)
    for chains, spl in ((1, 1), (2, 1), (2, None)):
        with make_engine(pkg, "MIZ", st, par, ncol, launch_chains=chains, use_graph=False, integrate_steps_per_launch=spl) as eng:
            eng.set_column_forcing(fcol)
            eng.set_time_table(st.t)
            eng.run(0, 40, None, False)
            eng.set_time_table(st.t[:8])
            res = eng.integrate(8, years, None, True, 0, 0, ("E", "T", "phi", "Ew", "h", "Ei"), want_raw=False, want_seasonal=False)
            out[(chains, spl)] = res["avg"]
    assert np.array_equal(out[(1, 1)], out[(2, 1)], equal_nan=True)
    assert np.array_equal(out[(1, 1)], out[(2, None)], equal_nan=True)
    assert np.isfinite(out[(1, 1)][2]).all() and np.any(out[(1, 1)][2] > 0)


@pytest.mark.parametrize("nlat,ncol,nt", [(180, 5, 2000), (1000, 3, 60000), (4096, 4, 1048576)])
def test_diagnostic_fields_read_back_in_the_natural_layout(pkg, nlat, ncol, nt):
    """The step kernels store the diagnostic fields in a layout of their own (whole lines per store instruction); every
    reader must see [ncol][nlat]: host copy, device copy, zero-copy pointer, hemispheric mean — equal to the fused
    kernel's (natural-layout) result bit for bit, also after repeated diagnostic steps, after setting one diagnostic
    field by hand while the others are still in the private layout, and for the seasonal snapshots of ebm_integrate."""
    import torch
    hip = C.CDLL("libamdhip64.so")
    st = pkg.SpaceTime("sin", nlat, nt, 1)
    par = pkg.default_parameters("MIZ")
    fcol = np.linspace(-1.0, 1.0, ncol)
    with make_engine(pkg, "MIZ", st, par, ncol, cells_per_thread=4) as eng:
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(0, 24, None, True)                                  # diagnostics stored, never read ...
        eng.step(float(eng.ttab[24]), 0.0, 0.0, True)               # ... and stored again
        got = eng.get_state(ALL)
        for k in DIAG:
            dev = torch.empty((ncol, nlat), dtype=torch.float64, device="cuda")
            eng.get_field_device(k, dev.data_ptr())
            assert np.array_equal(dev.cpu().numpy(), got[k], equal_nan=True), k
            ptr, pitch = eng.field_device_ptr(k)
            view = np.empty((ncol, pitch))
            assert hip.hipMemcpy(view.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(view.nbytes), 2) == 0
            assert np.array_equal(view[:, :nlat], got[k], equal_nan=True), k
            assert not view[:, nlat:].any()                        # padding cells stay zero
        hm = eng.hemispheric_mean("E")
        assert np.array_equal(hm, pkg.hemispheric_mean(got["E"], st.x))
        # one more diagnostic step, then set ONE diagnostic field by hand: the others keep their values
        eng.step(float(eng.ttab[25]), 0.0, 0.0, True)
        marker = np.arange(ncol * nlat, dtype=np.float64).reshape(ncol, nlat)
        eng.set_field("n", marker)
        assert np.array_equal(eng.get_field("n"), marker)
        again = eng.get_state(DIAG)
    with make_engine(pkg, "MIZ", st, par, ncol, cells_per_thread=4) as eng:
        eng.set_column_forcing(fcol)
        eng.set_time_table(st.t)
        eng.run(0, 26, None, True, steps_per_launch=1)
        want = eng.get_state(ALL)
    for k in DIAG:
        if k != "n":
            assert np.array_equal(again[k], want[k], equal_nan=True), k
    # the diagnostics of the one-launch-per-step kernel == those of the fused kernel (natural-layout stores)
    if nlat <= 2048:
        with make_engine(pkg, "MIZ", st, par, ncol, cells_per_thread=4) as eng:
            eng.set_column_forcing(fcol)
            eng.set_time_table(st.t)
            eng.run(0, 26, None, True, steps_per_launch=13)
            fused = eng.get_state(ALL)
        for k in ALL:
            assert np.array_equal(fused[k], want[k], equal_nan=True), k


@pytest.mark.parametrize("nlat,ncol", [(180, 1), (181, 7), (1024, 4100), (4096, 1100), (2, 3)])
def test_host_transfers_through_the_pinned_ring(pkg, nlat, ncol):
    """ebm_set_field / ebm_get_field move fields through a pinned staging ring in pieces of 16 MiB, host threads on one
    side and the DMA engine on the other: whatever the shape (rows shorter than the pitch, fields of several pieces,
    a piece boundary inside the field), what comes back is what went in, and padding cells stay zero."""
    rng = np.random.default_rng(nlat * 1000 + ncol)
    st = pkg.SpaceTime("sin", nlat, 2000, 1)
    par = pkg.default_parameters("MIZ")
    with make_engine(pkg, "MIZ", st, par, ncol) as eng:
        sent = {}
        for k in PROG + ("T0", "E"):
            sent[k] = rng.standard_normal((ncol, nlat))
            eng.set_field(k, sent[k])
        for k in sent:
            assert np.array_equal(eng.get_field(k), sent[k]), k
        ptr, pitch = eng.field_device_ptr("Ei")
        assert pitch >= nlat
        view = np.empty((ncol, pitch))
        assert C.CDLL("libamdhip64.so").hipMemcpy(view.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(view.nbytes), 2) == 0
        assert np.array_equal(view[:, :nlat], sent["Ei"]) and not view[:, nlat:].any()      # padding cells stay zero


def test_chunked_seasonal_means_equal_one_call(pkg):
    """Model time continues across calls: a hysteresis ramp (per-member Forcing schedules + a scalar Forcing) integrated
    as 2 + 1 + 3 years of seasonal_means equals the same six years in one call, bit for bit."""
    st = pkg.SpaceTime("sin", 90, 500, 1)
    par = pkg.default_parameters("MIZ")
    members = [pkg.Forcing(0.0), pkg.Forcing(0.0, 4.0, 1.0, (1, 1), (2.0, -1.0)), pkg.Forcing(-1.0, 2.0, 0.0, (0, 2), (1.0, -2.0))]
    scalar = pkg.Forcing(0.0, 1.0, 0.0, (2, 1), (1.0, -1.0))
    init = {k: np.zeros(90) for k in PROG}
    run = pkg.EnsembleRun("MIZ", st, par, init, forcings=members)
    one = run.seasonal_means(6, ("T", "phi"), forcing=scalar)
    run.close()
    run = pkg.EnsembleRun("MIZ", st, par, init, forcings=members)
    parts = [run.seasonal_means(n, ("T", "phi"), forcing=scalar) for n in (2, 1, 3)]
    with pytest.raises(ValueError, match="starts a year"):
        run.run(7)
        run.seasonal_means(1)
    run.close()
    for k in ("winter", "summer", "avg"):
        got = np.concatenate([p[k] for p in parts], axis=1)
        assert np.array_equal(got, one[k], equal_nan=True), k
    assert np.ptp(one["avg"][0, :, 1]) > 0.1                       # the ramps really did something
