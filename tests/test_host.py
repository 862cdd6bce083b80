"""CPU tests of the host side: the mirror of the reference's operator surface, the column
sharding logic, and that the C-ABI library loads and exports every symbol that
include/ebm_hip.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "ebm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ebm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(pkg.EXPORTS), declared ^ set(pkg.EXPORTS)
    assert os.path.exists(pkg.LIB_PATH), "libebm_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(pkg.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.ebm_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ebm_version()


def test_no_gpu_fails_loudly(pkg):
    """The product has no CPU path: without a GPU ebm_create must fail, never compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    st = pkg.SpaceTime("sin", 180, 2000, 1)
    init = pkg.Collection({k: np.zeros(180) for k in ("Ei", "Ew", "h", "D", "phi")})
    with pytest.raises(pkg.EBMError, match="no HIP device"):
        pkg.integrate("MIZ", st, pkg.Forcing(0.0), pkg.default_parameters("MIZ"), init)
    with pytest.raises(pkg.EBMError, match="no HIP device"):
        pkg.step_("MIZ", 0.00025, 0.0, init, st, pkg.default_parameters("MIZ"))


def test_create_rejects_bad_arguments_before_touching_the_device(pkg):
    """Argument checks come first, so they are observable without a GPU: empty / degenerate
    grids, non-positive dt, unknown model or grid tags, null pointers."""
    from energybalancemodel_jl_amd import _lib
    import ctypes as C
    lib = _lib.load()
    x = np.linspace(0.01, 0.99, 16)
    par = np.zeros(len(_lib.PARAM_ORDER))
    h = C.c_void_p()

    def create(model=0, grid=1, nlat=16, ncol=1, dt=1e-3, xp=x, pp=par):
        return lib.ebm_create(C.byref(h), model, grid, nlat, ncol, _lib.dptr(xp), _lib.dptr(pp), dt, 0)

    for kw, msg in ((dict(nlat=1), b"nlat >= 2"), (dict(nlat=0), b"nlat >= 2"), (dict(ncol=0), b"ncol >= 1"),
                    (dict(dt=0.0), b"dt must be positive"), (dict(dt=-1.0), b"dt must be positive"),
                    (dict(model=7), b"unknown model"), (dict(grid=5), b"unknown grid"),
                    (dict(xp=None), b"null argument")):
        assert create(**kw) == -1, kw                       # EBM_ERR_ARG
        assert msg in lib.ebm_last_error(), (kw, lib.ebm_last_error())
        assert not h.value
    assert lib.ebm_destroy(None) == 0                       # destroying a null handle is a no-op
    for fn in (lib.ebm_sync, lib.ebm_timer_start, lib.ebm_reset_counters):
        assert fn(None) == -1


def test_product_does_not_import_oracle():
    """The package must not reference oracle/ (the oracle is test infrastructure)."""
    pdir = os.path.join(ROOT, "energybalancemodel.jl_amd")
    for dirpath, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "ebm_oracle" not in src and "c_oracle" not in src, f
                if f.endswith(".py"):
                    assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_spacetime_matches_oracle_grid(pkg, oracle):
    for kind in ("sin", "identity"):
        a, b = pkg.SpaceTime(kind, 180, 2000, 3), oracle.SpaceTime(kind, 180, 2000, 3)
        assert np.array_equal(a.x, b.x) and np.array_equal(a.t, b.t) and np.array_equal(a.T, b.T)
        assert (a.winter.inx, a.summer.inx, a.dt) == (b.winter_inx, b.summer_inx, b.dt)
    assert pkg.SpaceTime("sin", 180, 2000, 1).grid_kind == "nonuniform"
    assert pkg.SpaceTime("identity", 10, 20, 1).grid_kind == "identity"
    st = pkg.SpaceTime(lambda u: u * u, 50, 100, 1, urange=(0.0, 1.0))     # custom F
    assert st.grid_kind == "nonuniform" and st.x[0] == pytest.approx(1e-4)
    assert repr(pkg.SpaceTime("sin", 180, 2000, 30)) == "SpaceTime{sin}(180, 2000, 30)"


def test_forcing_and_parameters(pkg, oracle):
    f = pkg.Forcing(0.0, 5.0, -5.0, (10, 10), (0.5, -0.5))
    assert f.domain == (0, 10, 20, 30, 50) and f(17.57) == pytest.approx(3.785, abs=1e-12)
    g = oracle.Forcing(0.0, 5.0, -5.0, (10, 10), (0.5, -0.5))
    for T in np.linspace(0, 60, 241):
        assert f(float(T)) == g(float(T))
    with pytest.raises(ValueError, match="Warming time"):
        pkg.Forcing(0.0, 5.0, -5.0, (10, 10), (0.3, -0.5))
    with pytest.raises(ValueError, match="Cooling time"):
        pkg.Forcing(0.0, 5.0, -5.0, (10, 10), (0.5, 0.3))
    assert len(pkg.default_parameters("MIZ")) == 22 and len(pkg.default_parameters("Classic")) == 16
    assert set(pkg.default_parameters("anything else")) == set(pkg.classic_paramset)
    p = pkg.default_parameters("MIZ")
    assert p.Dmax == 156.0 and p.m1 == 1.6e-6 * 31536000
    p.F = 1.5
    assert p["F"] == 1.5 and "F" in p.propertynames()
    with pytest.raises(KeyError):
        _ = p.nonexistent
    assert dict(pkg.default_parval) == oracle.default_parval


def test_param_vector_order(pkg):
    from energybalancemodel_jl_amd import _lib, engine
    hdr = open(os.path.join(ROOT, "include", "ebm_hip.h")).read()
    enum = re.search(r"enum ebm_param \{(.*?)\}", hdr, flags=re.S).group(1)
    names = [n.replace("EBM_P_", "") for n in re.findall(r"EBM_P_[A-Za-z0-9]+", enum)]
    assert tuple(names[:-1]) == _lib.PARAM_ORDER and names[-1] == "COUNT"
    fld = re.search(r"enum ebm_field \{(.*?)\}", hdr, flags=re.S).group(1)
    fnames = [n.replace("EBM_F_", "") for n in re.findall(r"EBM_F_[A-Za-z0-9]+", re.sub(r"/\*.*?\*/", "", fld, flags=re.S))]
    assert fnames[:-1] == sorted(_lib.FIELD, key=_lib.FIELD.get)
    v = engine.param_vector(pkg.default_parameters("MIZ"), pkg.default_parval)
    assert v[_lib.PARAM_ORDER.index("F")] == 0.0 and v[_lib.PARAM_ORDER.index("Lf")] == 9.5


def test_classic_time_index(pkg, oracle):
    st = pkg.SpaceTime("identity", 10, 2000, 1)
    for m in (1, 2, 999, 1000, 1999, 2000):
        assert pkg.classic_time_index(float(st.t[m - 1]), st.dt, st.nt) == m
        assert oracle.classic_time_index(float(st.t[m - 1]), st.dt, st.nt) == m


def test_model_symbol_and_debug_errors(pkg):
    st = pkg.SpaceTime("identity", 10, 20, 1)
    with pytest.raises(ValueError, match="no step! method"):
        pkg.integrate("classic", st, pkg.Forcing(0.0), pkg.default_parameters("Classic"), {})
    with pytest.raises(NotImplementedError):
        pkg.step_("MIZ", 0.1, 0.0, {}, st, pkg.default_parameters("MIZ"), debug="vars.Ei")


def test_shard_columns(pkg):
    for ncol, ws in ((256, 8), (10, 3), (5, 8), (2048, 2)):
        got = []
        for r in range(ws):
            s = pkg.shard_columns(ncol, ws, r)
            got.extend(range(s.start, s.stop))
        assert got == list(range(ncol))
        sizes = [pkg.shard_columns(ncol, ws, r).stop - pkg.shard_columns(ncol, ws, r).start for r in range(ws)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pkg.shard_columns(10, 2, 2)


def test_hemispheric_mean(pkg):
    x = np.linspace(0, 1, 101)
    assert pkg.hemispheric_mean(2 * x, x) == pytest.approx(1.0, rel=1e-12)   # src/utilities.jl:397-403


def test_broadcast_inputs_single_process(pkg):
    """Without a process group the I/O broadcast is the identity (fp64, contiguous copies)."""
    x = np.linspace(0.0, 1.0, 7)
    got = pkg.broadcast_inputs(dict(x=x[::2], f=[1, 2, 3]))
    assert np.array_equal(got["x"], x[::2]) and got["x"].flags["C_CONTIGUOUS"]
    assert got["f"].dtype == np.float64


def test_schedule_words_follow_the_forcing_domain(pkg):
    """The 9 words handed to ebm_set_column_schedule are the Forcing's own fields
    (src/infrastructure.jl:208-241): evaluating them as the kernel does reproduces the call
    operator (:294-307) at every point, including the breakpoints themselves."""
    words = pkg.engine.schedule_words

    def device_eval(w, T):
        base, peak, cool, up, down, d1, d2, d3, d4 = w
        if T < d1:
            return base
        if T < d2:
            return base + up * (T - d1)
        if T < d3:
            return peak
        if T < d4:
            return peak + down * (T - d3)
        return cool

    for f in (pkg.Forcing(0.0, 5.0, -5.0, (10, 10), (0.5, -0.5)), pkg.Forcing(1.0, 3.0, 0.0, (0, 0), (2.0, -1.5)),
              pkg.Forcing(0.75)):
        w = words(f)
        assert len(w) == 9
        for T in list(np.linspace(0.0, 60.0, 481)) + [float(d) for d in f.domain]:
            assert device_eval(w, T) == f(T), (repr(f), T)
    assert words(pkg.Forcing(0.0, 5.0, -5.0, (10, 10), (0.5, -0.5)))[5:] == [10.0, 20.0, 30.0, 50.0]


def test_header_is_valid_c99(tmp_path):
    """include/ebm_hip.h is a C header (no C++, no torch types): it must compile as strict C99, and a
    plain-C caller (examples/c_abi_example.c) must compile against it."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    src = tmp_path / "t.c"
    src.write_text('#include "ebm_hip.h"\nint main(void) { return (int)sizeof(ebm_handle_t) * 0 + EBM_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, "-c",
                           os.path.join(ROOT, "examples", "c_abi_example.c"), "-o", str(tmp_path / "ex.o")])


# ---- the reviewed-only Julia shim, machine-checked against the header ------------------------------
_C2J = {  # C parameter type (normalised) -> the Julia ccall argument types that bind it
    "ebm_handle_t*": {"Ref{Ptr{Cvoid}}"}, "ebm_handle_t": {"Ptr{Cvoid}"}, "int": {"Cint"}, "double": {"Cdouble"},
    "const double*": {"Ptr{Cdouble}"}, "double*": {"Ptr{Cdouble}"}, "const int*": {"Ptr{Cint}"},
    "int*": {"Ptr{Cint}"}, "long long*": {"Ptr{Clonglong}"}, "long long": {"Clonglong"},
    "float*": {"Ptr{Cfloat}", "Ref{Cfloat}"}, "double**": {"Ptr{Ptr{Cdouble}}", "Ref{Ptr{Cdouble}}"},
    "const ebm_options*": {"Ref{EBMOptions}", "Ptr{EBMOptions}"}, "ebm_options*": {"Ref{EBMOptions}", "Ptr{EBMOptions}"},
}


def _header_prototypes():
    hdr = open(os.path.join(ROOT, "include", "ebm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"\b(int|const char \*)\s*(ebm_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", hdr):
        params = []
        for a in (p.strip() for p in args.split(",")):
            if a in ("void", ""):
                continue
            a = re.sub(r"\s+", " ", a)
            m = re.match(r"(.*?)(\**)\s*([A-Za-z_][A-Za-z0-9_]*)$", a)          # type, stars, name
            params.append((m.group(1).strip() + m.group(2)).replace(" *", "*"))
        protos[name] = (ret.replace(" ", ""), params)
    return protos


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _julia_ccalls():
    src = open(os.path.join(ROOT, "julia", "EBMHip.jl")).read()
    src = re.sub(r"#=.*?=#", "", src, flags=re.S)
    src = "\n".join(line.split("#")[0] if not line.lstrip().startswith("#") else "" for line in src.splitlines())
    calls = []
    for m in re.finditer(r"ccall\(\(:(ebm_[a-z0-9_]+), libebm\),", src):
        i = m.end()
        depth, j = 1, i                                    # find the matching ')' of ccall(
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        parts = _split_top(src[i:j - 1])
        ret, types = parts[0], _split_top(parts[1].strip()[1:-1])
        calls.append((m.group(1), ret, [t for t in types if t], parts[2:]))
    return calls


def test_julia_shim_ccalls_match_the_header():
    """julia/EBMHip.jl cannot be executed here (no Julia), but its ccalls can be read: every one must name
    a function the header declares, with the header's return type, the header's number of parameters, a
    Julia type that binds each C parameter type, and exactly as many arguments as parameter types."""
    protos = _header_prototypes()
    assert "ebm_integrate" in protos and len(protos["ebm_integrate"][1]) == 13
    calls = _julia_ccalls()
    assert {c[0] for c in calls} >= {"ebm_create_ex", "ebm_destroy", "ebm_set_field", "ebm_get_field", "ebm_step",
                                    "ebm_set_time_table", "ebm_integrate", "ebm_get_counters", "ebm_last_error",
                                    "ebm_set_column_schedule"}
    for name, ret, types, args in calls:
        assert name in protos, f"{name} is not declared in include/ebm_hip.h"
        cret, cparams = protos[name]
        assert ret == ("Cstring" if cret == "constchar*" else "Cint"), (name, ret)
        assert len(types) == len(cparams), f"{name}: {len(types)} Julia types for {len(cparams)} C parameters"
        assert len(args) == len(types), f"{name}: {len(args)} arguments for {len(types)} parameter types"
        for jt, ct in zip(types, cparams):
            assert jt in _C2J[ct], f"{name}: Julia {jt} does not bind C {ct}"


def test_julia_shim_options_struct_matches_the_header(pkg):
    """struct ebm_options in the header, EBMOptions in the shim and Options in the ctypes binding: the same
    fields, all int, in the same order."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ebm_hip.h")).read(), flags=re.S)
    body = re.search(r"typedef struct ebm_options \{(.*?)\} ebm_options;", hdr, flags=re.S).group(1)
    cfields = re.findall(r"\bint\s+([a-z_]+)\s*;", body)
    assert cfields and len(cfields) == len([l for l in body.split(";") if l.strip()])
    jl = open(os.path.join(ROOT, "julia", "EBMHip.jl")).read()
    jbody = re.search(r"struct EBMOptions\n(.*?)\nend", jl, flags=re.S).group(1)
    jfields = re.findall(r"^\s*([a-z_]+)::Cint", jbody, flags=re.M)
    import sys
    lib = sys.modules[pkg.__name__ + "._lib"]
    pfields = [n for n, t in lib.Options._fields_]
    assert cfields == jfields == pfields
    assert all(t is ctypes.c_int for _, t in lib.Options._fields_)


def test_julia_shim_enums_match_the_header():
    """PARAM_ORDER, FIELD and MODEL of the shim against enum ebm_param / ebm_field / ebm_model."""
    hdr = open(os.path.join(ROOT, "include", "ebm_hip.h")).read()
    jl = open(os.path.join(ROOT, "julia", "EBMHip.jl")).read()
    params = re.findall(r"EBM_P_([A-Za-z0-9]+)", re.search(r"enum ebm_param \{(.*?)\};", hdr, flags=re.S).group(1))
    params = [p for p in params if p != "COUNT"]
    jl_params = re.findall(r":([A-Za-z0-9]+)", re.search(r"const PARAM_ORDER = \((.*?)\)", jl, flags=re.S).group(1))
    assert jl_params == params
    fields = [f for f in re.findall(r"EBM_F_([A-Za-z0-9]+)", re.search(r"enum ebm_field \{(.*?)\};", hdr, flags=re.S).group(1))
              if f != "COUNT"]
    jl_fields = dict((k, int(v)) for k, v in re.findall(r":([A-Za-z0-9]+) => (\d+)", re.search(r"const FIELD = Dict\((.*?)\)\n", jl, flags=re.S).group(1)))
    assert jl_fields == {f: i for i, f in enumerate(fields)}
    assert re.search(r"const MODEL = Dict\(:MIZ => 0, :Classic => 1, :MIZ_IMEX => 2\)", jl)
    assert "EBM_MODEL_MIZ = 0, EBM_MODEL_CLASSIC = 1, EBM_MODEL_MIZ_IMEX = 2" in hdr


# ---- property tests of the host logic ------------------------------------------------------------------
from hypothesis import given, settings, strategies as hst  # noqa: E402


@settings(max_examples=200, deadline=None)
@given(ncol=hst.integers(1, 5000), world=hst.integers(1, 64))
def test_shard_columns_tiles_the_columns(pkg, ncol, world):
    """Block partition: contiguous, disjoint, covering, sizes differing by at most one, the
    larger blocks first."""
    slices = [pkg.shard_columns(ncol, world, r) for r in range(world)]
    assert slices[0].start == 0 and slices[-1].stop == ncol
    assert all(a.stop == b.start for a, b in zip(slices, slices[1:]))
    sizes = [s.stop - s.start for s in slices]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True) and sum(sizes) == ncol
    with pytest.raises(ValueError):
        pkg.shard_columns(ncol, world, world)


@settings(max_examples=200, deadline=None)
@given(base=hst.integers(-5, 5), up=hst.integers(1, 6), down=hst.integers(1, 6), hold0=hst.integers(0, 4),
       hold1=hst.integers(0, 4), ru=hst.sampled_from([0.5, 1.0, 2.0]), rd=hst.sampled_from([0.5, 1.0, 2.0]),
       T=hst.floats(0.0, 60.0))
def test_forcing_is_the_references_piecewise_linear_schedule(pkg, oracle, base, up, down, hold0, hold1, ru, rd, T):
    """Forcing(base, peak, cool, holdyrs, rates) (src/infrastructure.jl:208-241, :294-307): integer
    breakpoints, hold / ramp up / hold / ramp down / hold, continuous, identical to the oracle's, and
    the 9 schedule words the device evaluates reproduce it."""
    peak, cool = base + up * ru, base + up * ru - down * rd
    f = pkg.Forcing(float(base), peak, cool, (hold0, hold1), (ru, -rd))
    g = oracle.Forcing(float(base), peak, cool, (hold0, hold1), (ru, -rd))
    assert f.domain == g.domain == (0, hold0, hold0 + up, hold0 + up + hold1, hold0 + up + hold1 + down)
    assert f(T) == g(T)
    d = f.domain
    for edge in d[1:]:                                           # continuity at the breakpoints
        assert abs(f(edge) - f(np.nextafter(float(edge), -1.0))) < 1e-9 or edge == 0
    assert f(0.0) == base and f(d[4] + 1.0) == cool and min(base, peak, cool) <= f(T) <= max(base, peak, cool)
    from energybalancemodel_jl_amd.engine import schedule_words
    w = schedule_words(f)
    v = w[2] if T >= w[8] else (w[0] if T < w[5] else (w[0] + w[3] * (T - w[5]) if T < w[6] else
                                                       (w[1] if T < w[7] else w[1] + w[4] * (T - w[7]))))
    assert v == f(T)
    with pytest.raises(ValueError):
        pkg.Forcing(0.0, 1.0, 0.0, (1, 1), (0.3, -1.0))          # warming time 1/0.3 is not an integer


def test_default_parameter_values_are_the_published_ones(pkg, oracle):
    """The default parameter table (src/infrastructure.jl:407-433), value by value, in the host mirror AND in the oracle (two
    separate transcriptions) against a third statement of them: the first sixteen are Table 1 of Wagner & Eisenman (2015,
    J. Climate 28, 3998-4014) — D, A, B, cw, S0, S1, S2, a0, a2, ai, Fb, k, Lf, F, cg = 0.01 cw, tau_g — and the MIZ ones are as
    the reference's comments give them, with the two that it writes as products of a per-second rate and the seconds of a
    year.  A typo here would change the physics silently: every GPU-vs-oracle test hands the SAME table to both sides."""
    year = 365 * 24 * 3600
    want = dict(D=0.6, A=193.0, B=2.1, cw=9.8, S0=420.0, S1=338.0, S2=240.0, a0=0.7, a2=0.1, ai=0.4, Fb=4.0, k=2.0, Lf=9.5, F=0.0,
                cg=0.01 * 9.8, tau=1e-5, Tm=0.0, m1=1.6e-6 * year, m2=1.36, alpha=0.66, rl=0.5, Dmin=1.0, Dmax=156.0, hmin=0.1,
                kappa=0.01 * year)
    assert year == 31536000 and abs(want["m1"] - 50.4576) < 1e-12      # the docstring's printed m1 (src/EnergyBalanceModel.jl:34)
    for table in (pkg.default_parval, oracle.default_parval):
        got = dict(table.items()) if hasattr(table, "items") else dict(table)
        assert set(got) == set(want)
        for k, v in want.items():
            assert got[k] == v, (k, got[k], v)
    assert len(pkg.default_parameters("MIZ")) == 22 and len(pkg.default_parameters("Classic")) == 16


def test_season_indices_round_half_to_even_like_julia(pkg, oracle):
    """`round(Int, nt*winter)` (src/infrastructure.jl:128-129) rounds to nearest, ties to even — not truncation, not
    half-up: the docstring's 522 / 1548 at nt = 2000 (522.5 -> 522, 1547.5 -> 1548) and a few made-up seasons."""
    for kw, nt, want in ((dict(), 2000, (522, 1548)), (dict(winter=0.2661, summer=0.7749), 100, (27, 77)),
                         (dict(winter=0.265, summer=0.275), 100, (26, 28)), (dict(winter=0.255, summer=0.745), 100, (26, 74))):
        for mod in (pkg, oracle):
            st = mod.SpaceTime("identity", 10, nt, 1, **kw)
            got = (st.winter.inx, st.summer.inx) if mod is pkg else (st.winter_inx, st.summer_inx)
            assert got == want, (kw, got, want)


def test_integration_doc_lists_every_exported_symbol():
    """INTEGRATION.md is the page a maintainer of the reference binds from: every entry point that include/ebm_hip.h
    declares appears in it (`ebm_timer_*`-style wildcards do not count)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "ebm_hip.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    syms = sorted(set(re.findall(r"\b(ebm_[a-z_]+)\s*\(", hdr)))
    assert len(syms) >= 27
    missing = [s for s in syms if not re.search(r"\b" + s + r"\b", doc)]
    assert not missing, missing


def test_bench_gpus_n_refuses_without_devices():
    """`python bench.py --gpus 2` on a machine with fewer than two GPUs exits non-zero and says why — it never runs
    one GPU and reports it as two (here: no GPU at all; on the GPU box tests/test_gpu_configs.py checks the one-GPU
    case and the two-rank rehearsal).  `--gpus` that disagrees with a launcher's world size is refused as well."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "GPU" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "must agree" in r.stderr and r.stdout.strip() == ""


def test_bench_names_the_kernel_the_library_picks():
    """bench.py's roofline block names the kernel of a workload by the library's own rule (csrc/ebm_kernels.hip:
    fused_state_in_lds; csrc/ebm_runtime.hip: ebm_create_ex): per-step kernel at K = 1; fused with the state in registers for
    a few columns of up to 512 threads and for two cells per thread; resident in LDS for longer meridians, for the
    extension, for more columns than the register kernel runs in one round, or when told so."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    name = bench.kernel_name
    t256 = {"threads": 256, "cells_per_thread": 4}
    assert name("Classic", 64, t256) == "classic_step_kernel"
    assert name("MIZ", 1, t256, 16384) == "miz_step_kernel"
    assert name("MIZ", 64, t256, 256, 256) == "miz_fused_kernel" and name("MIZ", 64, t256, 257, 256) == "miz_resident_kernel"
    assert name("MIZ", 64, {"threads": 64, "cells_per_thread": 4}, 1024, 256) == "miz_fused_kernel"
    assert name("MIZ", 64, {"threads": 64, "cells_per_thread": 4}, 1025, 256) == "miz_resident_kernel"
    assert name("MIZ", 64, t256, 16384, 256, False) == "miz_fused_kernel" and name("MIZ", 64, t256, 1, 256, True) == "miz_resident_kernel"
    assert name("MIZ", 64, {"threads": 1024, "cells_per_thread": 4}, 1, 256, False) == "miz_resident_kernel"
    assert name("MIZ_IMEX", 64, {"threads": 64, "cells_per_thread": 4}, 1, 256, False) == "miz_resident_kernel"
    assert name("MIZ", 64, {"threads": 768, "cells_per_thread": 2}, 9999, 256, True) == "miz_fused_kernel"


def test_visible_gpu_count_reads_the_driver_topology(pkg, tmp_path, monkeypatch):
    """Launch decisions count GPUs without initialising HIP: KFD topology nodes with SIMDs, narrowed by the
    *_VISIBLE_DEVICES variables the runtime honours."""
    dev = pkg._devices if hasattr(pkg, "_devices") else __import__("sys").modules[pkg.__name__ + "._devices"]
    for i, simd in enumerate((0, 256, 256, 256)):               # node 0: the CPU
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {16 if simd == 0 else 0}\nsimd_count {simd}\n")
    monkeypatch.setattr(dev, "_KFD_NODES", str(tmp_path / "*" / "properties"))
    assert dev._kfd_gpu_nodes() == 3
    assert dev._visible_filter(3, {}) == 3
    assert dev._visible_filter(3, {"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert dev._visible_filter(3, {"ROCR_VISIBLE_DEVICES": "1", "HIP_VISIBLE_DEVICES": "0,1,2"}) == 1
    assert dev._visible_filter(3, {"HIP_VISIBLE_DEVICES": ""}) == 0
    if not os.path.exists("/dev/kfd"):
        assert pkg.visible_gpu_count() == 0
    port = pkg.free_port()
    assert 1024 < port < 65536
