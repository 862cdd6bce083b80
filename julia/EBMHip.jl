# EBMHip.jl — Julia shim that routes EnergyBalanceModel.jl's step!/integrate hot path to the
# MI355X library behind include/ebm_hip.h.
#
# NOT EXECUTED IN THIS PIPELINE: no Julia toolchain exists here (SURVEY F3).  It is the binding a
# maintainer of the reference would add; it contains no numerics, only `ccall`s.  The Python
# mirror (energybalancemodel.jl_amd/infrastructure.py) makes the same calls through ctypes and is
# what the tests exercise.
#
# THE supported calls (the same two forms are documented in INTEGRATION.md and include/ebm_hip.h):
#
#   using EnergyBalanceModel, EBMHip
#   sols = EBMHip.integrate(:MIZ, st, forcing, par, init; lastonly=true, verbose=false)
#   EBMHip.step!(Val(:MIZ), t, f, vars, st, par)
#
# i.e. the reference's own signatures (src/infrastructure.jl:615-618, src/miz.jl:150-154,
# src/classic.jl:37-41) and the reference's own model symbols, as functions of THIS module.
# `EBMHip.integrate` keeps the state on the device for the whole run and fills a `Solutions` exactly
# as the reference does; `EBMHip.step!` is the per-call form (five fields up, ten down per call).
#
# Why there is no `:MIZ_HIP` model tag.  Extending the reference's generic `step!` with a
# `Val{:MIZ_HIP}` method would make `EnergyBalanceModel.integrate(:MIZ_HIP, ...)` callable, and that
# call does the wrong thing without failing: the reference's `integrate` only adds the seven MIZ
# variables to the solution when `model === :MIZ` (src/infrastructure.jl:621-624), so E, T, h alone
# would be stored, and `default_parameters(:MIZ_HIP)` returns the CLASSIC parameter set
# (src/infrastructure.jl:473-474).  The reference dispatches on the symbol's VALUE; a new tag cannot
# be made to mean "MIZ".  Hence: same symbols, functions of this module.
module EBMHip

using EnergyBalanceModel
using EnergyBalanceModel.Infrastructure: Vec, Collection, SpaceTime, Forcing, Solutions, default_parval
import EnergyBalanceModel.Infrastructure

const libebm = get(ENV, "EBM_HIP_LIB", "libebm_hip.so")

# enum ebm_param / ebm_field (include/ebm_hip.h)
const PARAM_ORDER = (:D, :A, :B, :cw, :S0, :S1, :S2, :a0, :a2, :ai, :Fb, :k, :Lf, :F, :cg, :tau,
                     :Tm, :m1, :m2, :alpha, :rl, :Dmin, :Dmax, :hmin, :kappa)
const FIELD = Dict(:Ei => 0, :Ew => 1, :h => 2, :D => 3, :phi => 4, :T0 => 5, :Tw => 6, :Ti => 7,
                   :n => 8, :E => 9, :T => 10, :Tg => 11)
const MODEL = Dict(:MIZ => 0, :Classic => 1, :MIZ_IMEX => 2)
# :MIZ_IMEX is NOT a model of the reference: the MIZ model with its meridional diffusion treated implicitly
# (EBM_MODEL_MIZ_IMEX in include/ebm_hip.h), for time steps beyond the explicit limit.  It exists only through
# EBMHip.integrate / EBMHip.step!; pass default_parameters(:MIZ) with it.
ismiz(model::Symbol) = model === :MIZ || model === :MIZ_IMEX

struct EBMError <: Exception
    msg::String
end
check(rc::Cint, what) = rc == 0 ? nothing :
    throw(EBMError("$what failed ($rc): " * unsafe_string(ccall((:ebm_last_error, libebm), Cstring, ()))))

gridkind(::SpaceTime{identity}) = Cint(0)
gridkind(::SpaceTime) = Cint(1)

parvec(par::Collection{Float64}) =
    Float64[haskey(getfield(par, :dict), k) ? getproperty(par, k) : getproperty(default_parval, k) for k in PARAM_ORDER]

# struct ebm_options (include/ebm_hip.h): launch options; the library reads no environment variable.
struct EBMOptions
    struct_bytes::Cint
    cells_per_thread::Cint      # 0 = default (4); 2: twice the waves per meridian, for a few short meridians
    use_graph::Cint             # -1 = by size
    prefetch_cols::Cint         # -1 = default
    launch_chains::Cint         # -1 / 1 = one launch per step; 2 = two independent chains over the two halves of the columns
    integrate_steps_per_launch::Cint   # -1 = 64 steps per launch where integrate needs only the running sums; 1 = every step its own launch
    fused_state_in_lds::Cint    # -1 = by column count; 0 / 1: fused-K launches keep the state in registers / in LDS (same bits)
end
EBMOptions(; cells_per_thread::Integer=0, use_graph::Integer=-1, prefetch_cols::Integer=-1, launch_chains::Integer=-1,
           integrate_steps_per_launch::Integer=-1, fused_state_in_lds::Integer=-1) =
    EBMOptions(Cint(sizeof(EBMOptions)), Cint(cells_per_thread), Cint(use_graph), Cint(prefetch_cols), Cint(launch_chains),
               Cint(integrate_steps_per_launch), Cint(fused_state_in_lds))

# One ebm_handle_t.  Every ccall that passes `h.ptr` sits inside `GC.@preserve h`: the pointer alone
# does not keep `h` alive, and its finalizer calls ebm_destroy.
mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(model::Symbol, st::SpaceTime, par::Collection{Float64}; ncol::Int=1, device::Int=0,
                    cells_per_thread::Int=(ncol == 1 && st.nx <= 1536 && model !== :MIZ_IMEX) ? 2 : 0)
        haskey(MODEL, model) || throw(MethodError(Infrastructure.step!, (Val(model),)))   # as the reference: no such method
        out = Ref{Ptr{Cvoid}}(C_NULL)
        # The reference's own shapes are ONE meridian: latency-bound, so two latitudes per thread.  Chosen here,
        # explicitly — the library never derives its launch geometry from the number of columns, because the rounding
        # of the tridiagonal solves depends on it (a member must give the same bits alone and in an ensemble).
        opt = Ref(EBMOptions(cells_per_thread=cells_per_thread))
        check(ccall((:ebm_create_ex, libebm), Cint,
                    (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cint, Ref{EBMOptions}),
                    out, MODEL[model], gridkind(st), st.nx, ncol, st.x, parvec(par), st.dt, device, opt), "ebm_create_ex")
        h = new(out[])
        finalizer(destroy!, h)
        return h
    end
end
function destroy!(h::Handle)
    if h.ptr != C_NULL
        ccall((:ebm_destroy, libebm), Cint, (Ptr{Cvoid},), h.ptr)
        h.ptr = C_NULL
    end
    return nothing
end

setfield_dev!(h::Handle, f::Symbol, v::Vec) = GC.@preserve h check(
    ccall((:ebm_set_field, libebm), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), h.ptr, FIELD[f], v), "ebm_set_field")
function getfield_dev(h::Handle, f::Symbol, n::Int)::Vec
    v = Vector{Float64}(undef, n)
    GC.@preserve h check(
        ccall((:ebm_get_field, libebm), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), h.ptr, FIELD[f], v), "ebm_get_field")
    return v
end

cos2pit(t::Float64) = cos(2.0*pi * t)              # as written at src/miz.jl:11

# The reference keeps its T0 warm start in a module-level closure (src/miz.jl:47); the mirror of
# that for direct step! calls is one cached handle per (model, grid, parameters).
const _handles = Dict{UInt64,Handle}()
handle_for(model, st, par) = get!(() -> Handle(model, st, par), _handles, hash((model, st.x, st.dt, parvec(par))))

const INIT = Dict(:MIZ => (:Ei, :Ew, :h, :D, :phi), :Classic => (:E, :Tg), :MIZ_IMEX => (:Ei, :Ew, :h, :D, :phi))
const OUT = Dict(:MIZ => (:Ei, :Ew, :h, :D, :phi, :Tw, :Ti, :n, :E, :T), :Classic => (:E, :Tg, :T, :h),
                 :MIZ_IMEX => (:Ei, :Ew, :h, :D, :phi, :Tw, :Ti, :n, :E, :T))

"""
    EBMHip.step!(Val(model), t, f, vars, st, par; debug=nothing, verbose=false) -> vars

The reference's `step!` (`src/miz.jl:150-196`, `src/classic.jl:37-71`) on the GPU: uploads the
prognostic fields of `vars`, takes one step (`ebm_step`), rebinds the fields of `vars` to fresh
vectors as the reference does, returns `vars`.  The hidden T0 warm start persists between calls like
the reference's (one cached handle per model / grid / parameter set).  `debug` expressions are
evaluated inside the reference's step! (`src/miz.jl:188-191`) and cannot cross a C ABI: with
`debug !== nothing` the call is forwarded to the reference's own method.
"""
function step!(::Val{M}, t::Float64, f::Float64, vars::Collection{Vec}, st::SpaceTime, par::Collection{Float64};
               debug::Union{Expr,Nothing}=nothing, verbose::Bool=false) where M
    model = M::Symbol
    if !isnothing(debug)      # the :Classic method has no `verbose` keyword at the reference commit (SURVEY F7)
        model === :MIZ_IMEX && throw(ArgumentError("debug expressions need the reference's step!, which has no :MIZ_IMEX"))
        return model === :MIZ ? Infrastructure.step!(Val(model), t, f, vars, st, par; debug=debug, verbose=verbose) :
                                Infrastructure.step!(Val(model), t, f, vars, st, par; debug=debug)
    end
    h = handle_for(model, st, par)
    foreach(k -> setfield_dev!(h, k, getproperty(vars, k)), INIT[model])
    if ismiz(model)
        ct, ctn = cos2pit(t), 0.0
    else
        i = round(Int, mod1((t + st.dt/2.0) * st.nt, st.nt))          # src/classic.jl:45
        ct, ctn = cos2pit(st.t[i]), cos2pit(st.t[mod1(i+1, st.nt)])
    end
    before = verbose ? counters(h)[3] : 0
    GC.@preserve h check(ccall((:ebm_step, libebm), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Cint),
                               h.ptr, ct, ctn, f, 1), "ebm_step")
    foreach(k -> setproperty!(vars, k, getfield_dev(h, k, st.nx)), OUT[model])
    verbose && counters(h)[3] > before && @warn "Solving for T0 failed at t=$t."     # src/miz.jl:61-63
    return vars
end

function counters(h::Handle)
    c = zeros(Clonglong, 4)
    GC.@preserve h check(ccall((:ebm_get_counters, libebm), Cint, (Ptr{Cvoid}, Ptr{Clonglong}), h.ptr, c), "ebm_get_counters")
    return c
end

"""
    EBMHip.integrate(model, st, forcing, par, init; lastonly=true, debug=nothing, verbose=false, device=0) -> Solutions

The reference's `integrate` (`src/infrastructure.jl:615-636`) with the state resident on the device
for the whole run: `ebm_integrate` steps `st.nt * st.dur` times and performs `savesol!`
(`:549-591`) and `annual_mean` (`:536-544`) there; the returned `Solutions` holds the same `raw`,
`seasonal.winter / summer / avg` and `ts` as the reference's.  `model` is `:MIZ` or `:Classic`
(the reference's `integrate(:Classic, ...)` throws on its `verbose` keyword at this commit — SURVEY
F7 — this one works and stores E, T, h as `:621` prescribes).  With `debug !== nothing` the call is
forwarded to the reference's own `integrate`.
"""
function integrate(model::Symbol, st::SpaceTime{F}, forcing::Forcing{C}, par::Collection{Float64}, init::Collection{Vec};
                   lastonly::Bool=true, debug::Union{Expr,Nothing}=nothing, verbose::Bool=false,
                   device::Int=0)::Solutions{F,C} where {F, C}
    isnothing(debug) || return Infrastructure.integrate(model, st, forcing, par, init; lastonly=lastonly, debug=debug, verbose=verbose)
    solvars = Set{Symbol}((:E, :T, :h))                                     # src/infrastructure.jl:621-624
    ismiz(model) && union!(solvars, Set{Symbol}((:Ei, :Ew, :Ti, :Tw, :D, :phi, :n)))
    sols = Solutions(st, forcing, par, init, solvars, lastonly)
    names = collect(solvars)
    h = Handle(model, st, par; device=device)          # owns the T0 warm start of this run (re-entrant, unlike src/miz.jl:47)
    try
        foreach(k -> setfield_dev!(h, k, getproperty(init, k)), INIT[model])
        ctab = cos2pit.(st.t)
        fsteps = Float64[forcing(T) for T in st.T]
        nraw, nx, dur = length(sols.ts), st.nx, st.dur
        raw = Array{Float64,3}(undef, nx, nraw, length(names))         # == C [nvars][nraw][1][nlat]
        win, sum_, avg = (Array{Float64,3}(undef, nx, dur, length(names)) for _ in 1:3)
        GC.@preserve h begin
            check(ccall((:ebm_set_time_table, libebm), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), h.ptr, st.nt, ctab),
                  "ebm_set_time_table")
            check(ccall((:ebm_integrate, libebm), Cint,
                        (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Ptr{Cint},
                         Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                        h.ptr, st.nt, dur, fsteps, lastonly, st.winter.inx, st.summer.inx, length(names),
                        Cint[FIELD[k] for k in names], raw, win, sum_, avg), "ebm_integrate")
        end
        for (vi, k) in enumerate(names)
            setproperty!(sols.raw, k, [raw[:, ti, vi] for ti in 1:nraw])
            setproperty!(sols.seasonal.winter, k, [win[:, y, vi] for y in 1:dur])
            setproperty!(sols.seasonal.summer, k, [sum_[:, y, vi] for y in 1:dur])
            setproperty!(sols.seasonal.avg, k, [avg[:, y, vi] for y in 1:dur])
        end
        if verbose
            nfail = counters(h)[3]
            nfail > 0 && @warn "Solving for T0 hit the iteration cap at $nfail time steps."
        end
    finally
        destroy!(h)                                    # explicit: do not wait for the finalizer
    end
    return sols
end

"""
    set_member_forcings!(h, forcings::Vector{<:Forcing})

One `Forcing` per column (ensemble member), evaluated on the device at `st.T[tinx]` of every
step with the reference's own call operator (`src/infrastructure.jl:294-307`).
"""
function set_member_forcings!(h::Handle, forcings::AbstractVector)
    words = Matrix{Float64}(undef, 9, length(forcings))           # column-major == C [ncol][9]
    for (c, f) in enumerate(forcings)
        if f isa Forcing{true}
            words[:, c] = [f.base, f.base, f.base, 0.0, 0.0, Inf, Inf, Inf, Inf]
        else
            words[:, c] = [f.base, f.peak, f.cool, f.rates[1], f.rates[2],
                           Float64(f.domain[2]), Float64(f.domain[3]), Float64(f.domain[4]), Float64(f.domain[5])]
        end
    end
    GC.@preserve h check(ccall((:ebm_set_column_schedule, libebm), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h.ptr, words),
                         "ebm_set_column_schedule")
end

end # module EBMHip
