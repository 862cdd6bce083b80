# EBMHip.jl — Julia shim that routes EnergyBalanceModel.jl's step!/integrate hot path to the
# MI355X library behind include/ebm_hip.h.
#
# NOT EXECUTED IN THIS PIPELINE: no Julia toolchain exists here (SURVEY F3).  It is the binding a
# maintainer of the reference would add; it contains no numerics, only `ccall`s.  The Python
# mirror (energybalancemodel.jl_amd/infrastructure.py) makes the same calls through ctypes and is
# what the tests exercise.
#
#   using EnergyBalanceModel, EBMHip
#   sols = integrate(:MIZ_HIP, st, forcing, par, init)        # same signature, new model tag
#   step!(Val(:MIZ_HIP), t, f, vars, st, par)
module EBMHip

using EnergyBalanceModel
using EnergyBalanceModel.Infrastructure: Vec, Collection, SpaceTime, Forcing, Solutions, default_parval
import EnergyBalanceModel.Infrastructure: step!, integrate

const libebm = get(ENV, "EBM_HIP_LIB", "libebm_hip.so")

# enum ebm_param / ebm_field (include/ebm_hip.h)
const PARAM_ORDER = (:D, :A, :B, :cw, :S0, :S1, :S2, :a0, :a2, :ai, :Fb, :k, :Lf, :F, :cg, :tau,
                     :Tm, :m1, :m2, :alpha, :rl, :Dmin, :Dmax, :hmin, :kappa)
const FIELD = Dict(:Ei => 0, :Ew => 1, :h => 2, :D => 3, :phi => 4, :T0 => 5, :Tw => 6, :Ti => 7,
                   :n => 8, :E => 9, :T => 10, :Tg => 11)
const MODEL = Dict(:MIZ => 0, :Classic => 1)

struct EBMError <: Exception
    msg::String
end
check(rc::Cint, what) = rc == 0 ? nothing :
    throw(EBMError("$what failed ($rc): " * unsafe_string(ccall((:ebm_last_error, libebm), Cstring, ()))))

gridkind(::SpaceTime{identity}) = Cint(0)
gridkind(::SpaceTime) = Cint(1)

parvec(par::Collection{Float64}) =
    Float64[haskey(getfield(par, :dict), k) ? getproperty(par, k) : getproperty(default_parval, k) for k in PARAM_ORDER]

mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(model::Symbol, st::SpaceTime, par::Collection{Float64}; ncol::Int=1, device::Int=0)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ebm_create, libebm), Cint,
                    (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cint),
                    out, MODEL[model], gridkind(st), st.nx, ncol, st.x, parvec(par), st.dt, device), "ebm_create")
        h = new(out[])
        finalizer(x -> ccall((:ebm_destroy, libebm), Cint, (Ptr{Cvoid},), x.ptr), h)
        return h
    end
end

setfield_dev!(h::Handle, f::Symbol, v::Vec) =
    check(ccall((:ebm_set_field, libebm), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), h.ptr, FIELD[f], v), "ebm_set_field")
function getfield_dev(h::Handle, f::Symbol, n::Int)::Vec
    v = Vector{Float64}(undef, n)
    check(ccall((:ebm_get_field, libebm), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), h.ptr, FIELD[f], v), "ebm_get_field")
    return v
end

cos2pit(t::Float64) = cos(2.0*pi * t)              # as written at src/miz.jl:11

# The reference keeps its T0 warm start in a module-level closure (src/miz.jl:47); the mirror of
# that for direct step! calls is one cached handle per (model, grid, parameters).
const _handles = Dict{UInt64,Handle}()
handle_for(model, st, par) = get!(() -> Handle(model, st, par), _handles, hash((model, st.x, st.dt, parvec(par))))

const INIT = Dict(:MIZ => (:Ei, :Ew, :h, :D, :phi), :Classic => (:E, :Tg))
const OUT = Dict(:MIZ => (:Ei, :Ew, :h, :D, :phi, :Tw, :Ti, :n, :E, :T), :Classic => (:E, :Tg, :T, :h))

function hip_step!(model::Symbol, t::Float64, f::Float64, vars::Collection{Vec}, st::SpaceTime, par::Collection{Float64};
                   debug::Union{Expr,Nothing}=nothing, verbose::Bool=false)
    # debug expressions are evaluated inside the reference's step! (src/miz.jl:188-191): not across a C ABI
    isnothing(debug) || return step!(Val(model), t, f, vars, st, par; debug=debug)
    h = handle_for(model, st, par)
    foreach(k -> setfield_dev!(h, k, getproperty(vars, k)), INIT[model])
    if model === :MIZ
        ct, ctn = cos2pit(t), 0.0
    else
        i = round(Int, mod1((t + st.dt/2.0) * st.nt, st.nt))          # src/classic.jl:45
        ct, ctn = cos2pit(st.t[i]), cos2pit(st.t[mod1(i+1, st.nt)])
    end
    check(ccall((:ebm_step, libebm), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Cint), h.ptr, ct, ctn, f, 1), "ebm_step")
    foreach(k -> setproperty!(vars, k, getfield_dev(h, k, st.nx)), OUT[model])
    return vars
end

step!(::Val{:MIZ_HIP}, t::Float64, f::Float64, vars::Collection{Vec}, st::SpaceTime, par::Collection{Float64}; kw...) =
    hip_step!(:MIZ, t, f, vars, st, par; kw...)
step!(::Val{:Classic_HIP}, t::Float64, f::Float64, vars::Collection{Vec}, st::SpaceTime, par::Collection{Float64}; kw...) =
    hip_step!(:Classic, t, f, vars, st, par; kw...)

# integrate(:MIZ_HIP, ...): state stays on the device, savesol! (src/infrastructure.jl:549-591)
# runs there too; the Solutions object is filled exactly as the reference fills it.
function hip_integrate(model::Symbol, st::SpaceTime{F}, forcing::Forcing{C}, par::Collection{Float64}, init::Collection{Vec};
                       lastonly::Bool=true, verbose::Bool=false) where {F, C}
    solvars = model === :MIZ ? Set{Symbol}((:E, :T, :h, :Ei, :Ew, :Ti, :Tw, :D, :phi, :n)) : Set{Symbol}((:E, :T, :h))
    sols = Solutions(st, forcing, par, init, solvars, lastonly)
    names = collect(solvars)
    h = Handle(model, st, par)
    foreach(k -> setfield_dev!(h, k, getproperty(init, k)), INIT[model])
    ctab = cos2pit.(st.t)
    check(ccall((:ebm_set_time_table, libebm), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), h.ptr, st.nt, ctab), "ebm_set_time_table")
    fsteps = Float64[forcing(T) for T in st.T]
    nraw, nx, dur = length(sols.ts), st.nx, st.dur
    raw = Array{Float64,3}(undef, nx, nraw, length(names))         # == C [nvars][nraw][1][nlat]
    win, sum_, avg = (Array{Float64,3}(undef, nx, dur, length(names)) for _ in 1:3)
    check(ccall((:ebm_integrate, libebm), Cint,
                (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Ptr{Cint},
                 Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                h.ptr, st.nt, dur, fsteps, lastonly, st.winter.inx, st.summer.inx, length(names),
                Cint[FIELD[k] for k in names], raw, win, sum_, avg), "ebm_integrate")
    for (vi, k) in enumerate(names)
        setproperty!(sols.raw, k, [raw[:, ti, vi] for ti in 1:nraw])
        setproperty!(sols.seasonal.winter, k, [win[:, y, vi] for y in 1:dur])
        setproperty!(sols.seasonal.summer, k, [sum_[:, y, vi] for y in 1:dur])
        setproperty!(sols.seasonal.avg, k, [avg[:, y, vi] for y in 1:dur])
    end
    if verbose
        c = zeros(Clonglong, 4)
        ccall((:ebm_get_counters, libebm), Cint, (Ptr{Cvoid}, Ptr{Clonglong}), h.ptr, c)
        c[3] > 0 && @warn "Solving for T0 hit the iteration cap at $(c[3]) time steps."
    end
    return sols
end

function integrate(model::Symbol, st::SpaceTime{F}, forcing::Forcing{C}, par::Collection{Float64}, init::Collection{Vec},
                   ::Val{:hip}; kw...) where {F, C}
    return hip_integrate(model, st, forcing, par, init; kw...)
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
    check(ccall((:ebm_set_column_schedule, libebm), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h.ptr, words),
          "ebm_set_column_schedule")
end

end # module EBMHip
