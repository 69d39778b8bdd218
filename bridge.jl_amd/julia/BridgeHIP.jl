# BridgeHIP.jl -- thin Julia shim over libbridgehip.so (C ABI: include/bridgehip.h).
#
# Drop-in for ONE hot path of Bridge.jl v0.11.7: sample!(W, Wiener()) -> pCN mix -> solve!(Euler(), ...)
# -> llikelihood(LeftRule(), ...) -> MH accept, for ensembles of independent paths / chains on an
# MI355X.  It adds methods to Bridge's own generic functions, so call sites keep reading
#
#     W  = sample(tt, Wiener(), HIPEnsemble(npaths))     # instead of a single path
#     X  = solve(HIPEuler(), x0, W, Po)
#     ll = llikelihood(LeftRule(), X, Po)
#
# NOTE: there is no `julia` binary in the build image, so this file has never been parsed or executed -- it is UNTESTED;
# it is the declarative ccall layer a maintainer would add (see INTEGRATION.md) and is kept thin on purpose.  The same ABI
# is exercised by the Python ctypes mirror (bridge.jl_amd/api.py) and from plain C (examples/fhn_chains.c) in the test-suite.
module BridgeHIP

using Bridge, StaticArrays, LinearAlgebra
import Bridge: sample, sample!, solve, solve!, llikelihood, lptilde, LeftRule, Wiener, ContinuousTimeProcess

const lib = get(ENV, "BRIDGEHIP_SO", joinpath(@__DIR__, "..", "libbridgehip.so"))

# ---------------------------------------------------------------- errors (include/bridgehip.h codes)
struct BHIPError <: Exception
    code::Cint
    msg::String
end
Base.showerror(io::IO, e::BHIPError) = print(io, "bridgehip error ", e.code, ": ", e.msg)

mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer = 0, stream::Ptr{Cvoid} = C_NULL)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:bhip_ctx_create, lib), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, stream, r)
        rc == 0 || throw(BHIPError(rc, "bhip_ctx_create failed (no HIP device? there is no CPU fallback)"))
        c = new(r[])
        finalizer(c -> ccall((:bhip_ctx_destroy, lib), Cvoid, (Ptr{Cvoid},), c.h), c)
    end
end
check(ctx::Context, rc) = rc == 0 ? nothing :
    throw(BHIPError(rc, unsafe_string(ccall((:bhip_last_error, lib), Cstring, (Ptr{Cvoid},), ctx.h))))

# bhip_ctx_set_option: the BHIP_OPT_* ids of include/bridgehip.h
const OPT_WAVE_SPECIALISED, OPT_TUNE_PLACEMENT, OPT_MID_VALU, OPT_FUSED_ARITHMETIC, OPT_NOISE_SPEC = 1, 2, 3, 4, 5
"""
    set_option!(c::Context, option, value)

e.g. `set_option!(c, OPT_NOISE_SPEC, 3)`: everything drawn afterwards on `c` uses the Box-Muller stream bhip-philox-v3 instead of the
default bhip-philox-v4 (one normal per 32-bit Philox word through the inverse distribution function); 2: bhip-philox-v2.
"""
set_option!(c::Context, option::Integer, value::Integer) =
    check(c, ccall((:bhip_ctx_set_option, lib), Cint, (Ptr{Cvoid}, Cint, Cint), c.h, option, value))
"`get_option(c, OPT_NOISE_SPEC)` -> 4 | 3 | 2: the stream of normals this context draws (store it with a run's results)"
function get_option(c::Context, option::Integer)
    v = Ref{Cint}(0)
    check(c, ccall((:bhip_ctx_get_option, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}), c.h, option, v))
    Int(v[])
end

const default_ctx = Ref{Union{Nothing,Context}}(nothing)
function default_context()          # (not named `ctx`: a keyword default `ctx = ctx()` would refer to the keyword itself)
    default_ctx[] === nothing && (default_ctx[] = Context())
    default_ctx[]
end

# ---------------------------------------------------------------- process type -> device functor
# trait: hipmodel(P) -> (model id, d, parameter vector).  Users with their own process types add a
# method for one of the registry ids, e.g. for the script's FitzhughDiffusion
#     BridgeHIP.hipmodel(P::FitzhughDiffusion) = (BridgeHIP.MODEL_FHN, 2, [P.ϵ, P.s, P.γ, P.β, P.σ])
# the BHIP_MODEL_* ids of include/bridgehip.h
const MODEL_WIENER, MODEL_OU, MODEL_LINPRO, MODEL_FHN, MODEL_NCLAR = 0, 1, 2, 3, 4
const MODEL_INTDIFF, MODEL_LORENZ, MODEL_FHN2, MODEL_PENDULUM = 5, 6, 7, 8
hipmodel(P) = error("no device functor registered for $(typeof(P)); define BridgeHIP.hipmodel")
hipmodel(P::Bridge.LinPro) = (MODEL_LINPRO, length(P.μ), vcat(vec(collect(P.B)), collect(P.μ), vec(collect(P.σ))))
hipmodel(P::Bridge.Models.FitzHughNagumo) = (MODEL_FHN2, 2, [P.ϵ, P.s, P.γ, P.β, P.σ1, P.σ2])
hipmodel(P::Bridge.Models.Lorenz) = (MODEL_LORENZ, 3, vcat(collect(P.θ), diag(P.σ)))
hipmodel(P::Bridge.Models.Pendulum) = (MODEL_PENDULUM, 2, [P.θ², P.γ])
hipmodel(::Wiener{SVector{d,Float64}}) where {d} = (MODEL_WIENER, d, Float64[])
hipmodel(::Wiener{Float64}) = (MODEL_WIENER, 1, Float64[])

# A process WITHOUT a registry functor: give the body of its Bridge.b method as HIP C++ text; the
# library compiles it for gfx950 with hipRTC (bhip_model_define).  sigma must be constant (d x m').
#     struct DoubleWell <: ContinuousTimeProcess{Float64}; θ::Float64; σ::Float64; end
#     BridgeHIP.hipmodel(P::DoubleWell) = BridgeHIP.userdrift(1, "o[0] = par[0]*(x[0] - x[0]*x[0]*x[0]);", [P.θ], fill(P.σ, 1, 1))
const _user_ids = Dict{Tuple{Int,Int,Int,String},Cint}()
function userdrift(d::Integer, src::String, par::Vector{Float64}, sigma::AbstractMatrix; c::Context = default_context())
    key = (Int(d), size(sigma, 2), length(par), src)
    id = get!(_user_ids, key) do
        r = Ref{Cint}(0)
        check(c, ccall((:bhip_model_define, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cstring, Ref{Cint}),
            c.h, d, size(sigma, 2), length(par), src, r))
        r[]
    end
    (id, Int(d), vcat(par, vec(collect(Float64, sigma))))
end

# The same with a state-dependent Bridge.σ(t, x, P): `sigsrc` fills s (d x m', column-major, zero-initialised)
#     BridgeHIP.hipmodel(P::MyDiff) = BridgeHIP.userprocess(1, 1, "o[0] = par[0]*(par[1] - x[0]);",
#                                                           "s[0] = par[2]*sqrt(1.0 + x[0]*x[0]);", [P.κ, P.θ, P.s])
function userprocess(d::Integer, mp::Integer, bsrc::String, sigsrc::String, par::Vector{Float64}; c::Context = default_context())
    key = (Int(d), Int(mp), length(par), bsrc * "\0" * sigsrc)
    id = get!(_user_ids, key) do
        r = Ref{Cint}(0)
        check(c, ccall((:bhip_model_define_sigma, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cstring, Cstring, Ref{Cint}),
            c.h, d, mp, length(par), bsrc, sigsrc, r))
        r[]
    end
    (id, Int(d), par)
end

# auxiliary process: constant coefficients by default, any Bridge.B/β/a methods through a C callback
function aux_callback(t::Cdouble, B::Ptr{Cdouble}, beta::Ptr{Cdouble}, a::Ptr{Cdouble}, user::Ptr{Cvoid})::Cvoid
    Pt = unsafe_pointer_to_objref(user)[]
    Bt = Bridge.B(t, Pt); bt = Bridge.β(t, Pt); at = Bridge.a(t, Pt)
    d = length(bt)
    for k in 1:d*d
        unsafe_store!(B, Bt[k], k); unsafe_store!(a, at[k], k)       # column-major, like SMatrix
    end
    for k in 1:d
        unsafe_store!(beta, bt[k], k)
    end
    nothing
end

# ---------------------------------------------------------------- ensembles (SoA on the device)
"`HIPEnsemble(npaths)`: request marker for ensemble versions of sample/solve"
struct HIPEnsemble
    npaths::Int
    seed::UInt64
    iter::UInt32
    path0::UInt32
end
HIPEnsemble(n; seed = 0, iter = 0, path0 = 0) = HIPEnsemble(n, seed, iter, path0)

"""
An ensemble of sample paths in HBM, Float64, struct of arrays: element (i, k, p) (0-based) of buffer j at `(i*dim + k)*ld + (p - j*ld)`.
ONE container (src/types.jl:71-81 has one `SamplePath` type; an ensemble is a loop there), kept in `length(ptrs)` device buffers:
paths `j*ld : (j+1)*ld - 1` live in buffer j.  Ensembles of 1 GiB and more get TWO buffers lying in different 96-GiB pieces of the
device memory (`bhip_alloc_apart`): an ensemble written by one kernel is one write stream, and a write stream inside one piece moves
4.3-4.4 TB/s where two streams in two pieces move 5.8-6.0.  Every method below walks the column ranges that lie inside one buffer of
every ensemble involved (`segments`); the fused proposal `sample_solve!` writes all buffers in one launch.
"""
mutable struct EnsemblePath{T} <: Bridge.AbstractPath{T}
    tt::Vector{Float64}
    ptrs::Vector{Ptr{Cvoid}}     # the buffers (bhip_alloc_apart)
    dim::Int
    npaths::Int
    ld::Int                      # leading dimension of every buffer = paths per buffer (a multiple of 64 when there are several)
    apart::Int                   # how many of the buffers were found pairwise in different pieces
    ctx::Context
end
Base.length(X::EnsemblePath) = length(X.tt)
const PARTS_MIN_BYTES = 1 << 30
function EnsemblePath{T}(tt, dim, npaths, c::Context = default_context(); parts::Union{Nothing,Integer} = nothing) where {T}
    nparts = parts === nothing ? (8 * length(tt) * dim * npaths >= PARTS_MIN_BYTES && dim <= 12 ? 2 : 1) : Int(parts)
    ld = nparts == 1 ? npaths : cld(cld(npaths, nparts), 64) * 64
    ptrs = fill(Ptr{Cvoid}(C_NULL), nparts)
    apart = Ref{Cint}(0)
    check(c, ccall((:bhip_alloc_apart, lib), Cint, (Ptr{Cvoid}, Cint, Csize_t, Ptr{Ptr{Cvoid}}, Ref{Cint}),
        c.h, nparts, 8 * length(tt) * dim * ld, ptrs, apart))
    X = EnsemblePath{T}(collect(Float64, tt), ptrs, dim, npaths, ld, Int(apart[]), c)
    # (the buffers hold a reference to their context: finalizers may run in any order)
    finalizer(x -> ccall((:bhip_free_apart, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}), x.ctx.h, length(x.ptrs), x.ptrs), X)
end
"device address of column p (0-based); its buffer's leading dimension is `X.ld`"
colptr(X::EnsemblePath, p::Integer) = Ptr{Cdouble}(X.ptrs[p ÷ X.ld + 1]) + 8 * (p % X.ld)
"column ranges (start (0-based), n) lying inside ONE buffer of every ensemble given"
function segments(X::EnsemblePath, others::EnsemblePath...)
    cuts = Set{Int}((0, X.npaths))
    for E in (X, others...)
        E.npaths == X.npaths || error("ensembles differ in the number of paths")
        for c in E.ld:E.ld:(E.npaths - 1)
            push!(cuts, c)
        end
    end
    cs = sort!(collect(cuts))
    [(cs[k], cs[k + 1] - cs[k]) for k in 1:length(cs) - 1]
end
"download path p (1-based) as an ordinary Bridge.SamplePath"
function Bridge.SamplePath(X::EnsemblePath{T}, p::Integer) where {T}
    yy = Vector{T}(undef, length(X))
    check(X.ctx, ccall((:bhip_download_aos, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Clong, Clong, Clong, Ptr{Cdouble}),
        X.ctx.h, colptr(X, p - 1), length(X), X.dim, X.ld, 0, 1, pointer(reinterpret(Float64, yy))))
    Bridge.SamplePath(copy(X.tt), yy)
end

# ---------------------------------------------------------------- proposals
"device twin of GuidedBridge / PartialBridge / PartialBridgeνH (holds the bhip_proposal handle)"
mutable struct HIPProposal{T} <: ContinuousTimeProcess{T}
    h::Ptr{Cvoid}
    tt::Vector{Float64}
    d::Int
    mp::Int
    ctx::Context
    keep::Any            # roots the callback closure
end

function _proposal(tt, P, Pt, c::Context)
    id, d, par = hipmodel(P)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    ttv = collect(Float64, tt)
    check(c, ccall((:bhip_proposal_create, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint, Ptr{Cdouble}, Cint, Ref{Ptr{Cvoid}}),
        c.h, ttv, length(ttv), id, d, par, length(par), r))
    keep = Ref{Any}(Pt)
    cb = @cfunction(aux_callback, Cvoid, (Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}))
    check(c, ccall((:bhip_proposal_set_aux_callback, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cdouble}),
        r[], cb, pointer_from_objref(keep), Pt isa Bridge.LinPro ? 1 : 0, Pt isa Bridge.LinPro ? collect(Pt.μ) : C_NULL))
    mp = P isa Bridge.LinPro ? d : size(Bridge.σ(ttv[1], zero(valtype(P)), P), 2)
    Po = HIPProposal{valtype(P)}(r[], ttv, d, mp, c, keep)
    finalizer(p -> ccall((:bhip_proposal_destroy, lib), Cvoid, (Ptr{Cvoid},), p.h), Po)
end

"GuidedBridge(tt, P, Pt, v, h♢)  src/guip.jl:172-180"
function HIPGuidedBridge(tt, P, Pt, v, h = nothing; ctx = default_context())
    Po = _proposal(tt, P, Pt, ctx)
    check(ctx, ccall((:bhip_proposal_guide_hv, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}),
        Po.h, collect(Float64, v), h === nothing ? C_NULL : vec(collect(Float64, h))))
    Po
end
"PartialBridge(tt, P, Pt, L, v, Σ)  src/partialbridge.jl:42-50"
function HIPPartialBridge(tt, P, Pt, L, v, Σ = nothing; ctx = default_context())
    Po = _proposal(tt, P, Pt, ctx)
    check(ctx, ccall((:bhip_proposal_guide_lmmu, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
        Po.h, size(L, 1), vec(collect(Float64, L)), collect(Float64, v), Σ === nothing ? C_NULL : vec(collect(Float64, Σ))))
    Po
end
"PartialBridgeνH(tt, P, Pt, L, v, ϵ, Σ)  src/partialbridgenuH.jl:134-145  (inplace=true: PartialBridge!)"
function HIPPartialBridgeνH(tt, P, Pt, L, v, ϵ, Σ = nothing; inplace = false, ctx = default_context())
    Po = _proposal(tt, P, Pt, ctx)
    check(ctx, ccall((:bhip_proposal_guide_nuh, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Cint),
        Po.h, size(L, 1), vec(collect(Float64, L)), collect(Float64, v), ϵ, Σ === nothing ? C_NULL : vec(collect(Float64, Σ)), inplace))
    Po
end
"wrap guide arrays computed by Bridge.jl's own constructors (bhip_proposal_guide_arrays)"
function HIPProposal(Po::Bridge.PartialBridge; ctx = default_context())
    Q = _proposal(Po.tt, Po.Target, Po.Pt, ctx)
    m = length(Po.v)
    check(ctx, ccall((:bhip_proposal_guide_arrays, lib), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
        Q.h, 2, m, reinterpret(Float64, Po.L), reinterpret(Float64, Po.M), reinterpret(Float64, Po.μ), collect(Float64, Po.v)))
    Q
end

function lptilde(Po::HIPProposal, u)
    r = Ref{Cdouble}(0)
    check(Po.ctx, ccall((:bhip_proposal_lptilde, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ref{Cdouble}), Po.h, collect(Float64, u), r))
    r[]
end

# ---------------------------------------------------------------- sample / solve / llikelihood
struct HIPEuler <: Bridge.SDESolver end

"sample(tt, Wiener{T}(), HIPEnsemble(n)): n Wiener paths in HBM  (src/wiener.jl:11-15)"
function sample(tt, P::Wiener{T}, E::HIPEnsemble; ctx = default_context()) where {T}
    W = EnsemblePath{T}(tt, length(zero(T)), E.npaths, ctx)
    sample!(W, P; seed = E.seed, iter = E.iter, path0 = E.path0)
end
"sample!(W, Wiener())  src/wiener.jl:24-58 (the noise is keyed by the global path id: path0 + column)"
function sample!(W::EnsemblePath{T}, ::Wiener{T}; seed = 0, iter = 0, path0 = 0) where {T}
    if W.dim <= 12       # all buffers by ONE launch
        check(W.ctx, ccall((:bhip_wiener_sample_parts, lib), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}, Clong, Clong, Clong, UInt64, UInt32, UInt32),
            W.ctx.h, W.tt, length(W.tt), W.dim, length(W.ptrs), W.ptrs, W.ld, W.ld, W.npaths, seed, iter, path0))
        return W
    end
    for (a, n) in segments(W)
        check(W.ctx, ccall((:bhip_wiener_sample, lib), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Clong, Clong, UInt64, UInt32, UInt32),
            W.ctx.h, W.tt, length(W.tt), W.dim, colptr(W, a), W.ld, n, seed, iter, path0 + a))
    end
    W
end

"solve!(HIPEuler(), Y, u, W, Po): src/euler.jl:247-268 for every path; returns Y (endpoints are Y[end]); `ll`: device pointer to npaths values or C_NULL"
function solve!(::HIPEuler, Y::EnsemblePath, u, W::EnsemblePath, Po::HIPProposal; ll::Ptr{Cdouble} = Ptr{Cdouble}(C_NULL), skip = 0)
    length(W) != length(Y) && error("Y and W differ in length.")        # src/euler.jl:251
    Y.tt[:] = Po.tt                                                       # src/euler.jl:256
    # all buffers of W and Y by ONE launch (one path per lane); the tile kernel (BHIP_EUNSUPPORTED = -3) takes a buffer per launch: range by range
    rc = ccall((:bhip_solve_parts, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Ptr{Cvoid}}, Clong, Clong, Cint, Ptr{Ptr{Cvoid}}, Clong, Clong, Ptr{Cdouble}, Cint, Clong),
        Y.ctx.h, Po.h, collect(Float64, u), length(W.ptrs), W.ptrs, W.ld, W.ld, length(Y.ptrs), Y.ptrs, Y.ld, Y.ld, ll, skip, Y.npaths)
    rc == -3 || (check(Y.ctx, rc); return Y)
    for (a, n) in segments(Y, W)
        check(Y.ctx, ccall((:bhip_solve, lib), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Clong, Ptr{Cdouble}, Clong, Ptr{Cdouble}, Cint, Clong),
            Y.ctx.h, Po.h, collect(Float64, u), C_NULL, colptr(W, a), W.ld, colptr(Y, a), Y.ld, ll == C_NULL ? ll : ll + 8 * a, skip, n))
    end
    Y
end
solve(m::HIPEuler, u, W::EnsemblePath, Po::HIPProposal{T}; kw...) where {T} =
    solve!(m, EnsemblePath{T}(W.tt, Po.d, W.npaths, W.ctx; parts = length(W.ptrs) > 1 ? length(W.ptrs) : nothing), u, W, Po; kw...)
"deprecated alias  src/deprecated.jl:16-17"
bridge!(Y::EnsemblePath, u, W::EnsemblePath, Po::HIPProposal) = solve!(HIPEuler(), Y, u, W, Po)

"""
    sample_solve!(X, u, Po; ll, skip, seed, iter, path0)

The fused proposal -- `sample!(W, Wiener())`, `solve!(Euler(), X, u, W, Po)`, `llikelihood(LeftRule(), X, Po; skip)` in one kernel with
in-kernel Philox noise, W never stored -- into `X`, ALL its buffers by one launch (`bhip_sample_solve_parts`; one buffer: the values and
the kernel of `bhip_sample_solve`).  `ll`: device pointer to npaths values or C_NULL.
"""
function sample_solve!(X::EnsemblePath, u, Po::HIPProposal; ll::Ptr{Cdouble} = Ptr{Cdouble}(C_NULL), skip = 0, seed = 0, iter = 0, path0 = 0)
    X.tt[:] = Po.tt
    check(X.ctx, ccall((:bhip_sample_solve_parts, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Ptr{Cvoid}}, Clong, Clong, Ptr{Cdouble}, Cint, Clong, UInt64, UInt32, UInt32),
        X.ctx.h, Po.h, collect(Float64, u), length(X.ptrs), X.ptrs, X.ld, X.ld, ll, skip, X.npaths, seed, iter, path0))
    X
end
sample_solve(u, Po::HIPProposal{T}, E::HIPEnsemble; kw...) where {T} =
    sample_solve!(EnsemblePath{T}(Po.tt, Po.d, E.npaths, Po.ctx), u, Po; seed = E.seed, iter = E.iter, path0 = E.path0, kw...)

"llikelihood(LeftRule(), X, Po; skip): one value per path  src/partialbridge.jl:67-77 etc."
function llikelihood(::LeftRule, X::EnsemblePath, Po::HIPProposal; skip = 0)
    out = Vector{Float64}(undef, X.npaths)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(X.ctx, ccall((:bhip_malloc, lib), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), X.ctx.h, 8 * X.npaths, r))
    rc = ccall((:bhip_llikelihood_parts, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}, Clong, Clong, Ptr{Cdouble}, Cint, Clong),
        X.ctx.h, Po.h, length(X.ptrs), X.ptrs, X.ld, X.ld, Ptr{Cdouble}(r[]), skip, X.npaths)      # ONE launch over all buffers ...
    rc == -3 || check(X.ctx, rc)
    if rc == -3                                                                                      # ... the tile kernel: range by range
        for (a, n) in segments(X)
            check(X.ctx, ccall((:bhip_llikelihood, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Clong, Ptr{Cdouble}, Cint, Clong),
                X.ctx.h, Po.h, colptr(X, a), X.ld, Ptr{Cdouble}(r[]) + 8 * a, skip, n))
        end
    end
    check(X.ctx, ccall((:bhip_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), X.ctx.h, out, r[], 8 * X.npaths))
    ccall((:bhip_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), X.ctx.h, r[])
    out
end

"""
girsanov(X::EnsemblePath, Po::HIPProposal, Pt): src/diffusion.jl:109-123 for every stored path; P = the
target of `Po`, `Pt` = the same process type with other parameters (example/fitzhugh_nagumo_full.jl:317)
or `Wiener` (test/guip.jl:72).
"""
function Bridge.girsanov(X::EnsemblePath, Po::HIPProposal, Pt)
    par = Pt isa Wiener ? Float64[] : hipmodel(Pt)[3]
    out = Vector{Float64}(undef, X.npaths)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(X.ctx, ccall((:bhip_malloc, lib), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), X.ctx.h, 8 * X.npaths, r))
    check(X.ctx, ccall((:bhip_girsanov_parts, lib), Cint,      # all buffers by ONE launch
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Ptr{Ptr{Cvoid}}, Clong, Clong, Ptr{Cdouble}, Clong),
        X.ctx.h, Po.h, Pt isa Wiener ? C_NULL : par, length(par), length(X.ptrs), X.ptrs, X.ld, X.ld, Ptr{Cdouble}(r[]), X.npaths))
    check(X.ctx, ccall((:bhip_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), X.ctx.h, out, r[], 8 * X.npaths))
    ccall((:bhip_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), X.ctx.h, r[])
    out
end

# ---------------------------------------------------------------- the MCMC loop of the scripts
"An ensemble of pCN chains (bhip_chains); the device state (GBs of W / Xo) is released by the finalizer."
mutable struct Chains
    h::Ptr{Cvoid}
    ctx::Context
    n::Int
    function Chains(c::Context, h::Ptr{Cvoid}, n)
        ch = new(h, c, n)
        finalizer(x -> ccall((:bhip_chains_destroy, lib), Cvoid, (Ptr{Cvoid},), x.h), ch)
    end
end

"RCCL communicator of the library (bhip_comm): one process per GPU; `id` = the 128 bytes rank 0 drew with `comm_unique_id()`"
mutable struct Comm
    h::Ptr{Cvoid}
    ctx::Context
    nranks::Int
    function Comm(c::Context, nranks::Integer, rank::Integer, id::Vector{UInt8})
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(c, ccall((:bhip_comm_init_rank, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}, Ref{Ptr{Cvoid}}), c.h, nranks, rank, id, r))
        cm = new(r[], c, nranks)
        finalizer(x -> ccall((:bhip_comm_destroy, lib), Cvoid, (Ptr{Cvoid},), x.h), cm)
    end
end
function comm_unique_id()
    id = zeros(UInt8, 128)
    ccall((:bhip_comm_unique_id, lib), Cint, (Ptr{UInt8}, Csize_t), id, 128) == 0 || error("RCCL not available")
    id
end

"""
    mcmc(Po, x0, iterations; ρ, nchains, seed, path0, comm) -> (acc, ll, chains[, stats])

`nchains` independent copies of the loop in project_partialbridge/partialbridge_fitzhugh.jl:125-176
(one chain per GPU lane; sample!, pCN mix, solve!, llikelihood and the accept fused in one kernel launch per iteration).
Multi-GPU: one process per GPU, `path0 = rank*nchains`, `comm = Comm(ctx, nranks, rank, id)`: the acceptance / log-weight
statistics block of every rank is all-gathered over RCCL (`stats[:, r]` = {n, iterations, Σacc, Σll, Σll², min ll, max ll, Σacc²}).
"""
function mcmc(Po::HIPProposal, x0, iterations; ρ = 0.9, nchains = 1, seed = 0, path0 = 0, skip = 0, store_X = true, comm = nothing)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    c = Po.ctx
    check(c, ccall((:bhip_chains_create, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Clong, UInt32, UInt64, Cint, Ref{Ptr{Cvoid}}),
        c.h, Po.h, nchains, path0, seed, store_X ? 1 : 0, r))
    ch = Chains(c, r[], nchains)
    check(c, ccall((:bhip_chains_init, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint), ch.h, collect(Float64, x0), skip))
    check(c, ccall((:bhip_chains_step, lib), Cint, (Ptr{Cvoid}, Cdouble, Cint, Cint), ch.h, ρ, iterations, skip))
    ll = Vector{Float64}(undef, nchains); acc = Vector{Int64}(undef, nchains)
    check(c, ccall((:bhip_chains_get, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Int64}), ch.h, ll, acc))
    comm === nothing && return acc, ll, ch
    sd = Ref{Ptr{Cvoid}}(C_NULL); ad = Ref{Ptr{Cvoid}}(C_NULL)
    check(c, ccall((:bhip_malloc, lib), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), c.h, 64, sd))
    check(c, ccall((:bhip_malloc, lib), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), c.h, 64 * comm.nranks, ad))
    check(c, ccall((:bhip_chains_stats, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ch.h, sd[]))
    check(c, ccall((:bhip_comm_allgather_stats, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), comm.h, sd[], ad[]))
    stats = Matrix{Float64}(undef, 8, comm.nranks)
    check(c, ccall((:bhip_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), c.h, stats, ad[], 64 * comm.nranks))
    ccall((:bhip_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), c.h, sd[]); ccall((:bhip_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), c.h, ad[])
    acc, ll, ch, stats
end

"""
    step_group!(chs::Vector{Chains}, ρ, iterations; skip = -1)
    stats_group!(chs::Vector{Chains}, stats_dev::Vector{Ptr{Cvoid}})

ONE `ccall` that runs `iterations` pCN iterations on every ensemble of `chs` -- one per device of the node, each created on its own
`Context` -- issuing the launches round-robin on the contexts' streams (`bhip_chains_step_group`): a single-threaded session keeps
all GPUs busy with one crossing per call.  `skip = -1`: the skip every ensemble was initialised with.
"""
function step_group!(chs::Vector{Chains}, ρ, iterations; skip = -1)
    hs = Ptr{Cvoid}[ch.h for ch in chs]
    check(chs[1].ctx, ccall((:bhip_chains_step_group, lib), Cint, (Cint, Ptr{Ptr{Cvoid}}, Cdouble, Cint, Cint), length(hs), hs, ρ, iterations, skip))
    chs
end
"""
    iterations(ch::Chains) -> Int

pCN iterations the ensemble has completed (`bhip_chains_iterations`; the library keeps the count -- after a `step_group!` that threw,
the ensembles of the group stand at different counts).
"""
function iterations(ch::Chains)
    r = Ref{UInt32}(0)
    check(ch.ctx, ccall((:bhip_chains_iterations, lib), Cint, (Ptr{Cvoid}, Ref{UInt32}), ch.h, r))
    Int(r[])
end
"""
    placement(ch::Chains) -> (tries, gbs_same_piece, gbs_kept, piece_w, piece_xo)

What `BHIP_OPT_TUNE_PLACEMENT` did for a large ensemble (`bhip_chains_placement_info`, `bhip_chains_placement_pieces`): candidates tested,
GB/s of two write streams into one piece of the device memory and into the kept (W, Xo) pair, the pieces of the context's map.
"""
function placement(ch::Chains)
    n = Ref{Cint}(0); a = Ref{Cfloat}(0); b = Ref{Cfloat}(0); pw = Ref{Cint}(-1); px = Ref{Cint}(-1)
    check(ch.ctx, ccall((:bhip_chains_placement_info, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cfloat}, Ref{Cfloat}), ch.h, n, a, b))
    check(ch.ctx, ccall((:bhip_chains_placement_pieces, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}), ch.h, pw, px))
    (tries = Int(n[]), gbs_same_piece = a[], gbs_kept = b[], piece_w = Int(pw[]), piece_xo = Int(px[]))
end
function stats_group!(chs::Vector{Chains}, stats_dev::Vector{Ptr{Cvoid}})
    hs = Ptr{Cvoid}[ch.h for ch in chs]
    check(chs[1].ctx, ccall((:bhip_chains_stats_group, lib), Cint, (Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}), length(hs), hs, stats_dev))
    stats_dev
end

# ---------------------------------------------------------------- the smoothing loop (supplements/smoothing/smoothing.jl:99-213)
"""
    SegChains(Po::Vector{<:HIPProposal}, π0μ, π0C, nchains; seed, path0, skip, mcnext)

`nchains` copies of the smoothing loop over the chained segments `Po` (GuidedBridge's linked backwards by `Bridge.gpupdate`,
built by the caller as the script does): pCN move of the start drawn from `π0 = Gaussian(π0μ, π0C*π0C')`, every segment
proposed with the end point of the previous one, ONE accept per iteration, `mcnext!` per chain (bhip_segchains_*).
`step!(sc, w_old, w_new)`: iterations with the script's weights `sqrt(1 - ρ_)`, `sqrt(ρ_)` per iteration;
`adapt!(sc, L, Σ, obs, H♢T, vT)`: the adaptation block (:130-160) for every chain at once on the device.
"""
mutable struct SegChains
    h::Ptr{Cvoid}
    ctx::Context
    m::Int
    n::Int
    keep::Vector{HIPProposal}        # the proposals must outlive the ensemble
    function SegChains(Po::Vector{<:HIPProposal}, μ, C, nchains::Integer; seed = 0, path0 = 0, skip = 0, mcnext = true)
        c = Po[1].ctx
        hs = Ptr{Cvoid}[P.h for P in Po]
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(c, ccall((:bhip_segchains_create, lib), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}, Clong, UInt32, UInt64, Cint, Ref{Ptr{Cvoid}}),
            c.h, length(Po), hs, nchains, path0, seed, mcnext ? 1 : 0, r))
        sc = new(r[], c, length(Po), nchains, collect(HIPProposal, Po))
        finalizer(x -> ccall((:bhip_segchains_destroy, lib), Cvoid, (Ptr{Cvoid},), x.h), sc)
        check(c, ccall((:bhip_segchains_init, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint),
            sc.h, collect(Float64, μ), vec(collect(Float64, C)), skip))
        sc
    end
end
function step!(sc::SegChains, w_old::AbstractVector, w_new::AbstractVector)
    check(sc.ctx, ccall((:bhip_segchains_step, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint),
        sc.h, collect(Float64, w_old), collect(Float64, w_new), length(w_old)))
    sc
end
"`obs[i]`: the observation at the LEFT end of segment i (V.yy[i]); `(H♢T, vT) = gpupdate(prior, V.yy[end])`"
function adapt!(sc::SegChains, L, Σ, obs::AbstractVector, H♢T, vT; hwindow = 0, newblock = true, doaccept = false)
    mo = size(L, 1)
    o = Float64[]
    for i in 1:sc.m
        append!(o, collect(Float64, obs[i]))
    end
    check(sc.ctx, ccall((:bhip_segchains_adapt_device, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint),
        sc.h, mo, vec(collect(Float64, L)), vec(collect(Float64, Σ)), o, vec(collect(Float64, H♢T)), collect(Float64, vT),
        hwindow, (newblock ? 1 : 0) | (doaccept ? 2 : 0)))
    sc
end
"(ll [m x nchains], acc, y0 [d x nchains]) of the ensemble"
function state(sc::SegChains, d::Integer)
    ll = Matrix{Float64}(undef, sc.n, sc.m); acc = Vector{Int64}(undef, sc.n); y0 = Matrix{Float64}(undef, d, sc.n)
    check(sc.ctx, ccall((:bhip_segchains_get, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Int64}, Ptr{Cdouble}), sc.h, ll, acc, y0))
    permutedims(ll), acc, y0      # the library writes ll as [m][nchains] row-major = an nchains x m column-major matrix
end

end # module
