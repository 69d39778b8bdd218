# bridgejl_baseline.jl -- times Bridge.jl's OWN CPU hot path on the bench workload (SURVEY 8(d)).
#
# DOCUMENTATION ONLY: there is no `julia` in the build image or on the GPU box, so the build never runs
# this file; bench.py reports the C restatement (oracle/) as `cpu_baseline` with kind "port".  Anyone with
# Julia >= 1.5 and Bridge v0.11.7 can produce the true reference number with
#
#     julia --project=/path/to/Bridge.jl bridge.jl_amd/julia/bridgejl_baseline.jl [iterations]
#
# Workload = bench.py --mode mcmc for ONE chain: FitzHugh-Nagumo partial bridge of
# project_partialbridge/partialbridge_fitzhugh.jl (aux "linearised_end", endpoint "extreme": v = 1.1,
# rho = 0.9), 1001-point tau-grid on [0, 2]; one iteration = sample!(W2) + pCN mix + solve!(Euler) +
# llikelihood(LeftRule) + MH accept  (the loop at partialbridge_fitzhugh.jl:143-176).
# Output: path-steps per second of one Julia thread (Bridge.jl is single-threaded).
using Bridge, StaticArrays, LinearAlgebra, Random

const R2 = SVector{2,Float64}

struct FHN <: ContinuousTimeProcess{R2}
    eps::Float64; s::Float64; gamma::Float64; beta::Float64; sigma::Float64
end
Bridge.b(t, x, P::FHN) = R2((x[1] - x[2] - x[1]^3 + P.s) / P.eps, P.gamma * x[1] - x[2] + P.beta)
Bridge.σ(t, x, P::FHN) = R2(0.0, P.sigma)
Bridge.constdiff(::FHN) = true

# drift of the target linearised at the observed end value v of the first coordinate
struct FHNAuxEnd <: ContinuousTimeProcess{R2}
    eps::Float64; s::Float64; gamma::Float64; beta::Float64; sigma::Float64; v::Float64
end
Bridge.B(t, P::FHNAuxEnd) = @SMatrix [1 / P.eps - 3 * P.v^2 / P.eps  -1 / P.eps; P.gamma  -1.0]
Bridge.β(t, P::FHNAuxEnd) = R2(P.s / P.eps + 2 * P.v^3 / P.eps, P.beta)
Bridge.σ(t, P::FHNAuxEnd) = R2(0.0, P.sigma)
Bridge.a(t, P::FHNAuxEnd) = Bridge.σ(t, P) * Bridge.σ(t, P)'
Bridge.b(t, x, P::FHNAuxEnd) = Bridge.B(t, P) * x + Bridge.β(t, P)
Bridge.constdiff(::FHNAuxEnd) = true

function mh(X, Xo, W, Wo, W2, ll, Po, x0, rho, iterations)
    acc = 0
    for _ in 1:iterations
        sample!(W2, Wiener())
        Wo.yy .= rho * W.yy + sqrt(1 - rho^2) * W2.yy
        solve!(Euler(), Xo, x0, Wo, Po)
        llo = llikelihood(Bridge.LeftRule(), Xo, Po)
        if log(rand()) <= llo - ll
            X, Xo = Xo, X
            W, Wo = Wo, W
            ll = llo
            acc += 1
        end
    end
    acc
end

function run(iterations)
    T, N, v, rho = 2.0, 1001, 1.1, 0.9
    tt = map(s -> s * (2 - s / T), range(0.0, T; length = N))
    P = FHN(0.1, 0.0, 1.5, 0.8, 0.3)
    Pt = FHNAuxEnd(P.eps, P.s, P.gamma, P.beta, P.sigma, v)
    x0 = R2(-0.5, -0.6)
    L = @SMatrix [1.0 0.0]
    Po = Bridge.PartialBridge(tt, P, Pt, L, SVector(v), @SMatrix [1e-10])
    W = sample(tt, Wiener())
    X = solve(Euler(), x0, W, P)
    solve!(Euler(), X, x0, W, Po)
    ll = llikelihood(Bridge.LeftRule(), X, Po)
    Xo, Wo, W2 = copy(X), copy(W), copy(W)
    mh(X, Xo, W, Wo, W2, ll, Po, x0, rho, 10)                       # compile
    acc = 0
    t = @elapsed (acc = mh(X, Xo, W, Wo, W2, ll, Po, x0, rho, iterations))
    println("Bridge.jl CPU path: ", iterations, " iterations x ", N - 1, " steps in ", round(t; digits = 3), " s = ",
            round(iterations * (N - 1) / t; sigdigits = 4), " path-steps/s on 1 thread (acceptance ", acc / iterations, ")")
end

run(length(ARGS) > 0 ? parse(Int, ARGS[1]) : 2000)
