# bridgejl_fixtures.jl -- turns "parity unpinned" into "pinned": runs Bridge.jl's OWN solve!(Euler(), ...) and
# llikelihood(LeftRule(), ...) on externally supplied Wiener paths and writes what it computes in the format of
# tests/golden/, so that the CPU oracle (and through it the HIP kernels) can be compared with the reference bit for bit.
#
# NEVER EXECUTED BY THE BUILD: there is no `julia` in the build image or on the GPU box.  Anyone with Julia >= 1.5 and
# Bridge v0.11.7 -- but mind the VERSION (written to <outdir>/VERSION.txt, read by the test):
#
#   The bit-for-bit comparison is DEFINED ON JULIA <= 1.6 (Bridge.jl's CI pins 1.5).  src/partialbridge.jl:54,57 write the guided
#   drift as `a*L'*M*q` and `L'*M*q`.  Up to Julia 1.6 `*` with several arguments is the generic left fold, ((a*L')*M)*q and (L'*M)*q:
#   what the oracle (oracle/bridge_oracle.c bo_guided_drift) and the kernels (bhip_path_kernel.h guide_terms, LMMU branch) evaluate.
#   From Julia 1.7 on LinearAlgebra defines 3- and 4-argument `*` for AbstractMatrix ... AbstractVector chains and associates
#   them from the RIGHT (A*(B*x), A*B*(C*x)); SMatrix / SVector are AbstractMatrix / AbstractVector, so THE SAME SOURCE LINE rounds
#   differently: 1-ulp differences per step, amplified by the stiff guide near the end point.  On Julia >= 1.7 the test therefore
#   compares at the stated fp64 tolerance (1e-9 on paths, 1e-8 on log-likelihoods) and says why; a mismatch there under `==`
#   would not be a parity failure.  (GuidedBridge, src/guip.jl:192-193, has no such chain: H \ (V - x) and a*r.)
#
#     python tests/golden/export_for_julia.py            # writes tests/golden/julia_in/<case>_{tt,W}.csv
#     julia --project=/path/to/Bridge.jl bridge.jl_amd/julia/bridgejl_fixtures.jl tests/golden/julia_in tests/golden/julia_out
#     python -m pytest tests/test_oracle.py -k bridgejl   # compares tests/golden/julia_out/* with the oracle, ==
#
# Cases (the same definitions as tests/problems.py):
#   fhn_partialbridge_extreme / _first   project_partialbridge/partialbridge_fitzhugh.jl:31-50,88-101
#   ou_guidedbridge                      test/guip.jl:117-120,248
using Bridge, StaticArrays, LinearAlgebra, DelimitedFiles, Printf

const R1 = SVector{1,Float64}
const R2 = SVector{2,Float64}

struct FHN <: ContinuousTimeProcess{R2}
    eps::Float64; s::Float64; gamma::Float64; beta::Float64; sigma::Float64
end
Bridge.b(t, x, P::FHN) = R2((x[1] - x[2] - x[1]^3 + P.s) / P.eps, P.gamma * x[1] - x[2] + P.beta)
Bridge.σ(t, x, P::FHN) = R2(0.0, P.sigma)
Bridge.constdiff(::FHN) = true
struct FHNAuxEnd <: ContinuousTimeProcess{R2}
    eps::Float64; s::Float64; gamma::Float64; beta::Float64; sigma::Float64; v::Float64
end
Bridge.B(t, P::FHNAuxEnd) = @SMatrix [1 / P.eps - 3 * P.v^2 / P.eps  -1 / P.eps; P.gamma  -1.0]
Bridge.β(t, P::FHNAuxEnd) = R2(P.s / P.eps + 2 * P.v^3 / P.eps, P.beta)
Bridge.σ(t, P::FHNAuxEnd) = R2(0.0, P.sigma)
Bridge.a(t, P::FHNAuxEnd) = Bridge.σ(t, P) * Bridge.σ(t, P)'
Bridge.b(t, x, P::FHNAuxEnd) = Bridge.B(t, P) * x + Bridge.β(t, P)
Bridge.constdiff(::FHNAuxEnd) = true

hexf(x) = @sprintf("%a", x)     # exact (hexadecimal) floats: the comparison is bit for bit

function run_case(name, indir, outdir, build)
    tt = vec(readdlm(joinpath(indir, name * "_tt.csv"), ',', Float64))
    Wm = readdlm(joinpath(indir, name * "_W.csv"), ',', Float64)         # [npaths*N, m'] path-major
    N = length(tt)
    npaths = div(size(Wm, 1), N)
    Po, x0, wrap = build(tt)
    open(joinpath(outdir, name * "_X.csv"), "w") do fx
        open(joinpath(outdir, name * "_ll.csv"), "w") do fl
            for p in 1:npaths
                W = SamplePath(copy(tt), [wrap(Wm[(p-1)*N+i, :]) for i in 1:N])   # copy(tt): W.tt !== Po.tt (src/euler.jl:248)
                X = Bridge.samplepath(tt, zero(x0))
                solve!(Euler(), X, x0, W, Po)
                for i in 1:N
                    println(fx, join(hexf.(X.yy[i]), ","))
                end
                println(fl, hexf(llikelihood(LeftRule(), X, Po)))
            end
        end
    end
end

function main(indir, outdir)
    mkpath(outdir)
    open(joinpath(outdir, "VERSION.txt"), "w") do f
        println(f, string(VERSION))      # the test picks == (Julia <= 1.6) or the stated tolerance (>= 1.7) from this
    end
    fhn(v) = tt -> begin
        P = FHN(0.1, 0.0, 1.5, 0.8, 0.3)
        Pt = FHNAuxEnd(0.1, 0.0, 1.5, 0.8, 0.3, v)
        Bridge.PartialBridge(tt, P, Pt, (@SMatrix [1.0 0.0]), SVector(v), (@SMatrix [1e-10])), R2(-0.5, -0.6), w -> w[1]
    end
    run_case("fhn_partialbridge_extreme", indir, outdir, fhn(1.1))
    run_case("fhn_partialbridge_first", indir, outdir, fhn(-1.0))
    ou = tt -> begin
        P = LinPro(SMatrix{1,1}(-0.8), R1(0.0), SMatrix{1,1}(sqrt(0.7)))
        Pt = LinPro(SMatrix{1,1}(-0.8), R1(0.2), SMatrix{1,1}(sqrt(0.7)))
        GuidedBridge(tt, P, Pt, R1(0.1)), R1(0.5), w -> R1(w[1])
    end
    run_case("ou_guidedbridge", indir, outdir, ou)
end

main(ARGS[1], ARGS[2])
