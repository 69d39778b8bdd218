// bhip_guide_kernel.h -- guide pre-computation ON THE DEVICE, one guide per chain.
//
// Adaptive smoothing (supplements/smoothing/smoothing.jl:130-160) re-linearises every segment's auxiliary process around
// the running mean of THE CHAIN'S OWN paths (xx = mcstate[i][1]) and rebuilds the chain of GuidedBridge's backwards:
//
//     H, v = gpupdate(prior, last observation)                                    (host, once: shared by all chains)
//     for i = m .. 1:  linearappr!(Pt[i], Y_i, P)        B_j = bderiv(t_j, Y_j), b_j = b(t_j, Y_j), xx_j = Y_j   src/linpro.jl:196-204
//                      Po[i] = GuidedBridge(tt_i, P, Pt[i], v, H)     index-based Heun, src/guip.jl:181-189 (bhip_host.hpp: guide_hv_heuni)
//                      H, v = gpupdate(Po[i], L, Sigma, obs_i)        src/guip.jl:221-231
//     pi0 = Gaussian(v, Hermitian(H))
//
// With an ensemble every chain has its own means, hence its own guides: O(m N d^3) work and m N (d^2 + d + ...) doubles
// PER CHAIN -- the case where the host cannot keep up.  Here a lane owns a chain and walks a segment's grid backwards with
// (K, V) in registers; per grid point it evaluates the linearisation, does the two Heun steps and writes the chain's
// coefficient row in the layout the path kernels read (RowLayout minus the three shared time entries), SoA over chains:
//     prow[(i*PRL + q)*ld + p],   q: Hd_i (d*d), V_i (d), xx_i (d) -- compact: the path kernels derive B~_i = bderiv(xx_i),
//     beta~_i = b(xx_i) - B~_i xx_i, the cofactors and the determinant on the fly (bhip_smallmat.h: expand_pp_row)
// The operations and their order are those of bhip_host.hpp (Mat products accumulate left to right, StaticArrays closed
// forms for inv/det), so a chain's rows are bit-identical to what the host would compute for that chain's means.
#pragma once
#include "bhip_path_kernel.h"   // (brings bhip_smallmat.h)

namespace bhip {

// gpupdate(Hd, V, L, Sigma, v), finite branch  src/guip.jl:221-231 (bhip_host.hpp gpupdate).  Si = inv(Sigma) from the host.
template <int D, int MO>
BHIP_DEV void sm_gpupdate(const double *Hd, const double *V, const double *L, const double *Sigma, const double *Si, const double *v,
                          double *Hd_out, double *V_out)
{
    double LH[MO * D], S[MO * MO], Sinv[MO * MO], T1[D * MO], T2[D * MO], T3[D * D], Z[D * D], w1[D], w2[D];
    sm_mul<MO, D, D>(L, Hd, LH);            // L*Hd
    sm_mul_t<MO, D, MO>(LH, L, S);          // (L*Hd)*L'
#pragma unroll
    for (int k = 0; k < MO * MO; k++) S[k] = Sigma[k] + S[k];
    sm_inv<MO>(S, Sinv);
    sm_mul_t<D, D, MO>(Hd, L, T1);          // Hd*L'
    sm_mul<D, MO, MO>(T1, Sinv, T2);        // (Hd*L')*inv(S)
    sm_mul<D, MO, D>(T2, L, T3);            // (...)*L
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
        for (int i = 0; i < D; i++) Z[i + D * j] = (i == j ? 1.0 : 0.0) - T3[i + D * j];
    sm_mul<D, D, D>(Z, Hd, Hd_out);         // Z*Hd
    sm_mul_t<D, D, MO>(Hd_out, L, T1);      // (Z*Hd)*L'
    sm_mul<D, MO, MO>(T1, Si, T2);          // (...)*inv(Sigma)
    sm_mul<D, MO, 1>(T2, v, w1);
    sm_mul<D, D, 1>(Z, V, w2);
#pragma unroll
    for (int k = 0; k < D; k++) V_out[k] = w1[k] + w2[k];
}

// cholupper(Hermitian(A))': the lower factor from the UPPER triangle (StaticArrays cholesky closed forms; oracle bo_chol_lower)
template <int D>
BHIP_DEV void sm_chol_lower(const double *A, double *C)
{
#pragma unroll
    for (int k = 0; k < D * D; k++) C[k] = 0.0;
    if constexpr (D == 1) C[0] = sqrt(A[0]);
    else if constexpr (D == 2) {
        const double a = sqrt(A[0]), b = A[2] / a;
        C[0] = a; C[1] = b; C[3] = sqrt(A[3] - b * b);
    } else {
        const double a11 = sqrt(A[0]), a12 = A[3] / a11, a22 = sqrt(A[4] - a12 * a12);
        const double a13 = A[6] / a11, a23 = (A[7] - a12 * a13) / a22, a33 = sqrt(A[8] - a13 * a13 - a23 * a23);
        C[0] = a11; C[1] = a12; C[2] = a13; C[4] = a22; C[5] = a23; C[8] = a33;
    }
}

struct GArgs {
    long n, ld;
    int N, rs, hwindow;
    int lane_shift;        // chains per wave = 64 >> lane_shift (small ensembles: more, narrower waves)
    int lna;               // LinearNoiseAppr auxiliaries (src/guip.jl:114-146): B_j = 0, xx_j = 0, b_j = slope of the mean path at max(j, 1)
    const double *srows;   // the segment's shared rows: t_i, dt_i at [i*rs + 0], [i*rs + 1]
    const double *mean;    // [N][D][ld]: the chains' linearisation paths (running means, mcnext!)
    double *prow;          // out [N-1][PRL][ld]
    double *carry;         // [D*D + D][ld]: (Hdiamond, v) at the segment's right end in; after gpupdate at its left end out
    double *vend;          // out [D][ld]: V[N-1] of the chain (endpoint rule, src/euler.jl:241-242)
    unsigned char *uv;     // out [ld]: norm(Hd[N-1], 1) < eps()
    double L[9], Sigma[9], Si[9], obs[3];
    double mpar[40];
};

// per-chain rows: everything of RowLayout except (t, dt, sqrt(dt))

template <class M, int MO>
__global__ __launch_bounds__(64) void k_seg_guide(const GArgs g)
{
    constexpr int D = M::D, MP = M::MP, DD = D * D, PRL = pp_row_len<D>();
    // a lane walks its chain's grid alone and a wave is bound by the latency of that walk, not by issue slots: ensembles too
    // small to give every SIMD a wave of 64 chains run 32 or 16 chains per wave (the other lanes leave) -- twice / four times
    // the SIMDs at work for the same time per wave
    const int lanes = (int)blockDim.x >> g.lane_shift;
    if ((int)threadIdx.x >= lanes) return;
    const long p = (long)blockIdx.x * lanes + threadIdx.x;
    if (p >= g.n) return;
    const M model(g.mpar);
    const int N = g.N;
    const cptr_t srows = (cptr_t)(uintptr_t)g.srows;

    // outer(Sigma_j): sigma is constant for the processes with a bderiv; S*S' through the model's own sigma*dw
    double aS[DD];
    {
        double Sg[D * MP];
#pragma unroll
        for (int c = 0; c < MP; c++) {
            double e[MP], col[D], zero[D];
#pragma unroll
            for (int k = 0; k < MP; k++) e[k] = k == c ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < D; k++) zero[k] = 0.0;
            model.sdw(0.0, zero, e, col);
#pragma unroll
            for (int r = 0; r < D; r++) Sg[r + D * c] = col[r];
        }
        sm_mul_t<D, MP, D>(Sg, Sg, aS);
    }

    // the linearisation point at grid index j: the chain's mean, or its moving average (smoothing.jl:136,142)
    auto loadY = [&](int j, double *y) {
        if (g.hwindow <= 0) {
#pragma unroll
            for (int k = 0; k < D; k++) y[k] = g.mean[((size_t)j * D + k) * g.ld + p];
        } else {
            const int lo = max(0, j - g.hwindow), hi = min(N - 1, j + g.hwindow);
#pragma unroll
            for (int k = 0; k < D; k++) {
                double s = g.mean[((size_t)lo * D + k) * g.ld + p];
                for (int l = lo + 1; l <= hi; l++) s += g.mean[((size_t)l * D + k) * g.ld + p];
                y[k] = s / (double)(hi - lo + 1);
            }
        }
    };
    auto fH = [&](const double *B, const double *K, double *out) {   // B K + K B' - outer(Sigma)
        double BK[DD], KBt[DD];
        sm_mul<D, D, D>(B, K, BK);
        sm_mul_t<D, D, D>(K, B, KBt);
#pragma unroll
        for (int k = 0; k < DD; k++) out[k] = (BK[k] + KBt[k]) - aS[k];
    };
    auto fV = [&](const double *B, const double *xx, const double *b, const double *x, double *out) {   // B (x - xx) + b
        double xm[D], Bx[D];
#pragma unroll
        for (int k = 0; k < D; k++) xm[k] = x[k] - xx[k];
        sm_mul<D, D, 1>(B, xm, Bx);
#pragma unroll
        for (int k = 0; k < D; k++) out[k] = Bx[k] + b[k];
    };

    double K[DD], w[D];
#pragma unroll
    for (int k = 0; k < DD; k++) K[k] = g.carry[(size_t)k * g.ld + p];
#pragma unroll
    for (int k = 0; k < D; k++) w[k] = g.carry[(size_t)(DD + k) * g.ld + p];
    {   // endpoint(y, P::GuidedBridge): norm(Hd[end], 1) < eps() ? V[end] : y
        double n1 = 0.0;
#pragma unroll
        for (int k = 0; k < DD; k++) n1 += fabs(K[k]);
        g.uv[p] = n1 < 2.220446049250313e-16 ? 1 : 0;
#pragma unroll
        for (int k = 0; k < D; k++) g.vend[(size_t)k * g.ld + p] = w[k];
    }
    // Without the moving average the means are read THREE grid points ahead of their use (round 3): a lane walks its chain's grid
    // alone -- one wave per SIMD at 32 768 chains -- and the load of the next point's mean used to be exposed once per point
    // (~2 us of the ~3 us a point took).  Registers: yn = y_{j+1} (kept for the LinearNoiseAppr slope at j = 0), yc = y_j,
    // m1 = y_{j-1}, m2 = y_{j-2}, each clamped at index 0.
    const bool ring = g.hwindow <= 0;
    double yn[D], yc[D], m1[D], m2[D];
    auto loadRaw = [&](int j, double *y) {
#pragma unroll
        for (int k = 0; k < D; k++) y[k] = g.mean[((size_t)max(j, 0) * D + k) * g.ld + p];
    };
    if (ring) { loadRaw(N - 1, yc); loadRaw(N - 2, m1); loadRaw(N - 3, m2); loadRaw(N - 1, yn); }
    auto advance = [&](int j /* the index yc moves to */) {
#pragma unroll
        for (int k = 0; k < D; k++) { yn[k] = yc[k]; yc[k] = m1[k]; m1[k] = m2[k]; }
        loadRaw(j - 2, m2);
    };
    // the linearisation at grid index j: LinearAppr (linearappr!, src/linpro.jl:196-204) or LinearNoiseAppr (Pt.Y.yy[:] = xx)
    auto linearise = [&](int j, double *B, double *b, double *xx) {
        if (g.lna) {
            const int jj = max(j, 1);
            double ya[D], yb[D];
            if (ring) {
#pragma unroll
                for (int k = 0; k < D; k++) { ya[k] = j >= 1 ? yc[k] : yn[k]; yb[k] = j >= 1 ? m1[k] : yc[k]; }
            } else { loadY(jj, ya); loadY(jj - 1, yb); }
            const double h = srows[(size_t)(jj - 1) * g.rs + 1];   // tt[jj] - tt[jj-1]
#pragma unroll
            for (int k = 0; k < D; k++) { b[k] = (ya[k] - yb[k]) / h; xx[k] = 0.0; }
#pragma unroll
            for (int k = 0; k < DD; k++) B[k] = 0.0;
        } else {
            if (ring) {
#pragma unroll
                for (int k = 0; k < D; k++) xx[k] = yc[k];
            } else loadY(j, xx);
            model.bderiv(0.0, xx, B);   // (bderiv and b of these processes do not depend on t)
            model.b(0.0, xx, b);
        }
    };
    double B1[DD], b1[D], x1[D];
    linearise(N - 1, B1, b1, x1);
    double dtn = -srows[(size_t)(N - 2) * g.rs + 1];   // tt[i] - tt[i+1] = -(tt[i+1] - tt[i]) exactly; read one point ahead as well
    for (int i = N - 2; i >= 0; i--) {
        double B0[DD], b0[D], x0[D];
        if (ring) advance(i);
        linearise(i, B0, b0, x0);
        const double dt = dtn;
        dtn = -srows[(size_t)max(i - 1, 0) * g.rs + 1];
        {
            double k1[DD], k2[DD], yp[DD];
            fH(B0, K, k1);
#pragma unroll
            for (int k = 0; k < DD; k++) yp[k] = K[k] + dt * k1[k];
            fH(B1, yp, k2);
#pragma unroll
            for (int k = 0; k < DD; k++) K[k] = K[k] + (dt / 2) * (k1[k] + k2[k]);
        }
        {
            double k1[D], k2[D], wp[D];
            fV(B0, x0, b0, w, k1);
#pragma unroll
            for (int k = 0; k < D; k++) wp[k] = w[k] + dt * k1[k];
            fV(B1, x1, b1, wp, k2);
#pragma unroll
            for (int k = 0; k < D; k++) w[k] = w[k] + (dt / 2) * (k1[k] + k2[k]);
        }
        // the chain's COMPACT row of step i: Hd_i, V_i and the linearisation datum (xx_i, or the slope b_i of a LinearNoiseAppr);
        // the path kernels expand it (expand_pp_row, bhip_smallmat.h) into B~_i, beta~_i, cofactors, det
        double row[PRL];
#pragma unroll
        for (int k = 0; k < DD; k++) row[k] = K[k];
#pragma unroll
        for (int k = 0; k < D; k++) { row[DD + k] = w[k]; row[DD + D + k] = g.lna ? b0[k] : x0[k]; }
        double *o = g.prow + (size_t)i * PRL * g.ld + p;
#pragma unroll
        for (int q = 0; q < PRL; q++) o[(size_t)q * g.ld] = row[q];
#pragma unroll
        for (int k = 0; k < DD; k++) B1[k] = B0[k];
#pragma unroll
        for (int k = 0; k < D; k++) { b1[k] = b0[k]; x1[k] = x0[k]; }
    }
    // fold the observation at the segment's left end: the right-end condition of the previous segment (or pi0)
    double Hn[DD], vn[D];
    sm_gpupdate<D, MO>(K, w, g.L, g.Sigma, g.Si, g.obs, Hn, vn);
#pragma unroll
    for (int k = 0; k < DD; k++) g.carry[(size_t)k * g.ld + p] = Hn[k];
#pragma unroll
    for (int k = 0; k < D; k++) g.carry[(size_t)(DD + k) * g.ld + p] = vn[k];
}

// pi0 = Gaussian(v, Hermitian(H)) per chain from the carry: mu = v, C = cholupper(Hermitian(H))'
template <int D>
__global__ void k_seg_pi0(long n, long ld, const double *__restrict__ carry, double *__restrict__ mu, double *__restrict__ chol)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double H[D * D], C[D * D];
#pragma unroll
    for (int k = 0; k < D * D; k++) H[k] = carry[(size_t)k * ld + p];
    sm_chol_lower<D>(H, C);
#pragma unroll
    for (int k = 0; k < D * D; k++) chol[(size_t)k * ld + p] = C[k];
#pragma unroll
    for (int k = 0; k < D; k++) mu[(size_t)k * ld + p] = carry[(size_t)(D * D + k) * ld + p];
}

#ifndef BHIP_GUIDE_N32
#define BHIP_GUIDE_N32 24576   // from this many chains on: 32 chains per wave (below: 16)
#endif
typedef hipError_t (*guide_launch_fn)(const GArgs &, hipStream_t);
template <class M, int MO>
hipError_t launch_seg_guide(const GArgs &g, hipStream_t st)
{
    GArgs a = g;
    a.lane_shift = g.n >= 65536 ? 0 : g.n >= BHIP_GUIDE_N32 ? 1 : 2;   // 64, 32 or 16 chains per wave: >= ~1024 waves where the ensemble allows
    const long per = 64 >> a.lane_shift;
    hipLaunchKernelGGL((k_seg_guide<M, MO>), dim3((unsigned)((g.n + per - 1) / per)), dim3(64), 0, st, a);
    return hipGetLastError();
}
template <class M>
guide_launch_fn get_guide_launch(int mo)
{
    if (mo == 1) return launch_seg_guide<M, 1>;
    if constexpr (M::D >= 2) { if (mo == 2) return launch_seg_guide<M, 2>; }
    if constexpr (M::D >= 3) { if (mo == 3) return launch_seg_guide<M, 3>; }
    return nullptr;
}

}  // namespace bhip
