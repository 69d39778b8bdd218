// bhip_girsanov_kernel.h -- girsanov(X, P, Pt) of src/diffusion.jl:109-123 for a stored ensemble.
//
//   som = sum_{i=1}^{N-1} dot( Gamma(t_i,x_i,P) * (B - Bt),  x_{i+1} - x_i - 0.5(B + Bt)(t_{i+1}-t_i) )
//   B = b(t_i,x_i,P), Bt = b(t_i,x_i,Pt)
//
// the log-likelihood ratio dP/dPt of a discretely stored path; the reference's parameter updates call
// it with two parameter sets of the SAME process type (example/fitzhugh_nagumo_full.jl:313-321:
// girsanov(BBall, P_theta_new, P_theta)), its test with Pt = Wiener (test/guip.jl:72).  One path per
// lane, X read once as a coalesced stream (8d B/path-step: HBM bound), next row prefetched.
// Gamma = inv(a) is constant for every built-in model (src/types.jl:33) and comes from the host.
#pragma once
#include "bhip_path_kernel.h"

namespace bhip {

struct GirsArgs {
    const double *rows;   // packed rows of a guide-free proposal (t, dt, 1/dt per step)
    int rs, N;
    long P;
    const double *X;      // [N][D][ldX]
    const double *X1, *X2;   // the ensemble in parts (bhip_girsanov_parts): paths [j*xpart, (j+1)*xpart) in buffer j; xpart = 0: one buffer
    long xpart;
    long ldX;
    double *out;          // [P]
    int zero_t;           // Pt = Wiener: Bt = 0
    double gam[9];        // Gamma(P), column-major D x D
    double mpar[40];      // device parameter block of P
    double mpar_t[40];    // ... of Pt (same model type)
};

template <class M>
__global__ __launch_bounds__(256) void k_girsanov(const GirsArgs a)
{
    constexpr int D = M::D;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.P) return;
    const M mp(a.mpar), mt(a.mpar_t);
    const int part = a.xpart ? (int)(p / a.xpart) : 0;
    const double *x = (part == 0 ? a.X : part == 1 ? a.X1 : a.X2) + (p - (long)part * a.xpart);
    double xc[D], xn[D], xnn[D];
#pragma unroll
    for (int k = 0; k < D; k++) xc[k] = ld_stream(x + (size_t)k * a.ldX);
#pragma unroll
    for (int k = 0; k < D; k++) xn[k] = ld_stream(x + (size_t)(D + k) * a.ldX);
    double som = 0.0;
    for (int i = 0; i < a.N - 1; i++) {
        const int i2 = i + 2 < a.N ? i + 2 : a.N - 1;
#pragma unroll
        for (int k = 0; k < D; k++) xnn[k] = ld_stream(x + ((size_t)i2 * D + k) * a.ldX);
        const double *row = a.rows + (size_t)i * a.rs;
        const double t = row[0], dt = row[1];
        double B[D], Bt[D], df[D], g[D];
        mp.b(t, xc, B);
        if (a.zero_t) {
#pragma unroll
            for (int k = 0; k < D; k++) Bt[k] = 0.0;
        } else {
            mt.b(t, xc, Bt);
        }
#pragma unroll
        for (int k = 0; k < D; k++) df[k] = B[k] - Bt[k];
#pragma unroll
        for (int r = 0; r < D; r++) {   // Gamma * (B - Bt)
            double s = a.gam[r] * df[0];
#pragma unroll
            for (int c = 1; c < D; c++) s += a.gam[r + D * c] * df[c];
            g[r] = s;
        }
        double dot = 0.0;
#pragma unroll
        for (int k = 0; k < D; k++) {
            const double inc = (xn[k] - xc[k]) - (0.5 * (B[k] + Bt[k])) * dt;
            dot = k == 0 ? g[0] * inc : dot + g[k] * inc;
        }
        som += dot;
#pragma unroll
        for (int k = 0; k < D; k++) { xc[k] = xn[k]; xn[k] = xnn[k]; }
    }
    a.out[p] = som;
}

using girsanov_fn = hipError_t (*)(const GirsArgs &, hipStream_t);

template <class M>
hipError_t launch_girsanov(const GirsArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_girsanov<M>, dim3((unsigned)((a.P + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace bhip
