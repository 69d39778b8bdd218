// bhip_smallmat.h -- small static matrices (d <= 3), column-major, with the operation order of bhip_host.hpp's Mat operators
// (products accumulate left to right, StaticArrays closed forms for det / inv): device-side twins used by the guide kernel
// (bhip_guide_kernel.h) and by the path kernels when they expand a chain's compact guide row.
#pragma once
#include "bhip_models.h"

namespace bhip {

// ---- small static matrices, column-major; the operation order of bhip_host.hpp's Mat operators
template <int R, int K, int C>
BHIP_DEV void sm_mul(const double *A, const double *B, double *O)   // O(RxC) = A(RxK) * B(KxC)
{
#pragma unroll
    for (int j = 0; j < C; j++)
#pragma unroll
        for (int i = 0; i < R; i++) {
            double s = A[i] * B[K * j];
#pragma unroll
            for (int l = 1; l < K; l++) s += A[i + R * l] * B[l + K * j];
            O[i + R * j] = s;
        }
}
template <int R, int K, int C>
BHIP_DEV void sm_mul_t(const double *A, const double *B, double *O)   // O(RxC) = A(RxK) * B'(KxC), B stored CxK
{
#pragma unroll
    for (int j = 0; j < C; j++)
#pragma unroll
        for (int i = 0; i < R; i++) {
            double s = A[i] * B[j];
#pragma unroll
            for (int l = 1; l < K; l++) s += A[i + R * l] * B[j + C * l];
            O[i + R * j] = s;
        }
}
template <int N>
BHIP_DEV double sm_det(const double *a)   // StaticArrays det.jl
{
    if constexpr (N == 1) return a[0];
    else if constexpr (N == 2) return a[0] * a[3] - a[2] * a[1];
    else {
        const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
        return a[0] * c0 + a[1] * c1 + a[2] * c2;
    }
}
template <int N>
BHIP_DEV void sm_inv(const double *a, double *R)   // StaticArrays inv.jl
{
    if constexpr (N == 1) R[0] = 1.0 / a[0];
    else if constexpr (N == 2) {
        const double d = sm_det<2>(a);
        R[0] = a[3] / d; R[1] = -(a[1] / d); R[2] = -(a[2] / d); R[3] = a[0] / d;
    } else {
        double x0[3] = {a[0], a[1], a[2]};
        const double x1[3] = {a[3], a[4], a[5]}, x2[3] = {a[6], a[7], a[8]};
        double y0[3] = {x1[1] * x2[2] - x1[2] * x2[1], x1[2] * x2[0] - x1[0] * x2[2], x1[0] * x2[1] - x1[1] * x2[0]};
        const double d = x0[0] * y0[0] + x0[1] * y0[1] + x0[2] * y0[2];
#pragma unroll
        for (int k = 0; k < 3; k++) { x0[k] = x0[k] / d; y0[k] = y0[k] / d; }
        const double y1[3] = {x2[1] * x0[2] - x2[2] * x0[1], x2[2] * x0[0] - x2[0] * x0[2], x2[0] * x0[1] - x2[1] * x0[0]};
        const double y2[3] = {x0[1] * x1[2] - x0[2] * x1[1], x0[2] * x1[0] - x0[0] * x1[2], x0[0] * x1[1] - x0[1] * x1[0]};
        R[0] = y0[0]; R[1] = y1[0]; R[2] = y2[0]; R[3] = y0[1]; R[4] = y1[1]; R[5] = y2[1];
        R[6] = y0[2]; R[7] = y1[2]; R[8] = y2[2];
    }
}

// The divisions of the (Hdiamond, V) guide -- r = Hd_i \ (V_i - x), src/guip.jl:192-193: (V - x)/Hd for d = 1, cofactor
// products over det(Hd_i) for d = 2, 3 (StaticArrays' closed forms) -- all divide by a value that belongs to the ROW, not to the
// path.  The correctly rounded fp64 division the compiler emits is: reciprocal seed, two Newton steps (divisor only), then
// q0 = a*y, rem = fma(-c, q0, a), q = fma(rem, y, q0).  The divisor-only part is done once per row -- on the host for shared
// rows (1.0/c, the correctly rounded reciprocal the Newton steps converge to), here for per-chain rows -- and the step keeps
// the three operations that depend on the path: the same bits as `a / c` (what the oracle computes) for 2^-200 < |c| < 2^200
// (checked on the host for shared rows; UniformDivisor in bhip_models.h is the same construction for model constants) AND a
// numerator whose quotient and remainder stay normal: 2^-700 < |a| < 2^700 is ample (q0 = a*y then lies within 2^+-900 and the
// remainder, ~2^-53 |a|, above 2^-753).  Outside that numerator range -- a path that has left every physical scale -- the
// hardware's pre-scaled division and these three operations may differ in the last bit or in how they overflow; the parity
// contract is stated for states inside it (the path kernels never produce such numerators from finite, moderate inputs).
BHIP_DEV double sm_recip(double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double y0 = __builtin_amdgcn_rcp(c);
    const double e0 = __builtin_fma(-c, y0, 1.0);
    const double y1 = __builtin_fma(y0, e0, y0);
    const double e1 = __builtin_fma(-c, y1, 1.0);
    return __builtin_fma(y1, e1, y1);
#else
    return 1.0 / c;
#endif
}
BHIP_DEV double sm_div_by(double a, double c, double y /* = sm_recip(c) */)
{
    const double q0 = a * y;
    const double rem = __builtin_fma(-c, q0, a);
    return __builtin_fma(rem, y, q0);
}

// the guide part of a coefficient row for the (Hdiamond, V) guide: bhip_host.hpp pack_rows
template <int D>
BHIP_DEV void hv_row_part(const double *A, const double *V, double *q)
{
    if constexpr (D == 1) { q[0] = A[0]; q[1] = V[0]; q[2] = sm_recip(A[0]); }
    else if constexpr (D == 2) { q[0] = A[0]; q[1] = A[1]; q[2] = A[2]; q[3] = A[3]; q[4] = sm_det<2>(A); q[5] = V[0]; q[6] = V[1]; q[7] = sm_recip(q[4]); }
    else {
        auto a = [&](int i, int j) { return A[(i - 1) + 3 * (j - 1)]; };
        q[0] = a(2, 2) * a(3, 3) - a(2, 3) * a(3, 2); q[1] = a(1, 3) * a(3, 2) - a(1, 2) * a(3, 3); q[2] = a(1, 2) * a(2, 3) - a(1, 3) * a(2, 2);
        q[3] = a(2, 3) * a(3, 1) - a(2, 1) * a(3, 3); q[4] = a(1, 1) * a(3, 3) - a(1, 3) * a(3, 1); q[5] = a(1, 3) * a(2, 1) - a(1, 1) * a(2, 3);
        q[6] = a(2, 1) * a(3, 2) - a(2, 2) * a(3, 1); q[7] = a(1, 2) * a(3, 1) - a(1, 1) * a(3, 2); q[8] = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
        q[9] = sm_det<3>(A); q[10] = V[0]; q[11] = V[1]; q[12] = V[2]; q[13] = sm_recip(q[9]);
    }
}


// per-chain guide rows are stored COMPACT: Hd_i (d*d), V_i (d) and the linearisation datum of grid index i (d):
//   LinearAppr:       xx_i  -> B~_i = bderiv(xx_i), beta~_i = b(xx_i) - B~_i xx_i        src/linpro.jl:187-204
//   LinearNoiseAppr:  the slope b_i -> B~_i = 0, beta~_i = b_i                           src/guip.jl:142-145
// instead of the 25 (d = 3) expanded entries the kernels' row layout has: 120 instead of 200 bytes per chain and step.
template <int D> constexpr int pp_row_len() { return D * D + 2 * D; }

// expand a compact row into the (B~, beta~, guide part) entries of RowLayout<HV>, i.e. entries 3.. of a coefficient row;
// the same operations, in the same order, as pack_rows / bhip_proposal_set_aux_linearappr perform on the host
template <class M>
BHIP_DEV void expand_pp_row(const M &model, int lna, const double *c, double *e)
{
    constexpr int D = M::D, DD = D * D;
    const double *Hd = c, *V = c + DD, *aux = c + DD + D;
    if (lna) {
#pragma unroll
        for (int k = 0; k < DD; k++) e[k] = 0.0;
#pragma unroll
        for (int k = 0; k < D; k++) e[DD + k] = aux[k] - 0.0;    // beta = b_i - B_i*xx_i with B_i = 0, xx_i = 0: b_i - 0
    } else {
        double b[D], Bx[D];
        model.bderiv(0.0, aux, e);
        model.b(0.0, aux, b);
        sm_mul<D, D, 1>(e, aux, Bx);
#pragma unroll
        for (int k = 0; k < D; k++) e[DD + k] = b[k] - Bx[k];
    }
    hv_row_part<D>(Hd, V, e + DD + D);
}

}  // namespace bhip
