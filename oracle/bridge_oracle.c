/*
 * bridge_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see bridge_oracle.h)
 *
 * Plain-C restatement of Bridge.jl's guided-proposal hot path.  Single-threaded scalar code in
 * the reference's operation order; every function cites the reference file:line it follows
 * (paths relative to /root/reference).  Nothing here is used by the product library.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include "bridge_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* The Box-Muller helpers use explicit fma(); build an FMA3 clone so the cpu_baseline timing is
 * not dominated by libm's software fma on hosts that have the instruction.  Both clones are
 * bit-identical (fma is correctly rounded either way; implicit contraction is disabled). */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(BO_NO_CLONES)
#define BO_CLONES __attribute__((target_clones("fma", "default")))
#else
#define BO_CLONES
#endif

#define D2 (BO_MAXD * BO_MAXD)

/* ------------------------------------------------------------------------------------------
 * small dense linear algebra, column-major, StaticArrays operation order:
 *   (A*x)[i] = A[i,1]*x[1] + A[i,2]*x[2] + ...   (left-to-right, first product starts the sum)
 * ------------------------------------------------------------------------------------------ */
static void mv(int r, int c, const double *A, const double *x, double *y)
{
    for (int i = 0; i < r; i++) {
        double s = A[i] * x[0];
        for (int j = 1; j < c; j++) s += A[i + r * j] * x[j];
        y[i] = s;
    }
}
/* C(r x c) = A(r x k) * B(k x c) */
static void mm(int r, int k, int c, const double *A, const double *B, double *C)
{
    for (int j = 0; j < c; j++)
        for (int i = 0; i < r; i++) {
            double s = A[i] * B[k * j];
            for (int l = 1; l < k; l++) s += A[i + r * l] * B[l + k * j];
            C[i + r * j] = s;
        }
}
/* C(r x c) = A(r x k) * B'(k x c),  B is c x k */
static void mmt(int r, int k, int c, const double *A, const double *B, double *C)
{
    for (int j = 0; j < c; j++)
        for (int i = 0; i < r; i++) {
            double s = A[i] * B[j];
            for (int l = 1; l < k; l++) s += A[i + r * l] * B[j + c * l];
            C[i + r * j] = s;
        }
}
/* C(r x c) = A'(r x k) * B(k x c),  A is k x r */
static void mtm(int r, int k, int c, const double *A, const double *B, double *C)
{
    for (int j = 0; j < c; j++)
        for (int i = 0; i < r; i++) {
            double s = A[k * i] * B[k * j];
            for (int l = 1; l < k; l++) s += A[l + k * i] * B[l + k * j];
            C[i + r * j] = s;
        }
}
static double dotv(int n, const double *a, const double *b)
{
    double s = a[0] * b[0];
    for (int i = 1; i < n; i++) s += a[i] * b[i];
    return s;
}

/* LU with partial pivoting (generic n > 3; LinearAlgebra.lu semantics, unpinned op order) */
static int lu_factor(int n, double *A, int *piv)
{
    for (int k = 0; k < n; k++) {
        int p = k;
        double best = fabs(A[k + n * k]);
        for (int i = k + 1; i < n; i++)
            if (fabs(A[i + n * k]) > best) { best = fabs(A[i + n * k]); p = i; }
        piv[k] = p;
        if (best == 0.0) return -1;
        if (p != k)
            for (int j = 0; j < n; j++) { double t = A[k + n * j]; A[k + n * j] = A[p + n * j]; A[p + n * j] = t; }
        double inv = 1.0 / A[k + n * k];
        for (int i = k + 1; i < n; i++) A[i + n * k] *= inv;
        for (int j = k + 1; j < n; j++) {
            double akj = A[k + n * j];
            for (int i = k + 1; i < n; i++) A[i + n * j] -= A[i + n * k] * akj;
        }
    }
    return 0;
}
static void lu_solve(int n, const double *LU, const int *piv, double *b)
{
    for (int k = 0; k < n; k++) { int p = piv[k]; if (p != k) { double t = b[k]; b[k] = b[p]; b[p] = t; } }
    for (int k = 0; k < n; k++) for (int i = k + 1; i < n; i++) b[i] -= LU[i + n * k] * b[k];
    for (int k = n - 1; k >= 0; k--) { b[k] /= LU[k + n * k]; for (int i = 0; i < k; i++) b[i] -= LU[i + n * k] * b[k]; }
}

/* StaticArrays det.jl: 1x1, 2x2 (A[1]*A[4] - A[3]*A[2]), 3x3 (dot(x0, cross(x1,x2))); else LU */
double bo_det(int n, const double *A)
{
    if (n == 1) return A[0];
    if (n == 2) return A[0] * A[3] - A[2] * A[1];
    if (n == 3) {
        double c0 = A[4] * A[8] - A[5] * A[7];
        double c1 = A[5] * A[6] - A[3] * A[8];
        double c2 = A[3] * A[7] - A[4] * A[6];
        return A[0] * c0 + A[1] * c1 + A[2] * c2;
    }
    double *T = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    memcpy(T, A, sizeof(double) * n * n);
    double d = 1.0;
    if (lu_factor(n, T, piv) != 0) d = 0.0;
    else for (int k = 0; k < n; k++) { d *= T[k + n * k]; if (piv[k] != k) d = -d; }
    free(T); free(piv);
    return d;
}

/* StaticArrays inv.jl (v1.x): 1x1 inv(a); 2x2 adjugate/det; 3x3 cross-product form; else LU */
int bo_inv(int n, const double *A, double *Ai)
{
    if (n == 1) { Ai[0] = 1.0 / A[0]; return 0; }
    if (n == 2) {
        double d = bo_det(2, A);
        double a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3];
        Ai[0] = a3 / d; Ai[1] = -(a1 / d); Ai[2] = -(a2 / d); Ai[3] = a0 / d;
        return 0;
    }
    if (n == 3) {
        double x0[3] = {A[0], A[1], A[2]}, x1[3] = {A[3], A[4], A[5]}, x2[3] = {A[6], A[7], A[8]};
        double y0[3] = {x1[1] * x2[2] - x1[2] * x2[1], x1[2] * x2[0] - x1[0] * x2[2], x1[0] * x2[1] - x1[1] * x2[0]};
        double d = x0[0] * y0[0] + x0[1] * y0[1] + x0[2] * y0[2];
        for (int k = 0; k < 3; k++) { x0[k] = x0[k] / d; y0[k] = y0[k] / d; }
        double y1[3] = {x2[1] * x0[2] - x2[2] * x0[1], x2[2] * x0[0] - x2[0] * x0[2], x2[0] * x0[1] - x2[1] * x0[0]};
        double y2[3] = {x0[1] * x1[2] - x0[2] * x1[1], x0[2] * x1[0] - x0[0] * x1[2], x0[0] * x1[1] - x0[1] * x1[0]};
        Ai[0] = y0[0]; Ai[1] = y1[0]; Ai[2] = y2[0];
        Ai[3] = y0[1]; Ai[4] = y1[1]; Ai[5] = y2[1];
        Ai[6] = y0[2]; Ai[7] = y1[2]; Ai[8] = y2[2];
        return 0;
    }
    double *T = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    memcpy(T, A, sizeof(double) * n * n);
    int rc = lu_factor(n, T, piv);
    if (rc == 0)
        for (int j = 0; j < n; j++) {
            double *col = Ai + n * j;
            for (int i = 0; i < n; i++) col[i] = (i == j) ? 1.0 : 0.0;
            lu_solve(n, T, piv, col);
        }
    free(T); free(piv);
    return rc;
}

/* StaticArrays solve.jl: 1x1 b/a; 2x2, 3x3 Cramer with the published operation order; else LU */
int bo_solve(int n, const double *a, const double *b, double *x)
{
#define AA(i, j) a[(i - 1) + n * (j - 1)]
    if (n == 1) { x[0] = b[0] / a[0]; return 0; }
    if (n == 2) {
        double d = bo_det(2, a);
        double x0 = (AA(2, 2) * b[0] - AA(1, 2) * b[1]) / d;
        double x1 = (AA(1, 1) * b[1] - AA(2, 1) * b[0]) / d;
        x[0] = x0; x[1] = x1;
        return 0;
    }
    if (n == 3) {
        double d = bo_det(3, a);
        double x0 = ((AA(2, 2) * AA(3, 3) - AA(2, 3) * AA(3, 2)) * b[0] + (AA(1, 3) * AA(3, 2) - AA(1, 2) * AA(3, 3)) * b[1] +
                     (AA(1, 2) * AA(2, 3) - AA(1, 3) * AA(2, 2)) * b[2]) / d;
        double x1 = ((AA(2, 3) * AA(3, 1) - AA(2, 1) * AA(3, 3)) * b[0] + (AA(1, 1) * AA(3, 3) - AA(1, 3) * AA(3, 1)) * b[1] +
                     (AA(1, 3) * AA(2, 1) - AA(1, 1) * AA(2, 3)) * b[2]) / d;
        double x2 = ((AA(2, 1) * AA(3, 2) - AA(2, 2) * AA(3, 1)) * b[0] + (AA(1, 2) * AA(3, 1) - AA(1, 1) * AA(3, 2)) * b[1] +
                     (AA(1, 1) * AA(2, 2) - AA(1, 2) * AA(2, 1)) * b[2]) / d;
        x[0] = x0; x[1] = x1; x[2] = x2;
        return 0;
    }
#undef AA
    double *T = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    memcpy(T, a, sizeof(double) * n * n);
    int rc = lu_factor(n, T, piv);
    if (rc == 0) { for (int i = 0; i < n; i++) x[i] = b[i]; lu_solve(n, T, piv, x); }
    free(T); free(piv);
    return rc;
}

/* src/gaussian.jl:66-75  logpdf of a centred Gaussian:  S = chol(Sigma).L;
 *   -((norm(S\x))^2 + 2*sumlogdiag(S) + d*log(2pi))/2 ; scalar: -(x^2/S + log(S) + log(2pi))/2 */
double bo_logpdfnormal(int d, const double *x, const double *Sigma)
{
    const double log2pi = log(2 * M_PI);
    if (d == 1) return -(x[0] * x[0] / Sigma[0] + log(Sigma[0]) + log2pi) / 2;
    double S[D2], y[BO_MAXD];
    memset(S, 0, sizeof(double) * d * d);
    for (int j = 0; j < d; j++) { /* lower Cholesky, column by column */
        double s = Sigma[j + d * j];
        for (int k = 0; k < j; k++) s -= S[j + d * k] * S[j + d * k];
        S[j + d * j] = sqrt(s);
        for (int i = j + 1; i < d; i++) {
            double t = Sigma[i + d * j];
            for (int k = 0; k < j; k++) t -= S[i + d * k] * S[j + d * k];
            S[i + d * j] = t / S[j + d * j];
        }
    }
    for (int i = 0; i < d; i++) { /* forward substitution S\x */
        double t = x[i];
        for (int k = 0; k < i; k++) t -= S[i + d * k] * y[k];
        y[i] = t / S[i + d * i];
    }
    double n2 = 0, sl = 0;
    for (int i = 0; i < d; i++) { n2 += y[i] * y[i]; sl += log(S[i + d * i]); }
    double nrm = sqrt(n2);
    return -(nrm * nrm + 2 * sl + d * log2pi) / 2;
}

/* ------------------------------------------------------------------------------------------
 * RNG, specifications "bhip-philox-v4" (default), "-v3" and "-v2" (DESIGN.md section 4).  The reference draws from Julia's global
 * randn (src/wiener.jl:31,44,55) which cannot be reproduced outside Julia; this is
 * the counter-based replacement shared, by specification, with the HIP kernels.
 * ------------------------------------------------------------------------------------------ */
void bo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Constant tables of the specification (data; generated by scripts/gen_rng_tables.py, every entry correctly rounded):
 *   BO_LOGTAB[k] = {A_k, B_k}, k = 0..128;  BO_SCTAB[j] = {cos(2 pi j/32), sin(2 pi j/32)}, j = 0..31 */
#include "bo_rng_tables.h"
static const double BO_LOGTAB[2 * BO_LOGTAB_N] = BO_LOGTAB_INIT;
static const double BO_SCTAB[2 * BO_SCTAB_N] = BO_SCTAB_INIT;

/* L(x) = -2 ln x for x in (0,1], from integer operations, +, *, fma and the table only (bit-identical on CPU and
 * GPU).  x = 2^e m0 with m0 in [1,2); row k = round(128 m0) - 128; rows k > 53 treat m0/2 (exponent e + 1).
 * s = m0 A_k + 2 (one fma) is -2 (m/c_k - 1), and -2 ln(1 - s/2) = s + s^2/4 + s^3/12 + ... + s^7/448. */
BO_CLONES double bo_m2log(double x)
{
    union { double d; uint64_t u; } v; v.d = x;
    uint32_t top = (uint32_t)(v.u >> 32);
    uint32_t k = (((top >> 12) & 255u) + 1u) / 2u;
    int e = (int)(top >> 20) - 1023;
    if (k > 53u) e += 1;
    v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double s = fma(v.d, BO_LOGTAB[2 * k], 2.0);
    double q = 1.0 / 448.0;
    q = fma(q, s, 1.0 / 192.0);
    q = fma(q, s, 1.0 / 80.0);
    q = fma(q, s, 1.0 / 32.0);
    q = fma(q, s, 1.0 / 12.0);
    q = fma(q, s, 1.0 / 4.0);
    double series = fma(s * s, q, s);
    double de = (double)e;
    /* -2 ln2 = hi + lo, hi with 32 significant bits so that de*hi is exact */
    double hi = -2.0 * 6.93147180369123816490e-01, lo = -2.0 * 1.90821492927058770002e-10;
    return fma(de, hi, fma(de, lo, BO_LOGTAB[2 * k + 1] + series));
}
BO_CLONES double bo_log(double x) { return -0.5 * bo_m2log(x); }

/* sin/cos(2 pi u) for u = K 2^-53 in [0,1), `w` = the upper 32 of the 64 source bits: jr = round(32 u) from the
 * top six bits, f = u - jr/32 exactly, x = 2 pi f, Taylor sine / cosine of x (to x^9 / x^10), then the rotation by
 * row jr mod 32 of the table. */
BO_CLONES void bo_sincos2pi(double u, uint32_t w, double *sn, double *cs)
{
    uint32_t jr = ((w >> 26) + 1u) / 2u;
    double f = fma((double)jr, -1.0 / 32.0, u);
    double ck = BO_SCTAB[2 * (jr % 32u)], sk = BO_SCTAB[2 * (jr % 32u) + 1];
    double x = f * 6.283185307179586;
    double z = x * x;
    double ps = 1.0 / 362880.0;                  /*  1/9!  */
    ps = fma(ps, z, -1.0 / 5040.0);              /* -1/7!  */
    ps = fma(ps, z, 1.0 / 120.0);                /*  1/5!  */
    ps = fma(ps, z, -1.0 / 6.0);                 /* -1/3!  */
    double sf = fma(x * z, ps, x);
    double pc = -1.0 / 3628800.0;                /* -1/10! */
    pc = fma(pc, z, 1.0 / 40320.0);              /*  1/8!  */
    pc = fma(pc, z, -1.0 / 720.0);               /* -1/6!  */
    pc = fma(pc, z, 1.0 / 24.0);                 /*  1/4!  */
    pc = fma(pc, z, -0.5);                       /* -1/2!  */
    double cf = fma(z, pc, 1.0);
    *cs = fma(-sk, sf, ck * cf);
    *sn = fma(ck, sf, sk * cf);
}

/* sin/cos(2 pi u) for u = K24 2^-24 (24 bits of angle): jr = round(32 u) from the top six bits, f = u - jr/32 =
 * (K24 - jr 2^19) 2^-24 exactly, x = fl(2 pi) f; then as bo_sincos2pi. */
BO_CLONES void bo_sincos2pi_k24(uint32_t k24, double *sn, double *cs)
{
    uint32_t jr = ((k24 >> 18) + 1u) / 2u;
    int32_t m = (int32_t)k24 - (int32_t)(jr << 19);
    double f = (double)m * 0x1.0p-24;                    /* = u - jr/32, exact */
    double ck = BO_SCTAB[2 * (jr % 32u)], sk = BO_SCTAB[2 * (jr % 32u) + 1];
    double x = f * 6.283185307179586;
    double z = x * x;
    double ps = 1.0 / 362880.0;
    ps = fma(ps, z, -1.0 / 5040.0);
    ps = fma(ps, z, 1.0 / 120.0);
    ps = fma(ps, z, -1.0 / 6.0);
    double sf = fma(x * z, ps, x);
    double pc = -1.0 / 3628800.0;
    pc = fma(pc, z, 1.0 / 40320.0);
    pc = fma(pc, z, -1.0 / 720.0);
    pc = fma(pc, z, 1.0 / 24.0);
    pc = fma(pc, z, -0.5);
    double cf = fma(z, pc, 1.0);
    *cs = fma(-sk, sf, ck * cf);
    *sn = fma(ck, sf, sk * cf);
}

/* Specification v3: one Philox call -> FOUR standard normals, two Box-Muller pairs.  counter = (path, stream, iter, call),
 * key = (seed_lo, seed_hi); stream 0 = Wiener normals, 1 = accept uniforms, 2 = pCN move of the start.
 * Pair h (normals 2h, 2h+1) = half h & 1 of call h >> 1: words a = r[2s], b = r[2s+1];
 *   u1 = (K40 + 1) 2^-40 in (0,1], K40 = (b >> 24) 2^32 + a;   u2 = K24 2^-24, K24 = b & 0xffffff. */
/* Specification v2 (selectable in the product: BHIP_OPT_NOISE_SPEC = 2; here: bo_set_noise_spec(2)) -- the full-resolution stream the
 * round-2 library drew and tests/golden/guided_paths_v2 / _v3.npz hold: Philox call h -> pair h (normals 2h, 2h+1) with all 128 bits,
 *   u1 = (bits53(r0, r1) + 1) 2^-53 in (0,1],   u2 = bits53(r2, r3) 2^-53 in [0,1)   (bits53(lo, hi) = ((hi << 32 | lo) >> 11)). */
/* Specification v4 (round 5; the product's default): the same calls and pairs as v3 -- pair h = half h & 1 of call h >> 1, words
 * a = r[2s], b = r[2s+1] -- but each 32-bit word is ONE normal through a piecewise polynomial inverse of the normal distribution
 * function: v = 2 (w mod 2^31) + 1 (odd; upper-tail probability p = v 2^-33), d = (double) v, row R = bits 17..24 of the high word
 * of d (five exponent bits, three mantissa bits: the octave of p and its eighth) of the table BO_ICDF (data; generated by
 * scripts/gen_icdf_table.py: degree-4 minimax polynomials of d -> -Phi^-1(d 2^-33)), |z| by Horner in four fma, sign = bit 31 of w. */
#include "bo_icdf_table.h"
static const double BO_ICDF[5 * BO_ICDF_ROWS] = BO_ICDF_INIT;
BO_CLONES double bo_icdf_normal(uint32_t w)
{
    union { double d; uint64_t u; } b;
    uint32_t v = (w << 1) | 1u;
    b.d = (double)v;
    double d = b.d;
    uint32_t R = ((uint32_t)(b.u >> 32) >> 17) & 255u;
    const double *c = BO_ICDF + 5 * R;
    double q = fma(c[4], d, c[3]);
    q = fma(q, d, c[2]);
    q = fma(q, d, c[1]);
    q = fma(q, d, c[0]);
    b.d = q;
    b.u = (b.u & 0x7fffffffffffffffULL) | ((uint64_t)(w & 0x80000000u) << 32);
    return b.d;
}
void bo_icdf_normals(const uint32_t *w, long n, double *z)
{
    for (long i = 0; i < n; i++) z[i] = bo_icdf_normal(w[i]);
}
static int bo_noise_spec = 4;
void bo_set_noise_spec(int spec) { bo_noise_spec = (spec == 2 || spec == 3) ? spec : 4; }
int bo_get_noise_spec(void) { return bo_noise_spec; }
static void bo_normal_pair_v2(uint64_t seed, uint32_t path, uint32_t stream, uint32_t iter, uint32_t h, double z[2])
{
    uint32_t ctr[4] = {path, stream, iter, h}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
    bo_philox4x32_10(ctr, key, r);
    uint64_t a = ((uint64_t)r[1] << 32) | r[0], b = ((uint64_t)r[3] << 32) | r[2];
    double u1 = (double)((a >> 11) + 1) * 0x1.0p-53;    /* (0,1] */
    double u2 = (double)(b >> 11) * 0x1.0p-53;          /* [0,1) */
    double rad = sqrt(bo_m2log(u1));
    double s, c;
    bo_sincos2pi(u2, r[3], &s, &c);
    z[0] = rad * c;
    z[1] = rad * s;
}
BO_CLONES void bo_normal_pair_stream(uint64_t seed, uint32_t path, uint32_t stream, uint32_t iter, uint32_t h, double z[2])
{
    if (bo_noise_spec == 2) { bo_normal_pair_v2(seed, path, stream, iter, h, z); return; }
    uint32_t ctr[4] = {path, stream, iter, h >> 1}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
    bo_philox4x32_10(ctr, key, r);
    uint32_t a = r[2 * (h & 1u)], b = r[2 * (h & 1u) + 1];
    if (bo_noise_spec != 3) { z[0] = bo_icdf_normal(a); z[1] = bo_icdf_normal(b); return; }
    uint64_t k40 = ((uint64_t)(b >> 24) << 32) | a;
    double u1 = (double)(k40 + 1) * 0x1.0p-40;          /* (0,1] */
    double rad = sqrt(bo_m2log(u1));
    double s, c;
    bo_sincos2pi_k24(b & 0xffffffu, &s, &c);
    z[0] = rad * c;
    z[1] = rad * s;
}
BO_CLONES void bo_normal_pair(uint64_t seed, uint32_t path, uint32_t iter, uint32_t block, double z[2])
{
    bo_normal_pair_stream(seed, path, 0u, iter, block, z);
}

double bo_uniform_accept(uint64_t seed, uint32_t path, uint32_t iter)
{
    uint32_t ctr[4] = {path, 1u, iter, 0u}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
    bo_philox4x32_10(ctr, key, r);
    uint64_t a = ((uint64_t)r[1] << 32) | r[0];
    return (double)((a >> 11) + 1) * 0x1.0p-53;
}

void bo_normals(uint64_t seed, uint32_t path, uint32_t iter, int n0, int n, double *z)
{
    double pr[2]; long have = -1;
    for (int j = 0; j < n; j++) {
        int idx = n0 + j;
        if ((idx >> 1) != have) { bo_normal_pair(seed, path, iter, (uint32_t)(idx >> 1), pr); have = idx >> 1; }
        z[j] = pr[idx & 1];
    }
}

/* ------------------------------------------------------------------------------------------
 * target models: b(t,x,P), _scale(dw, sigma(t,x,P)), a(t,x,P)
 * ------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------
 * sin / cos of the drift functions (NclarDiffusion, IntegratedDiffusion-with-sin, Pendulum).
 * Julia's sin(::Float64) / cos(::Float64) (base/special/trig.jl -- Julia Base, not part of /root/reference) are ports of
 * fdlibm: argument reduction by pi/2 (Cody-Waite with a 33+33+53-bit split of pi/2 for |x| < 2^20*pi/2, Payne-Hanek
 * beyond), then the kernels __kernel_sin / __kernel_cos on the reduced double-double argument.  Restated here in that form
 * so that oracle, host C++ and the HIP kernels (bhip_trig.h) share ONE definition and agree bit for bit -- glibc's sin and
 * the device library's differ from each other and from Julia's in the last place.  Not bit-identical to Julia either:
 * its kernels evaluate the polynomials with muladd, which fuses or not depending on the CPU; the restatement never fuses.
 * Always two reduction steps (118 bits of pi/2: the closest approach of a double below 2^20*pi/2 to a multiple of pi/2
 * leaves > 53 significant bits), no early exit, round-to-nearest-even for the quadrant count.
 * Domain: |x| < 2^20*pi/2 ~ 1.647e6; outside (and for NaN/Inf) the result is NaN -- the Payne-Hanek range is not
 * restated; a drift argument of that size means the path has already blown up.  Error <= 1 ulp (tests/test_oracle.py). */
static const double TRIG_INVPIO2 = 6.36619772367581382433e-01, TRIG_PIO2_1 = 1.57079632673412561417e+00,
                    TRIG_PIO2_1T = 6.07710050650619224932e-11, TRIG_PIO2_2 = 6.07710050630396597660e-11,
                    TRIG_PIO2_2T = 2.02226624879595063154e-21, TRIG_MAX = 1647099.3291652855;
static int trig_reduce(double x, double *y0, double *y1)
{
    const double fn = rint(x * TRIG_INVPIO2);
    double r = x - fn * TRIG_PIO2_1, w, t;          /* fn*pio2_1 is exact: 33 + 20 bits */
    t = r; w = fn * TRIG_PIO2_2; r = t - w;
    w = fn * TRIG_PIO2_2T - ((t - r) - w);
    (void)TRIG_PIO2_1T;
    *y0 = r - w;
    *y1 = (r - *y0) - w;
    return (int)fn;
}
static double trig_ksin(double x, double y)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
static double trig_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
double bo_sin(double x)
{
    if (!(fabs(x) < TRIG_MAX)) return NAN;
    double y0, y1;
    const int q = trig_reduce(x, &y0, &y1) & 3;
    const double s = trig_ksin(y0, y1), c = trig_kcos(y0, y1);
    const double r = (q & 1) ? c : s;
    return (q & 2) ? -r : r;
}
double bo_cos(double x)
{
    if (!(fabs(x) < TRIG_MAX)) return NAN;
    double y0, y1;
    const int q = trig_reduce(x, &y0, &y1) & 3;
    const double s = trig_ksin(y0, y1), c = trig_kcos(y0, y1);
    const double r = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -r : r;
}

int bo_model_dims(int model, int d_hint, int *d, int *mp)
{
    switch (model) {
    case BO_MODEL_WIENER: *d = d_hint; *mp = d_hint; return 0;
    case BO_MODEL_OU: *d = 1; *mp = 1; return 0;
    case BO_MODEL_LINPRO: *d = d_hint; *mp = d_hint; return 0;
    case BO_MODEL_LORENZ96: *d = d_hint; *mp = d_hint; return 0;
    case BO_MODEL_FHN: *d = 2; *mp = 1; return 0;
    case BO_MODEL_NCLAR: *d = 3; *mp = 1; return 0;
    case BO_MODEL_INTDIFF: *d = 2; *mp = 1; return 0;
    case BO_MODEL_LORENZ: *d = 3; *mp = 3; return 0;
    case BO_MODEL_FHN2: *d = 2; *mp = 2; return 0;
    case BO_MODEL_PENDULUM: *d = 2; *mp = 1; return 0;
    case BO_MODEL_SDIFF1: *d = 1; *mp = 1; return 0;
    case BO_MODEL_SDIFF2: *d = 2; *mp = 2; return 0;
    }
    return -1;
}

/* constdiff(P), src/types.jl:36 */
int bo_constdiff(int model) { return model != BO_MODEL_SDIFF1 && model != BO_MODEL_SDIFF2; }

void bo_b(int model, int d, const double *p, double t, const double *x, double *o)
{
    (void)t;
    switch (model) {
    case BO_MODEL_WIENER: /* src/wiener.jl:143 b = 0 */
        for (int k = 0; k < d; k++) o[k] = 0.0;
        break;
    case BO_MODEL_OU: /* test/guip.jl:21, README.md:75  b = -beta*x */
        o[0] = -p[0] * x[0];
        break;
    case BO_MODEL_LINPRO: { /* src/linpro.jl:80  b = P.B*(x .- P.mu) */
        double xm[BO_MAXD];
        const double *B = p, *mu = p + d * d;
        for (int k = 0; k < d; k++) xm[k] = x[k] - mu[k];
        mv(d, d, B, xm, o);
        break; }
    case BO_MODEL_LORENZ96: /* stand-in user drift at d > 3: b_k = (x_{k+1} - x_{k-2})*x_{k-1} - x_k + F, indices mod d */
        for (int k = 0; k < d; k++) o[k] = (x[(k + 1) % d] - x[(k + d - 2) % d]) * x[(k + d - 1) % d] - x[k] + p[0];
        break;
    case BO_MODEL_FHN: /* partialbridge_fitzhugh.jl:44  ((x1-x2-x1^3+s)/eps, gamma*x1-x2+beta) */
        o[0] = (x[0] - x[1] - x[0] * x[0] * x[0] + p[1]) / p[0];
        o[1] = p[2] * x[0] - x[1] + p[3];
        break;
    case BO_MODEL_NCLAR: /* partialbridge_nclar.jl:58  (x2, x3, -alpha*sin(omega*x3)) */
        o[0] = x[1]; o[1] = x[2]; o[2] = -p[0] * bo_sin(p[1] * x[2]);
        break;
    case BO_MODEL_INTDIFF: /* test/partialbridge.jl:11-12  (x2, -(x2+sin(x2)) + 1/2) */
        o[0] = x[1]; o[1] = -(x[1] + bo_sin(x[1])) + 0.5;
        break;
    case BO_MODEL_LORENZ: /* src/Models.jl:47 */
        o[0] = p[0] * (x[1] - x[0]);
        o[1] = x[0] * (p[1] - x[2]) - x[1];
        o[2] = x[0] * x[1] - p[2] * x[2];
        break;
    case BO_MODEL_FHN2: /* src/Models.jl:18  (eps\(x1 - x1^3 - x2 + s), gamma*x1 - x2 + beta) */
        o[0] = (x[0] - x[0] * x[0] * x[0] - x[1] + p[1]) / p[0];
        o[1] = p[2] * x[0] - x[1] + p[3];
        break;
    case BO_MODEL_PENDULUM: /* src/Models.jl:79  (x2, -theta2*sin(x1)) */
        o[0] = x[1]; o[1] = -p[0] * bo_sin(x[0]);
        break;
    case BO_MODEL_SDIFF1:
        o[0] = p[0] * (p[1] - x[0]);
        break;
    case BO_MODEL_SDIFF2:
        o[0] = p[0] * (p[1] - x[0]) + p[2] * x[1];
        o[1] = p[3] * (p[4] - x[1]);
        break;
    }
}

/* _scale(dw, sigma(t,x,P)) = sigma*dw, src/euler.jl:3-4 */
static void model_sigma_mat(int model, int d, int mp, const double *p, double t, const double *x, double *S);

void bo_sigma_apply(int model, int d, int mp, const double *p, double t, const double *x,
                    const double *dw, double *o)
{
    (void)t; (void)x; (void)mp;
    if (!bo_constdiff(model)) {   /* sigma(t,x,P)*dw with the full matrix */
        double S[D2];
        model_sigma_mat(model, d, mp, p, t, x, S);
        mv(d, mp, S, dw, o);
        return;
    }
    switch (model) {
    case BO_MODEL_WIENER: for (int k = 0; k < d; k++) o[k] = dw[k]; break;  /* sigma = I */
    case BO_MODEL_OU: o[0] = p[1] * dw[0]; break;
    case BO_MODEL_LINPRO: mv(d, d, p + d * d + d, dw, o); break;            /* P.sigma*dw */
    case BO_MODEL_LORENZ96: mv(d, d, p + 1, dw, o); break;
    case BO_MODEL_FHN: o[0] = 0.0 * dw[0]; o[1] = p[4] * dw[0]; break;      /* R2(0, sigma)*dw */
    case BO_MODEL_NCLAR: o[0] = 0.0 * dw[0]; o[1] = 0.0 * dw[0]; o[2] = p[2] * dw[0]; break;
    case BO_MODEL_INTDIFF: o[0] = 0.0 * dw[0]; o[1] = p[0] * dw[0]; break;
    case BO_MODEL_LORENZ: for (int k = 0; k < 3; k++) o[k] = p[3 + k] * dw[k]; break;  /* SDiagonal */
    case BO_MODEL_FHN2: o[0] = p[4] * dw[0]; o[1] = p[5] * dw[1]; break;
    case BO_MODEL_PENDULUM: o[0] = 0.0 * dw[0]; o[1] = p[1] * dw[0]; break;
    }
}

/* sigma as a d x mp matrix (for a = sigma*sigma') */
static void model_sigma_mat(int model, int d, int mp, const double *p, double t, const double *x, double *S)
{
    (void)t;
    memset(S, 0, sizeof(double) * d * mp);
    switch (model) {
    case BO_MODEL_SDIFF1: S[0] = p[2] * sqrt(1.0 + x[0] * x[0]); break;
    case BO_MODEL_SDIFF2: S[0] = p[5] * sqrt(1.0 + x[0] * x[0]); S[2] = p[7] * x[1]; S[3] = p[6]; break;
    case BO_MODEL_WIENER: for (int k = 0; k < d; k++) S[k + d * k] = 1.0; break;
    case BO_MODEL_OU: S[0] = p[1]; break;
    case BO_MODEL_LINPRO: memcpy(S, p + d * d + d, sizeof(double) * d * d); break;
    case BO_MODEL_LORENZ96: memcpy(S, p + 1, sizeof(double) * d * d); break;
    case BO_MODEL_FHN: S[1] = p[4]; break;
    case BO_MODEL_NCLAR: S[2] = p[2]; break;
    case BO_MODEL_INTDIFF: S[1] = p[0]; break;
    case BO_MODEL_LORENZ: for (int k = 0; k < 3; k++) S[k + 3 * k] = p[3 + k]; break;
    case BO_MODEL_FHN2: S[0] = p[4]; S[3] = p[5]; break;
    case BO_MODEL_PENDULUM: S[1] = p[1]; break;
    }
}

/* a(t,x,P): src/types.jl:32 fallback outer(sigma) = sigma*sigma'; OU: sigma^2 (test/guip.jl:23);
 * LinPro: P.a = sigma*sigma' (src/linpro.jl:72,84) */
void bo_a(int model, int d, int mp, const double *p, double t, const double *x, double *A)
{
    double S[D2];
    model_sigma_mat(model, d, mp, p, t, x, S);
    mmt(d, mp, d, S, S, A);
}

/* ------------------------------------------------------------------------------------------
 * auxiliary linear processes
 * ------------------------------------------------------------------------------------------ */
static double fhn_uv(const double *p, double t)
{ /* partialbridge_fitzhugh.jl:70-73 */
    double lam = (t - p[5]) / (p[7] - p[5]);
    return p[8] * lam + p[6] * (1 - lam);
}
/* LinearAppr (src/linpro.jl:181-192): coefficients per grid INDEX.  par = N, tt(N), xx(N*d), B(N*d*d), b(N*d), Sigma(N*d*mp);
 * the accessors below are called with grid times only, the index is the exact match. */
static int la_index(const double *p, double t)
{
    int N = (int)p[0], lo = 0, hi = N - 1;
    const double *tt = p + 1;
    while (lo < hi) { int mid = (lo + hi) / 2; if (tt[mid] < t) lo = mid + 1; else hi = mid; }
    return lo;   /* tt[lo] == t for grid times */
}
static const double *la_xx(const double *p, int d, int i) { return p + 1 + (int)p[0] + (size_t)i * d; }
static const double *la_B(const double *p, int d, int i) { int N = (int)p[0]; return p + 1 + N + (size_t)N * d + (size_t)i * d * d; }
static const double *la_b(const double *p, int d, int i) { int N = (int)p[0]; return p + 1 + N + (size_t)N * d + (size_t)N * d * d + (size_t)i * d; }
static const double *la_S(const double *p, int d, int mp, int i)
{
    int N = (int)p[0];
    return p + 1 + N + (size_t)N * d + (size_t)N * d * d + (size_t)N * d + (size_t)i * d * mp;
}

void bo_aux_B(int aux, int d, const double *p, double t, double *B)
{
    if (aux == BO_AUX_LINEARAPPR) { memcpy(B, la_B(p, d, la_index(p, t)), sizeof(double) * d * d); return; }   /* B((i,s), P) = P.B[i]  :188 */
    if (aux == BO_AUX_FHN_STARTEND) { /* :103  [1/eps-3*uv^2/eps  -1/eps; gamma -1.0] */
        double uv = fhn_uv(p, t);
        B[0] = 1 / p[0] - 3 * (uv * uv) / p[0]; B[2] = -1 / p[0];
        B[1] = p[2]; B[3] = -1.0;
        return;
    }
    memcpy(B, p, sizeof(double) * d * d);
}
void bo_aux_beta(int aux, int d, const double *p, double t, double *beta)
{
    if (aux == BO_AUX_LINEARAPPR) {   /* beta((i,s), P) = P.b[i] - P.B[i]*P.xx[i]   src/linpro.jl:189 */
        int i = la_index(p, t);
        double Bx[BO_MAXD];
        mv(d, d, la_B(p, d, i), la_xx(p, d, i), Bx);
        for (int k = 0; k < d; k++) beta[k] = la_b(p, d, i)[k] - Bx[k];
        return;
    }
    if (aux == BO_AUX_FHN_STARTEND) { /* :104  (s/eps + 2*uv^3/eps, beta) */
        double uv = fhn_uv(p, t);
        beta[0] = p[1] / p[0] + 2 * (uv * uv * uv) / p[0];
        beta[1] = p[3];
        return;
    }
    if (aux == BO_AUX_LINPRO) { /* src/linpro.jl:79  beta = -P.B*P.mu */
        double nB[D2];
        for (int k = 0; k < d * d; k++) nB[k] = -p[k];
        mv(d, d, nB, p + d * d, beta);
        return;
    }
    memcpy(beta, p + d * d, sizeof(double) * d);
}
void bo_aux_sigma(int aux, int d, int mp, const double *p, double t, double *S)
{
    if (aux == BO_AUX_LINEARAPPR) { memcpy(S, la_S(p, d, mp, la_index(p, t)), sizeof(double) * d * mp); return; }   /* a = outer(P.Sigma[i])  :190-191 */
    if (aux == BO_AUX_FHN_STARTEND) { S[0] = 0.0; S[1] = p[4]; return; }
    memcpy(S, p + d * d + d, sizeof(double) * d * mp);
}
void bo_aux_a(int aux, int d, int mp, const double *p, double t, double *A)
{ /* a(t,Pt) = sigma(t,Pt)*sigma(t,Pt)'  (partialbridge_fitzhugh.jl:115; linpro.jl:72) */
    double S[D2];
    bo_aux_sigma(aux, d, mp, p, t, S);
    mmt(d, mp, d, S, S, A);
}
/* b(t,x,Pt): affine processes B(t)*x + beta(t) (partialbridge_fitzhugh.jl:114); LinPro B*(x-mu) */
void bo_aux_b(int aux, int d, const double *p, double t, const double *x, double *o)
{
    if (aux == BO_AUX_LINPRO) {
        double xm[BO_MAXD];
        for (int k = 0; k < d; k++) xm[k] = x[k] - p[d * d + k];
        mv(d, d, p, xm, o);
        return;
    }
    double B[D2], beta[BO_MAXD];
    bo_aux_B(aux, d, p, t, B);
    bo_aux_beta(aux, d, p, t, beta);
    mv(d, d, B, x, o);
    for (int k = 0; k < d; k++) o[k] = o[k] + beta[k];
}

/* ------------------------------------------------------------------------------------------
 * Ralston-3 step  src/ode.jl:44-49
 *   k1=f(t,y); k2=f(t+1/2*dt, y+1/2*dt*k1); k3=f(t+3/4*dt, y+3/4*dt*k2);
 *   y + dt*(2/9*k1 + 1/3*k2 + 4/9*k3)
 * ------------------------------------------------------------------------------------------ */
typedef void (*rhs_fn)(double t, const double *y, double *k, void *ctx);
static void kernelr3(rhs_fn f, double t, const double *y, double dt, int n, void *ctx, double *out)
{
    double k1[D2], k2[D2], k3[D2], y2[D2];
    f(t, y, k1, ctx);
    double h2 = 1.0 / 2 * dt, h34 = 3.0 / 4 * dt;
    for (int i = 0; i < n; i++) y2[i] = y[i] + h2 * k1[i];
    f(t + h2, y2, k2, ctx);
    for (int i = 0; i < n; i++) y2[i] = y[i] + h34 * k2[i];
    f(t + h34, y2, k3, ctx);
    for (int i = 0; i < n; i++) out[i] = y[i] + dt * (2.0 / 9 * k1[i] + 1.0 / 3 * k2[i] + 4.0 / 9 * k3[i]);
}

typedef struct { int aux, d, mp, m; const double *apar; const double *L; } ode_ctx;

/* src/gode.jl:3  _dHinv(t,K,P) = B*K + K*B' - a */
static void rhs_dHinv(double t, const double *K, double *out, void *vc)
{
    ode_ctx *c = (ode_ctx *)vc; int d = c->d;
    double B[D2], A[D2], BK[D2], KBt[D2];
    bo_aux_B(c->aux, d, c->apar, t, B);
    bo_aux_a(c->aux, d, c->mp, c->apar, t, A);
    mm(d, d, d, B, K, BK);
    mmt(d, d, d, K, B, KBt);
    for (int i = 0; i < d * d; i++) out[i] = BK[i] + KBt[i] - A[i];
}
/* src/gode.jl:4  _dK(t,K,P) = B*K + K*B' + a */
static void rhs_dK(double t, const double *K, double *out, void *vc)
{
    ode_ctx *c = (ode_ctx *)vc; int d = c->d;
    double B[D2], A[D2], BK[D2], KBt[D2];
    bo_aux_B(c->aux, d, c->apar, t, B);
    bo_aux_a(c->aux, d, c->mp, c->apar, t, A);
    mm(d, d, d, B, K, BK);
    mmt(d, d, d, K, B, KBt);
    for (int i = 0; i < d * d; i++) out[i] = BK[i] + KBt[i] + A[i];
}
/* src/gode.jl:2  _F(t,x,P) = B*x + beta */
static void rhs_F(double t, const double *x, double *out, void *vc)
{
    ode_ctx *c = (ode_ctx *)vc; int d = c->d;
    double B[D2], beta[BO_MAXD];
    bo_aux_B(c->aux, d, c->apar, t, B);
    bo_aux_beta(c->aux, d, c->apar, t, beta);
    mv(d, d, B, x, out);
    for (int i = 0; i < d; i++) out[i] = out[i] + beta[i];
}
/* src/gode.jl:5  _dPhi(t,Phi,P) = B*Phi */
static void rhs_dPhi(double t, const double *Phi, double *out, void *vc)
{
    ode_ctx *c = (ode_ctx *)vc; int d = c->d;
    double B[D2];
    bo_aux_B(c->aux, d, c->apar, t, B);
    mm(d, d, d, B, Phi, out);
}
/* src/guip.jl:202  _traceB(t,x,P) = tr(B(t,P)) */
static void rhs_traceB(double t, const double *y, double *out, void *vc)
{
    (void)y;
    ode_ctx *c = (ode_ctx *)vc; int d = c->d;
    double B[D2];
    bo_aux_B(c->aux, d, c->apar, t, B);
    double s = B[0];
    for (int i = 1; i < d; i++) s += B[i + d * i];
    out[0] = s;
}
/* the process drift itself, b(t,x,P) of a LinPro, used by test/linprobridge.jl:22 */
static void rhs_b(double t, const double *x, double *out, void *vc)
{
    ode_ctx *c = (ode_ctx *)vc;
    bo_aux_b(c->aux, c->d, c->apar, t, x, out);
}

/* GuidedBridge constructor src/guip.jl:172-180: gpHinv!(Hd, Pt, hT); gpV!(V, Pt, v)
 * via _solvebackward!(R3, ...) src/ode.jl:88-97 */
void bo_gp_hv(const double *tt, int N, int d, int mp, int aux, const double *apar,
              const double *v, const double *hT, double *Hd, double *V)
{
    ode_ctx c = {aux, d, mp, 0, apar, 0};
    int dd = d * d;
    double y[D2];
    for (int k = 0; k < dd; k++) y[k] = hT ? hT[k] : 0.0;
    memcpy(Hd + (size_t)(N - 1) * dd, y, sizeof(double) * dd);
    for (int i = N - 2; i >= 0; i--) {
        kernelr3(rhs_dHinv, tt[i + 1], y, tt[i] - tt[i + 1], dd, &c, y);
        memcpy(Hd + (size_t)i * dd, y, sizeof(double) * dd);
    }
    double w[BO_MAXD];
    memcpy(w, v, sizeof(double) * d);
    memcpy(V + (size_t)(N - 1) * d, w, sizeof(double) * d);
    for (int i = N - 2; i >= 0; i--) {
        kernelr3(rhs_F, tt[i + 1], w, tt[i] - tt[i + 1], d, &c, w);
        memcpy(V + (size_t)i * d, w, sizeof(double) * d);
    }
}

/* GuidedBridge(tt, P, Pt::LinearAppr, v, hT)  src/guip.jl:181-189:
 *     solvebackwardi!(Heun(), ((i,t), K, P) -> B((i,t),P)*K + K*B((i,t),P)' - a((i,t),P), Hd, hT, Pt)
 *     solvebackwardi!(Heun(), b, V, v, Pt)
 * with  solvebackwardi!  src/ode.jl:104-113  (for i in N-1:-1:1: y = kerneli(ker, F, tt[i+1], y, tt[i]-tt[i+1], P))  and
 *     kerneli(::Heun, f, t, y, dt, P):  k1 = f((i,t), y, P);  k2 = f((i+1, t+dt), y + dt*k1, P);  y + dt/2*(k1 + k2)      :98-102
 * As committed in the reference `kerneli` reads an `i` that solvebackwardi! never passes (an UndefVarError), and the second
 * call hands `b` where only `_b((i,s), x, P::LinearAppr)` exists (src/linpro.jl:187): the constructor cannot run there.
 * Restated with the evident intention -- `i` is the loop index of solvebackwardi!, `b` means _b -- i.e. (1-based i)
 *     k1 = f_i(y),  k2 = f_{i+1}(y + dt*k1),  y <- y + dt/2*(k1 + k2),   dt = tt[i] - tt[i+1] < 0,
 * f_j(K) = B_j K + K B_j' - outer(Sigma_j),   f_j(x) = B_j (x - xx_j) + b_j.  A documented deviation. */
static void la_fH(int d, int mp, const double *Bj, const double *Sj, const double *K, double *out)
{
    double BK[D2], KBt[D2], a[D2];
    mm(d, d, d, Bj, K, BK);
    mmt(d, d, d, K, Bj, KBt);
    mmt(d, mp, d, Sj, Sj, a);
    for (int k = 0; k < d * d; k++) out[k] = BK[k] + KBt[k] - a[k];
}
static void la_fV(int d, const double *Bj, const double *xxj, const double *bj, const double *x, double *out)
{
    double xm[BO_MAXD];
    for (int k = 0; k < d; k++) xm[k] = x[k] - xxj[k];
    mv(d, d, Bj, xm, out);
    for (int k = 0; k < d; k++) out[k] = out[k] + bj[k];
}
void bo_gp_hv_heuni(const double *tt, int N, int d, int mp, const double *xx, const double *B, const double *b, const double *Sigma,
                    const double *v, const double *hT, double *Hd, double *V)
{
    const int dd = d * d;
    double y[D2], k1[D2], k2[D2], yp[D2];
    for (int k = 0; k < dd; k++) y[k] = hT ? hT[k] : 0.0;
    memcpy(Hd + (size_t)(N - 1) * dd, y, sizeof(double) * dd);
    for (int i = N - 2; i >= 0; i--) {   /* 0-based i = Julia's i - 1 */
        const double dt = tt[i] - tt[i + 1];
        la_fH(d, mp, B + (size_t)i * dd, Sigma + (size_t)i * d * mp, y, k1);
        for (int k = 0; k < dd; k++) yp[k] = y[k] + dt * k1[k];
        la_fH(d, mp, B + (size_t)(i + 1) * dd, Sigma + (size_t)(i + 1) * d * mp, yp, k2);
        for (int k = 0; k < dd; k++) y[k] = y[k] + dt / 2 * (k1[k] + k2[k]);
        memcpy(Hd + (size_t)i * dd, y, sizeof(double) * dd);
    }
    double w[BO_MAXD], wp[BO_MAXD];
    memcpy(w, v, sizeof(double) * d);
    memcpy(V + (size_t)(N - 1) * d, w, sizeof(double) * d);
    for (int i = N - 2; i >= 0; i--) {
        const double dt = tt[i] - tt[i + 1];
        la_fV(d, B + (size_t)i * dd, xx + (size_t)i * d, b + (size_t)i * d, w, k1);
        for (int k = 0; k < d; k++) wp[k] = w[k] + dt * k1[k];
        la_fV(d, B + (size_t)(i + 1) * dd, xx + (size_t)(i + 1) * d, b + (size_t)(i + 1) * d, wp, k2);
        for (int k = 0; k < d; k++) w[k] = w[k] + dt / 2 * (k1[k] + k2[k]);
        memcpy(V + (size_t)i * d, w, sizeof(double) * d);
    }
}

/* Bridge.bderiv(t, x, P): the Jacobian of the drift, for the processes the reference defines it for:
 * Lorenz src/Models.jl:49-53, Pendulum :81-84, LinPro src/linpro.jl:82, Wiener src/wiener.jl:147.  Column-major d x d. */
void bo_bderiv(int model, int d, const double *p, double t, const double *x, double *J)
{
    (void)t;
    memset(J, 0, sizeof(double) * d * d);
    if (model == BO_MODEL_LORENZ) {
        J[0] = -p[0];        J[3] = p[0];  J[6] = 0.0;
        J[1] = p[1] - x[2];  J[4] = -1.0;  J[7] = -x[0];
        J[2] = x[1];         J[5] = x[0];  J[8] = -p[2];
    } else if (model == BO_MODEL_PENDULUM) {
        J[0] = 0.0;                  J[2] = 1.0;
        J[1] = -p[0] * bo_cos(x[0]); J[3] = 0.0;
    } else if (model == BO_MODEL_LINPRO) {
        memcpy(J, p, sizeof(double) * d * d);
    }
}

/* linearappr(Y, P) / linearappr!(Pt, Y, P)  src/linpro.jl:196-204: B_i = bderiv(t_i, y_i, P), b_i = b(t_i, y_i, P),
 * Sigma_i = sigma(t_i, y_i, P) along the path Y (xx_i = y_i). */
void bo_linearappr(int model, int d, int mp, const double *par, const double *tt, int N, const double *Y,
                   double *B, double *b, double *Sigma)
{
    for (int i = 0; i < N; i++) {
        bo_bderiv(model, d, par, tt[i], Y + (size_t)i * d, B + (size_t)i * d * d);
        bo_b(model, d, par, tt[i], Y + (size_t)i * d, b + (size_t)i * d);
        /* sigma as a matrix: apply it to the unit vectors */
        for (int c = 0; c < mp; c++) {
            double e[BO_MAXD] = {0}, col[BO_MAXD];
            e[c] = 1.0;
            bo_sigma_apply(model, d, mp, par, tt[i], Y + (size_t)i * d, e, col);
            for (int r = 0; r < d; r++) Sigma[(size_t)i * d * mp + r + d * c] = col[r];
        }
    }
}

/* LinearNoiseAppr(tt, P, x, a, direction)  src/guip.jl:114-146 ("precursor of the linear noise approximation"):
 *   Y = the deterministic path y' = b(t, y, P) by Ralston-3: solve!(R3(), b, Y, x, P) forward from x at tt[1]
 *       (direction 1, src/ode.jl:178-184), solvebackward!(R3(), b, Y, x, P) backward from x at tt[N] (direction -1,
 *       src/ode.jl:88-97), zeros for :nothing (0);
 *   B(t, P) = 0I;  beta((i,t), P) = (Y[i] - Y[i-1])/(tt[i] - tt[i-1]);  _b((i,t), x, P) = beta((max(i,2), t), P);  a = P.a.
 * As committed the type cannot be used in a GuidedBridge either: `_b` calls an undefined `beta_`, `a((i,t), P)` (two
 * arguments) has no method, and the constructor hands `b` where only `_b` exists (same defects as LinearAppr, see
 * bo_gp_hv_heuni).  Restated with the evident intention; in the index-based Heun solver it is then exactly a LinearAppr
 * with B_i = 0, xx_i = 0, b_i = beta at max(i,2) and Sigma_i*Sigma_i' = a -- which is how oracle and product carry it.
 * The target's sigma supplies Sigma_i (the scripts pass a = a(t, v, P) of the target, supplements/smoothing/smoothing.jl:80,85). */
static void rhs_target_b(double t, const double *x, double *out, void *vc)
{
    ode_ctx *c = (ode_ctx *)vc;   /* aux field carries the MODEL id here, apar its parameters */
    bo_b(c->aux, c->d, c->apar, t, x, out);
}
void bo_lna_path(int model, int d, const double *par, const double *tt, int N, const double *x, int direction, double *Y)
{
    ode_ctx c = {model, d, 0, 0, par, 0};
    double y[BO_MAXD];
    memset(Y, 0, sizeof(double) * N * d);
    if (direction == 0) return;
    memcpy(y, x, sizeof(double) * d);
    if (direction > 0) {
        memcpy(Y, y, sizeof(double) * d);
        for (int i = 1; i < N; i++) { kernelr3(rhs_target_b, tt[i - 1], y, tt[i] - tt[i - 1], d, &c, y); memcpy(Y + (size_t)i * d, y, sizeof(double) * d); }
    } else {
        memcpy(Y + (size_t)(N - 1) * d, y, sizeof(double) * d);
        for (int i = N - 2; i >= 0; i--) { kernelr3(rhs_target_b, tt[i + 1], y, tt[i] - tt[i + 1], d, &c, y); memcpy(Y + (size_t)i * d, y, sizeof(double) * d); }
    }
}
/* the LinearAppr coefficients that carry a LinearNoiseAppr with deterministic path Y */
void bo_lna_coeffs(int model, int d, int mp, const double *par, const double *tt, int N, const double *Y,
                   double *xx, double *B, double *b, double *Sigma)
{
    memset(xx, 0, sizeof(double) * N * d);
    memset(B, 0, sizeof(double) * N * d * d);
    for (int j = 0; j < N; j++) {
        int jj = j < 1 ? 1 : j;                                    /* max(i, 2), 1-based */
        for (int k = 0; k < d; k++) b[(size_t)j * d + k] = (Y[(size_t)jj * d + k] - Y[(size_t)(jj - 1) * d + k]) / (tt[jj] - tt[jj - 1]);
        for (int c = 0; c < mp; c++) {
            double e[BO_MAXD] = {0}, col[BO_MAXD];
            e[c] = 1.0;
            bo_sigma_apply(model, d, mp, par, tt[j], Y + (size_t)j * d, e, col);
            for (int r = 0; r < d; r++) Sigma[(size_t)j * d * mp + r + d * c] = col[r];
        }
    }
}

/* gpupdate(Hd, V, L, Sigma, v)  src/guip.jl:221-231: fold an observation v = L x + N(0,Sigma) at the left
 * end of a segment into (Hdiamond, V) -- the backward link between chained GuidedBridge segments
 * (test/smoothing.jl:73-83).  Z = I - Hd*L'*inv(Sigma + L*Hd*L')*L ; (Z*Hd, Z*Hd*L'*inv(Sigma)*v + Z*V);
 * if all(diag(Hd) .== Inf): (inv(L'inv(Sigma)L), (L'inv(Sigma)L) \ (L'inv(Sigma)v)). */
void bo_gpupdate(int d, int m, const double *Hd, const double *V, const double *L, const double *Sigma,
                 const double *v, double *Hd_out, double *V_out)
{
    double Si[D2], LtSi[D2], T1[D2], T2[D2], T3[D2], Z[D2], ZH[D2], w1[BO_MAXD], w2[BO_MAXD];
    int allinf = 1;
    for (int k = 0; k < d; k++) if (!(isinf(Hd[k + d * k]) && Hd[k + d * k] > 0)) allinf = 0;
    bo_inv(m, Sigma, Si);
    if (allinf) {
        mtm(d, m, m, L, Si, LtSi);            /* L'*inv(Sigma)        d x m */
        mm(d, m, d, LtSi, L, T1);             /* (L'inv(Sigma))*L     d x d */
        bo_inv(d, T1, Hd_out);
        mv(d, m, LtSi, v, w1);                /* (L'inv(Sigma))*v           */
        bo_solve(d, T1, w1, V_out);
        return;
    }
    double LH[D2], S[D2], Sinv[D2];
    mm(m, d, d, L, Hd, LH);                   /* L*Hd                 m x d */
    mmt(m, d, m, LH, L, S);                   /* (L*Hd)*L'            m x m */
    for (int k = 0; k < m * m; k++) S[k] = Sigma[k] + S[k];
    bo_inv(m, S, Sinv);
    mmt(d, d, m, Hd, L, T1);                  /* Hd*L'                d x m */
    mm(d, m, m, T1, Sinv, T2);                /* (Hd*L')*inv(S)       d x m */
    mm(d, m, d, T2, L, T3);                   /* (...)*L              d x d */
    for (int j = 0; j < d; j++)
        for (int i = 0; i < d; i++) Z[i + d * j] = (i == j ? 1.0 : 0.0) - T3[i + d * j];
    mm(d, d, d, Z, Hd, ZH);
    memcpy(Hd_out, ZH, sizeof(double) * d * d);
    mmt(d, d, m, ZH, L, T1);                  /* (Z*Hd)*L'                  */
    mm(d, m, m, T1, Si, T2);                  /* (...)*inv(Sigma)           */
    mv(d, m, T2, v, w1);
    mv(d, d, Z, V, w2);
    for (int k = 0; k < d; k++) V_out[k] = w1[k] + w2[k];
}

/* forward R3 integration, src/ode.jl:178-184 solve(::R3, F, tt, x0, P);
 * what: 0 _F (gpmu), 1 _dK (gpK), 2 _dPhi (fundamental_matrix), 3 process drift b, 4 _dHinv */
double bo_r3_forward(const double *tt, int N, int d, int mp, int aux, const double *apar, int what,
                     const double *y0, int ny, double *yT)
{
    ode_ctx c = {aux, d, mp, 0, apar, 0};
    rhs_fn f = what == 0 ? rhs_F : what == 1 ? rhs_dK : what == 2 ? rhs_dPhi : what == 3 ? rhs_b : rhs_dHinv;
    double y[D2];
    memcpy(y, y0, sizeof(double) * ny);
    for (int i = 1; i < N; i++) kernelr3(f, tt[i - 1], y, tt[i] - tt[i - 1], ny, &c, y);
    memcpy(yT, y, sizeof(double) * ny);
    return y[0];
}

/* traceB(tt,P) = solve(R3(), _traceB, tt, 0.0, P)  src/guip.jl:203 */
double bo_traceB(const double *tt, int N, int d, int aux, const double *apar)
{
    ode_ctx c = {aux, d, 0, 0, apar, 0};
    double y = 0.0;
    for (int i = 1; i < N; i++) kernelr3(rhs_traceB, tt[i - 1], &y, tt[i] - tt[i - 1], 1, &c, &y);
    return y;
}

/* partialbridgeode!(::R3, t, L, Sigma, Lt, Mt, mut, P)  src/partialbridge.jl:1-22 */
static void rhs_L(double t, const double *L, double *out, void *vc)
{ /* (t,y,P) -> -y*B(t,P) */
    ode_ctx *c = (ode_ctx *)vc; int d = c->d, m = c->m;
    double B[D2], nL[D2];
    bo_aux_B(c->aux, d, c->apar, t, B);
    for (int i = 0; i < m * d; i++) nL[i] = -L[i];
    mm(m, d, d, nL, B, out);
}
static void rhs_Mplus(double t, const double *y, double *out, void *vc)
{ /* (t,y,(L,P)) -> -outer(L*sigma(t,P)) */
    (void)y;
    ode_ctx *c = (ode_ctx *)vc; int d = c->d, m = c->m, mp = c->mp;
    double S[D2], LS[D2], O[D2];
    bo_aux_sigma(c->aux, d, mp, c->apar, t, S);
    mm(m, d, mp, c->L, S, LS);
    mmt(m, mp, m, LS, LS, O);
    for (int i = 0; i < m * m; i++) out[i] = -O[i];
}
static void rhs_mu(double t, const double *y, double *out, void *vc)
{ /* (t,y,(L,P)) -> -L*beta(t,P) */
    (void)y;
    ode_ctx *c = (ode_ctx *)vc; int d = c->d, m = c->m;
    double beta[BO_MAXD], nL[D2];
    bo_aux_beta(c->aux, d, c->apar, t, beta);
    for (int i = 0; i < m * d; i++) nL[i] = -c->L[i];
    mv(m, d, nL, beta, out);
}
void bo_partialbridge_ode(const double *tt, int N, int d, int mp, int m, int aux, const double *apar,
                          const double *L0, const double *Sigma, double *Lt, double *Mt, double *mut)
{
    double L[D2], Mp[D2], mu[BO_MAXD], Mi[D2];
    ode_ctx c = {aux, d, mp, m, apar, L};
    memcpy(L, L0, sizeof(double) * m * d);
    memcpy(Mp, Sigma, sizeof(double) * m * m);
    for (int k = 0; k < m; k++) mu[k] = 0 * L0[k];              /* mu = 0*L[:,1] */
    memcpy(Lt + (size_t)(N - 1) * m * d, L, sizeof(double) * m * d);
    bo_inv(m, Sigma, Mi);                                       /* Mt[end] = inv(Sigma) (Inf if 0) */
    memcpy(Mt + (size_t)(N - 1) * m * m, Mi, sizeof(double) * m * m);
    memcpy(mut + (size_t)(N - 1) * m, mu, sizeof(double) * m);
    for (int i = N - 2; i >= 0; i--) {
        double dt = tt[i] - tt[i + 1];
        kernelr3(rhs_L, tt[i + 1], L, dt, m * d, &c, L);        /* L first ...                   */
        kernelr3(rhs_Mplus, tt[i + 1], Mp, dt, m * m, &c, Mp);  /* ... M+ and mu use the NEW L   */
        kernelr3(rhs_mu, tt[i + 1], mu, dt, m, &c, mu);
        memcpy(Lt + (size_t)i * m * d, L, sizeof(double) * m * d);
        bo_inv(m, Mp, Mi);
        memcpy(Mt + (size_t)i * m * m, Mi, sizeof(double) * m * m);
        memcpy(mut + (size_t)i * m, mu, sizeof(double) * m);
    }
}

/* updatenuH+C + partialbridgeodenuH!(::R3,...)  src/partialbridgenuH.jl:1-55; returns C */
static void rhs_dHplus(double t, const double *y, double *out, void *vc)
{ /* dH+(t,y,P) = B*y + (B*y)' - a */
    ode_ctx *c = (ode_ctx *)vc; int d = c->d;
    double B[D2], A[D2], By[D2];
    bo_aux_B(c->aux, d, c->apar, t, B);
    bo_aux_a(c->aux, d, c->mp, c->apar, t, A);
    mm(d, d, d, B, y, By);
    for (int j = 0; j < d; j++)
        for (int i = 0; i < d; i++) out[i + d * j] = By[i + d * j] + By[j + d * i] - A[i + d * j];
}
double bo_partialbridge_nuH(const double *tt, int N, int d, int mp, int m, int aux, const double *apar,
                            const double *L, const double *v, double eps, const double *Sigma,
                            double *nut, double *Ht)
{
    ode_ctx c = {aux, d, mp, m, apar, L};
    double Si[D2], LtSi[D2], H0[D2], Hp[D2], H[D2], nu[BO_MAXD], T1[D2], T2[D2];
    /* updatenuH+ :1-9   H = L'*inv(Sigma)*L + eps*I ; H+ = inv(H) ; nu = H+*L'*inv(Sigma)*v */
    bo_inv(m, Sigma, Si);
    mtm(d, m, m, L, Si, LtSi);                 /* L' * inv(Sigma)       d x m */
    mm(d, m, d, LtSi, L, H0);                  /* (L'*inv(Sigma)) * L   d x d */
    for (int k = 0; k < d; k++) H0[k + d * k] = H0[k + d * k] + eps;
    bo_inv(d, H0, Hp);
    mmt(d, d, m, Hp, L, T1);                   /* H+ * L'               d x m */
    mm(d, m, m, T1, Si, T2);                   /* (H+*L') * inv(Sigma)  d x m */
    mv(d, m, T2, v, nu);
    /* updateC :10-14   C = 0.5*dot(v, Sigma\v) + m/2*log(2pi) + 0.5*logdet(Sigma) */
    double sv[BO_MAXD];
    bo_solve(m, Sigma, v, sv);
    double C = 0.0;
    C += 0.5 * dotv(m, v, sv);
    C += m / 2.0 * log(2 * M_PI) + 0.5 * log(bo_det(m, Sigma));
    /* partialbridgeodenuH! :21-55 */
    bo_inv(d, Hp, H);
    memcpy(Ht + (size_t)(N - 1) * d * d, H, sizeof(double) * d * d);
    memcpy(nut + (size_t)(N - 1) * d, nu, sizeof(double) * d);
    for (int i = N - 2; i >= 0; i--) {
        double dt = tt[i] - tt[i + 1];
        kernelr3(rhs_dHplus, tt[i + 1], Hp, dt, d * d, &c, Hp);
        /* F = H*nu ; C += dC(t[i+1], (F,H), P)*dt,
         * dC = dot(beta,F) + 0.5*dot(F, a*F) - 0.5*tr(H*a)   (old H, old nu) */
        double F[BO_MAXD], beta[BO_MAXD], A[D2], aF[BO_MAXD], HA[D2];
        mv(d, d, H, nu, F);
        bo_aux_beta(aux, d, apar, tt[i + 1], beta);
        bo_aux_a(aux, d, mp, apar, tt[i + 1], A);
        mv(d, d, A, F, aF);
        mm(d, d, d, H, A, HA);
        double tr = HA[0];
        for (int k = 1; k < d; k++) tr += HA[k + d * k];
        C += (dotv(d, beta, F) + 0.5 * dotv(d, F, aF) - 0.5 * tr) * dt;
        kernelr3(rhs_F, tt[i + 1], nu, dt, d, &c, nu);
        memcpy(nut + (size_t)i * d, nu, sizeof(double) * d);
        bo_inv(d, Hp, H);
        memcpy(Ht + (size_t)i * d * d, H, sizeof(double) * d * d);
    }
    return C;
}

/* PartialBridge!  src/partialbridgen!.jl:7-56: in-place R3! (src/ode!.jl:21-29) for
 * nu' = b!(t,nu) (= B*nu+beta of the aux) and Sigma' = dP!(t,p) = B*p + (B*p)' - a, H = inv.(Sigma) */
void bo_partialbridge_inplace(const double *tt, int N, int d, int mp, int m, int aux, const double *apar,
                              const double *L, const double *v, double eps, const double *Sn,
                              double *nut, double *Ht)
{
    ode_ctx c = {aux, d, mp, m, apar, L};
    double G[D2], LtG[D2], H0[D2], S[D2], nu[BO_MAXD], T1[D2], T2[D2];
    int zero = 1;
    for (int k = 0; k < m * m; k++) if (Sn[k] != 0.0) zero = 0;
    if (zero) { /* :17  Gnoise = inv(eps()*one(Sn)) */
        memset(G, 0, sizeof(double) * m * m);
        for (int k = 0; k < m; k++) G[k + m * k] = 1.0 / 2.220446049250313e-16;
    } else bo_inv(m, Sn, G);
    mtm(d, m, m, L, G, LtG);
    mm(d, m, d, LtG, L, H0);
    for (int k = 0; k < d; k++) H0[k + d * k] = H0[k + d * k] + eps;
    bo_inv(d, H0, S);                           /* Sigma_t[end] = inv(L'*G*L + eps*I) */
    mmt(d, d, m, S, L, T1);
    mm(d, m, m, T1, G, T2);
    mv(d, m, T2, v, nu);                        /* nu_t[end] = Sigma_t[end]*L'*G*v     */
    double *St = (double *)malloc(sizeof(double) * (size_t)N * d * d);
    memcpy(St + (size_t)(N - 1) * d * d, S, sizeof(double) * d * d);
    memcpy(nut + (size_t)(N - 1) * d, nu, sizeof(double) * d);
    for (int i = N - 2; i >= 0; i--) {
        double dt = tt[i] - tt[i + 1];
        kernelr3(rhs_F, tt[i + 1], nu, dt, d, &c, nu);
        kernelr3(rhs_dHplus, tt[i + 1], S, dt, d * d, &c, S);
        memcpy(nut + (size_t)i * d, nu, sizeof(double) * d);
        memcpy(St + (size_t)i * d * d, S, sizeof(double) * d * d);
    }
    for (int i = 0; i < N; i++) bo_inv(d, St + (size_t)i * d * d, Ht + (size_t)i * d * d);  /* map!(inv, St, St) */
    free(St);
}

/* ------------------------------------------------------------------------------------------
 * LOOP A  sample!(W, Wiener())  src/wiener.jl:24-35 (SVector), :50-58 (scalar):
 *   W[1] = 0;  W[i] = W[i-1] + sqrt(tt[i]-tt[i-1])*randn()   (time-major, component-minor)
 * ------------------------------------------------------------------------------------------ */
void bo_wiener_sample(const double *tt, int N, int mp, uint64_t seed, uint32_t path, uint32_t iter,
                      double *W)
{
    for (int j = 0; j < mp; j++) W[j] = 0.0;
    double pr[2]; long have = -1;
    for (int i = 1; i < N; i++) {
        double rootdt = sqrt(tt[i] - tt[i - 1]);
        for (int j = 0; j < mp; j++) {
            int idx = (i - 1) * mp + j;
            if ((idx >> 1) != have) { bo_normal_pair(seed, path, iter, (uint32_t)(idx >> 1), pr); have = idx >> 1; }
            W[mp * i + j] = W[mp * (i - 1) + j] + rootdt * pr[idx & 1];
        }
    }
}

/* LOOP B (unguided)  solve!(::EulerMaruyama, Y, u, W, P)  src/euler.jl:135-152 */
void bo_solve_em(int model, int d, int mp, const double *par, const double *tt, int N,
                 const double *u, const double *W, double *X)
{
    double y[BO_MAXD], b[BO_MAXD], dw[BO_MAXD], s[BO_MAXD];
    memcpy(y, u, sizeof(double) * d);
    for (int i = 0; i < N - 1; i++) {
        memcpy(X + (size_t)i * d, y, sizeof(double) * d);
        bo_b(model, d, par, tt[i], y, b);
        for (int j = 0; j < mp; j++) dw[j] = W[mp * (i + 1) + j] - W[mp * i + j];
        bo_sigma_apply(model, d, mp, par, tt[i], y, dw, s);
        double dt = tt[i + 1] - tt[i];
        for (int k = 0; k < d; k++) y[k] = y[k] + b[k] * dt + s[k];
    }
    memcpy(X + (size_t)(N - 1) * d, y, sizeof(double) * d);     /* endpoint(y,P) = y :65 */
}

/* r((i,t),x,Po): src/guip.jl:193 | src/partialbridge.jl:57 | src/partialbridgenuH.jl:161 |
 * src/partialbridgen!.jl:67-70   (i is 0-based here) */
void bo_guided_r(const bo_proposal *P, int i, const double *x, double *r)
{
    int d = P->d, m = P->m;
    if (P->kind == BO_GUIDE_HV) { /* Hd[i] \ (V[i] - x) */
        double w[BO_MAXD];
        for (int k = 0; k < d; k++) w[k] = P->V[(size_t)i * d + k] - x[k];
        bo_solve(d, P->Hd + (size_t)i * d * d, w, r);
    } else if (P->kind == BO_GUIDE_LMMU) { /* L[i]'*M[i]*(v - mu[i] - L[i]*x) */
        const double *L = P->L + (size_t)i * m * d, *M = P->M + (size_t)i * m * m, *mu = P->mu + (size_t)i * m;
        double Lx[BO_MAXD], q[BO_MAXD], LtM[D2];
        mv(m, d, L, x, Lx);
        for (int k = 0; k < m; k++) q[k] = P->v[k] - mu[k] - Lx[k];
        mtm(d, m, m, L, M, LtM);
        mv(d, m, LtM, q, r);
    } else { /* H[i]*(nu[i] - x) */
        double w[BO_MAXD];
        for (int k = 0; k < d; k++) w[k] = P->nu[(size_t)i * d + k] - x[k];
        mv(d, d, P->H + (size_t)i * d * d, w, r);
    }
}

/* _b((i,t),x,Po): src/guip.jl:192 | src/partialbridge.jl:53-55 | src/partialbridgenuH.jl:157-159 |
 * src/partialbridgen!.jl:59-63 */
void bo_guided_drift(const bo_proposal *P, int i, const double *x, double *out)
{
    int d = P->d, m = P->m;
    double b[BO_MAXD], A[D2], g[BO_MAXD];
    double t = P->tt[i];
    bo_b(P->model, d, P->par, t, x, b);
    bo_a(P->model, d, P->mp, P->par, t, x, A);
    if (P->kind == BO_GUIDE_LMMU) { /* ((a*L')*M)*(v - mu - L*x) */
        const double *L = P->L + (size_t)i * m * d, *M = P->M + (size_t)i * m * m, *mu = P->mu + (size_t)i * m;
        double Lx[BO_MAXD], q[BO_MAXD], aLt[D2], aLtM[D2];
        mv(m, d, L, x, Lx);
        for (int k = 0; k < m; k++) q[k] = P->v[k] - mu[k] - Lx[k];
        mmt(d, d, m, A, L, aLt);
        mm(d, m, m, aLt, M, aLtM);
        mv(d, m, aLtM, q, g);
    } else { /* a*(Hd\(V-x))  |  a*(H*(nu-x)) */
        double r[BO_MAXD];
        bo_guided_r(P, i, x, r);
        mv(d, d, A, r, g);
    }
    for (int k = 0; k < d; k++) out[k] = b[k] + g[k];
}

/* LOOP B (guided)  solve!(::Euler, Y, u, W, P::Union{GuidedBridge,PartialBridge,PartialBridgeNuH})
 * src/euler.jl:247-268 ; endpoint src/euler.jl:241-242 (GuidedBridge) / :65 */
void bo_solve_guided(const bo_proposal *P, const double *u, const double *W, double *X)
{
    int d = P->d, mp = P->mp, N = P->N;
    double y[BO_MAXD], b[BO_MAXD], dw[BO_MAXD], s[BO_MAXD];
    memcpy(y, u, sizeof(double) * d);
    for (int i = 0; i < N - 1; i++) {
        memcpy(X + (size_t)i * d, y, sizeof(double) * d);
        bo_guided_drift(P, i, y, b);
        for (int j = 0; j < mp; j++) dw[j] = W[mp * (i + 1) + j] - W[mp * i + j];
        bo_sigma_apply(P->model, d, mp, P->par, P->tt[i], y, dw, s);
        double dt = P->tt[i + 1] - P->tt[i];
        for (int k = 0; k < d; k++) y[k] = y[k] + b[k] * dt + s[k];
    }
    if (P->kind == BO_GUIDE_HV) { /* norm(Hd[end],1) < eps() ? V[end] : y */
        double n1 = 0;
        for (int k = 0; k < d * d; k++) n1 += fabs(P->Hd[(size_t)(N - 1) * d * d + k]);
        if (n1 < 2.220446049250313e-16) memcpy(y, P->V + (size_t)(N - 1) * d, sizeof(double) * d);
    }
    memcpy(X + (size_t)(N - 1) * d, y, sizeof(double) * d);
}

/* LOOP C  llikelihood(::LeftRule, X, Po; skip)  constdiff branch only:
 * src/guip.jl:429-438 | src/partialbridge.jl:67-77 | src/partialbridgenuH.jl:171-181 :
 *     som += dot(_b(target) - _b(auxiliary), r) * (tt[i+1]-tt[i])
 * src/partialbridgen!.jl:81-97 (PartialBridge!):
 *     som += dot(bout, rout)*dt ; som -= dot(btout, rout)*dt */
double bo_llikelihood(const bo_proposal *P, const double *X, int skip)
{
    int d = P->d, N = P->N;
    double som = 0.0;
    double r[BO_MAXD], bt[BO_MAXD], ba[BO_MAXD], df[BO_MAXD];
    for (int i = 0; i < N - 1 - skip; i++) {
        const double *x = X + (size_t)i * d;
        double s = P->tt[i];
        bo_guided_r(P, i, x, r);
        bo_b(P->model, d, P->par, s, x, bt);
        bo_aux_b(P->aux, d, P->apar, s, x, ba);
        double dt = P->tt[i + 1] - P->tt[i];
        if (P->kind == BO_GUIDE_NUH_INPLACE) {
            som += dotv(d, bt, r) * dt;
            som -= dotv(d, ba, r) * dt;
        } else {
            for (int k = 0; k < d; k++) df[k] = bt[k] - ba[k];
            som += dotv(d, df, r) * dt;
        }
        if (!bo_constdiff(P->model)) {
            /* src/partialbridge.jl:79-84 (the one !constdiff branch of the reference that is well defined):
             *   H = L'*M*L (:58);  A = a((i,s),x,target) - a((i,s),auxiliary)
             *   som -= 0.5*tr(A*H)*dt;  som += 0.5*(r'*A*r)*dt */
            if (P->kind != BO_GUIDE_LMMU) return NAN;
            int m = P->m;
            const double *L = P->L + (size_t)i * m * d, *M = P->M + (size_t)i * m * m;
            double LtM[D2], H[D2], A[D2], At[D2], AH[D2], rA[BO_MAXD] = {0};
            mtm(d, m, m, L, M, LtM);
            mm(d, m, d, LtM, L, H);
            bo_a(P->model, d, P->mp, P->par, s, x, A);
            bo_aux_a(P->aux, d, P->mp, P->apar, s, At);
            for (int k = 0; k < d * d; k++) A[k] = A[k] - At[k];
            mm(d, d, d, A, H, AH);
            double trAH = AH[0];
            for (int k = 1; k < d; k++) trAH += AH[k + d * k];
            mtm(1, d, d, r, A, rA);                       /* r'*A */
            som -= 0.5 * trAH * dt;
            som += 0.5 * dotv(d, rA, r) * dt;
        }
    }
    return som;
}

static void mk_prop(bo_proposal *P, int kind, int N, int d, int mp, int m, int model, const double *par,
                    int aux, const double *apar, const double *tt,
                    const double *A1, const double *A2, const double *A3, const double *A4)
{
    memset(P, 0, sizeof(*P));
    P->kind = kind; P->N = N; P->d = d; P->mp = mp; P->m = m; P->model = model; P->par = par;
    P->aux = aux; P->apar = apar; P->tt = tt;
    if (kind == BO_GUIDE_HV) { P->Hd = A1; P->V = A2; }
    else if (kind == BO_GUIDE_LMMU) { P->L = A1; P->M = A2; P->mu = A3; P->v = A4; }
    else { P->nu = A1; P->H = A2; }
}

void bo_solve_guided_flat(int kind, int N, int d, int mp, int m, int model, const double *par,
                          int aux, const double *apar, const double *tt,
                          const double *A1, const double *A2, const double *A3, const double *A4,
                          const double *u, const double *W, double *X)
{
    bo_proposal P;
    mk_prop(&P, kind, N, d, mp, m, model, par, aux, apar, tt, A1, A2, A3, A4);
    bo_solve_guided(&P, u, W, X);
}
double bo_llikelihood_flat(int kind, int N, int d, int mp, int m, int model, const double *par,
                           int aux, const double *apar, const double *tt,
                           const double *A1, const double *A2, const double *A3, const double *A4,
                           const double *X, int skip)
{
    bo_proposal P;
    mk_prop(&P, kind, N, d, mp, m, model, par, aux, apar, tt, A1, A2, A3, A4);
    return bo_llikelihood(&P, X, skip);
}

/* ------------------------------------------------------------------------------------------
 * pCN Metropolis-Hastings chain  project_partialbridge/partialbridge_fitzhugh.jl:125-176
 *   init : W = sample(tt, Wiener()); solve!(Euler(), X, x0, W, Po); ll = llikelihood(X, Po, skip)
 *   iter : sample!(W2); Wo = rho*W + sqrt(1-rho^2)*W2; solve!(Xo, Wo); llo = llikelihood(Xo)
 *          if log(rand()) <= llo - ll: X<-Xo, W<-Wo, ll<-llo, acc+=1
 * RNG: iteration `it` (1-based; 0 = initial draw) uses Philox stream (seed, path, it).
 * ------------------------------------------------------------------------------------------ */
static void mcmc_chain(const bo_proposal *P, const double *x0, double rho, int iters, int skip,
                       uint64_t seed, uint32_t path, double *W, double *X, double *Wo, double *Xo,
                       double *W2, double *ll_trace, int *acc_trace, bo_mcmc_result *res)
{
    int N = P->N, d = P->d, mp = P->mp;
    bo_wiener_sample(P->tt, N, mp, seed, path, 0, W);
    bo_solve_guided(P, x0, W, X);
    double ll = bo_llikelihood(P, X, skip);
    long acc = 0;
    double sr = sqrt(1 - rho * rho);
    for (int it = 1; it <= iters; it++) {
        bo_wiener_sample(P->tt, N, mp, seed, path, (uint32_t)it, W2);
        for (int k = 0; k < N * mp; k++) Wo[k] = rho * W[k] + sr * W2[k];
        bo_solve_guided(P, x0, Wo, Xo);
        double llo = bo_llikelihood(P, Xo, skip);
        int accept = 0;
        if (bo_log(bo_uniform_accept(seed, path, (uint32_t)it)) <= llo - ll) {
            memcpy(X, Xo, sizeof(double) * (size_t)N * d);
            memcpy(W, Wo, sizeof(double) * (size_t)N * mp);
            ll = llo; accept = 1; acc += 1;
        }
        if (ll_trace) ll_trace[it - 1] = llo;
        if (acc_trace) acc_trace[it - 1] = accept;
    }
    res->acc = acc; res->ll = ll;
}

void bo_mcmc_flat(int kind, int N, int d, int mp, int m, int model, const double *par,
                  int aux, const double *apar, const double *tt,
                  const double *A1, const double *A2, const double *A3, const double *A4,
                  const double *x0, double rho, int iters, int skip, uint64_t seed, uint32_t path,
                  double *W, double *X, double *ll_trace, int *acc_trace, bo_mcmc_result *res)
{
    bo_proposal P;
    mk_prop(&P, kind, N, d, mp, m, model, par, aux, apar, tt, A1, A2, A3, A4);
    double *Wo = (double *)malloc(sizeof(double) * (size_t)N * mp), *W2 = (double *)malloc(sizeof(double) * (size_t)N * mp);
    double *Xo = (double *)malloc(sizeof(double) * (size_t)N * d);
    mcmc_chain(&P, x0, rho, iters, skip, seed, path, W, X, Wo, Xo, W2, ll_trace, acc_trace, res);
    free(Wo); free(W2); free(Xo);
}

double bo_ensemble_proposals(int kind, int N, int d, int mp, int m, int model, const double *par,
                             int aux, const double *apar, const double *tt,
                             const double *A1, const double *A2, const double *A3, const double *A4,
                             const double *x0, int npaths, uint32_t path0, uint64_t seed, uint32_t iter,
                             int threads, double *ll_out, double *Xlast_out)
{
    bo_proposal P;
    mk_prop(&P, kind, N, d, mp, m, model, par, aux, apar, tt, A1, A2, A3, A4);
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
#endif
    {
        double *W = (double *)malloc(sizeof(double) * (size_t)N * mp), *X = (double *)malloc(sizeof(double) * (size_t)N * d);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int p = 0; p < npaths; p++) {
            bo_wiener_sample(tt, N, mp, seed, path0 + (uint32_t)p, iter, W);   /* pass 1: sample!      */
            if (kind == BO_GUIDE_NONE) bo_solve_em(model, d, mp, par, tt, N, x0, W, X);
            else bo_solve_guided(&P, x0, W, X);                               /* pass 2: solve!       */
            if (ll_out) ll_out[p] = kind == BO_GUIDE_NONE ? 0.0 : bo_llikelihood(&P, X, 0); /* pass 3 */
            if (Xlast_out) memcpy(Xlast_out + (size_t)p * d, X + (size_t)(N - 1) * d, sizeof(double) * d);
        }
        free(W); free(X);
    }
    return (double)npaths * (N - 1);
}

double bo_ensemble_mcmc(int kind, int N, int d, int mp, int m, int model, const double *par,
                        int aux, const double *apar, const double *tt,
                        const double *A1, const double *A2, const double *A3, const double *A4,
                        const double *x0, double rho, int iters, int nchains, uint32_t path0,
                        uint64_t seed, int threads, double *ll_out, long *acc_out)
{
    bo_proposal P;
    mk_prop(&P, kind, N, d, mp, m, model, par, aux, apar, tt, A1, A2, A3, A4);
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
#endif
    {
        size_t nw = (size_t)N * mp, nx = (size_t)N * d;
        double *W = (double *)malloc(sizeof(double) * nw), *Wo = (double *)malloc(sizeof(double) * nw), *W2 = (double *)malloc(sizeof(double) * nw);
        double *X = (double *)malloc(sizeof(double) * nx), *Xo = (double *)malloc(sizeof(double) * nx);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int p = 0; p < nchains; p++) {
            bo_mcmc_result r;
            mcmc_chain(&P, x0, rho, iters, 0, seed, path0 + (uint32_t)p, W, X, Wo, Xo, W2, 0, 0, &r);
            if (ll_out) ll_out[p] = r.ll;
            if (acc_out) acc_out[p] = r.acc;
        }
        free(W); free(Wo); free(W2); free(X); free(Xo);
    }
    return (double)nchains * (iters + 1) * (N - 1);
}

/* mcnext! src/mclog.jl:48-56 (vector of d-vectors, m2 = outer(delta, x - m)); scalar case :31-38 */
/* ------------------------------------------------------------------------------------------
 * Joint Metropolis-Hastings over m chained GuidedBridge segments with a pCN move of the starting point and
 * mcnext! per iteration: the non-adaptive core of supplements/smoothing/smoothing.jl:99-213
 * (test/smoothing.jl:73-92 builds the same chain of proposals):
 *
 *   init (:101-106)  y = pi0.mu;  for i in 1:m: sample!(WW[i], Wiener); y = bridge!(XX[i], y, WW[i], Po[i])
 *   iteration (:165-213)
 *     y0o = pi0.mu + sqrt(rho_)*(rand(pi0) - pi0.mu) + sqrt(1-rho_)*(y0 - pi0.mu)            :172
 *     y = y0o;  for i in 1:m: sample!(WWo[i]); WWo[i] = sqrt(rho_)*WWo[i] + sqrt(1-rho_)*WW[i];
 *                             y = bridge!(XXo[i], y, WWo[i], Po[i])                          :177-184
 *     ll = sum_i llikelihood(XXo[i], Po[i]) - llikelihood(XX[i], Po[i])                      :187-190
 *     accept: y0 = y0o, XX <-> XXo, WW <-> WWo                                               :194-201
 *     for i in 1:m: mcstate[i] = mcnext!(mcstate[i], XX[i].yy)                               :211-213
 * with the weights (w_new, w_old) = (sqrt(rho_), sqrt(1-rho_)) supplied by the caller (the script draws
 * rho_ = exp(-alpha*randexp()) per iteration) and rand(pi0) = pi0.mu + C*randn (src/gaussian.jl:54, C = cholupper(Sigma)').
 * The accept test is log(U) <= ll, the form of partialbridge_fitzhugh.jl:161 (the script writes rand() < exp(ll):
 * the same event up to rounding at equality; exp is not reproducible across libm/ocml, log is the specification's).
 * Noise: segment i draws the Philox blocks offset by i*2^24 of stream 0; the start's normals are stream 2; U is stream 1.
 * All segments share d, m', the target model and the number of grid points N (as in the scripts: M + 1 points each).
 * ------------------------------------------------------------------------------------------ */
static void wiener_sample_blk(const double *tt, int N, int mp, uint64_t seed, uint32_t path, uint32_t iter, uint32_t blk0, double *W)
{
    for (int j = 0; j < mp; j++) W[j] = 0.0;
    double pr[2]; long have = -1;
    for (int i = 1; i < N; i++) {
        double rootdt = sqrt(tt[i] - tt[i - 1]);
        for (int j = 0; j < mp; j++) {
            int idx = (i - 1) * mp + j;
            if ((idx >> 1) != have) { bo_normal_pair(seed, path, iter, blk0 + (uint32_t)(idx >> 1), pr); have = idx >> 1; }
            W[mp * i + j] = W[mp * (i - 1) + j] + rootdt * pr[idx & 1];
        }
    }
}

/* props: m proposals.  Xall / Wall: [m][N][d] / [m][N][mp] current state out.  y0_out [d].  mean/m2/nstat: per segment
 * mcnext! states ([m][N][d], [m][N][d*d], one count) or NULL.  w_old/w_new: [iters] weights per iteration. */
void bo_smooth_mcmc(int m, const bo_proposal *props, const double *mu, const double *chol, const double *w_old, const double *w_new,
                    int iters, int skip, uint64_t seed, uint32_t path, double *Xall, double *Wall, double *y0_out,
                    double *ll_out /* [m] */, long *acc_out, double *mean, double *m2, long *nstat)
{
    const int N = props[0].N, d = props[0].d, mp = props[0].mp;
    const size_t nx = (size_t)N * d, nw = (size_t)N * mp;
    double *Xo = (double *)malloc(sizeof(double) * nx * m), *Wo = (double *)malloc(sizeof(double) * nw * m);
    double *W2 = (double *)malloc(sizeof(double) * nw);
    double *ll = ll_out, *llo = (double *)malloc(sizeof(double) * m);
    double y0[BO_MAXD], y0o[BO_MAXD], y[BO_MAXD];
    long acc = 0, ns = 0;
    for (int k = 0; k < d; k++) y0[k] = mu[k];
    memcpy(y, y0, sizeof(double) * d);
    for (int i = 0; i < m; i++) {
        wiener_sample_blk(props[i].tt, N, mp, seed, path, 0, (uint32_t)i << 24, Wall + nw * i);
        bo_solve_guided(&props[i], y, Wall + nw * i, Xall + nx * i);
        memcpy(y, Xall + nx * i + (size_t)(N - 1) * d, sizeof(double) * d);      /* bridge! returns yy[N]  src/euler.jl:267 */
        ll[i] = bo_llikelihood(&props[i], Xall + nx * i, skip);
    }
    if (mean) {   /* mcstart(XX[i].yy): zeros, count 0  src/mclog.jl:22 */
        memset(mean, 0, sizeof(double) * nx * m);
        memset(m2, 0, sizeof(double) * nx * d * m);
    }
    for (int it = 1; it <= iters; it++) {
        const double wo = w_old[it - 1], wn = w_new[it - 1];
        /* rand(pi0) = mu + C*xi, xi = the first d normals of stream 2 */
        double xi[BO_MAXD + 1], pr[2];
        for (int k = 0; k < d; k += 2) {
            bo_normal_pair_stream(seed, path, 2u, (uint32_t)it, (uint32_t)(k >> 1), pr);
            xi[k] = pr[0]; xi[k + 1] = pr[1];
        }
        for (int r = 0; r < d; r++) {
            double cz = chol[r] * xi[0];
            for (int c = 1; c < d; c++) cz += chol[r + d * c] * xi[c];
            const double z = mu[r] + cz;
            y0o[r] = mu[r] + wn * (z - mu[r]) + wo * (y0[r] - mu[r]);
        }
        memcpy(y, y0o, sizeof(double) * d);
        for (int i = 0; i < m; i++) {
            wiener_sample_blk(props[i].tt, N, mp, seed, path, (uint32_t)it, (uint32_t)i << 24, W2);
            const double *Wc = Wall + nw * i;
            double *Wp = Wo + nw * i;
            for (size_t k = 0; k < nw; k++) Wp[k] = wo * Wc[k] + wn * W2[k];
            bo_solve_guided(&props[i], y, Wp, Xo + nx * i);
            memcpy(y, Xo + nx * i + (size_t)(N - 1) * d, sizeof(double) * d);
            llo[i] = bo_llikelihood(&props[i], Xo + nx * i, skip);
        }
        double lls = 0.0;
        for (int i = 0; i < m; i++) lls += llo[i] - ll[i];
        if (bo_log(bo_uniform_accept(seed, path, (uint32_t)it)) <= lls) {
            acc += 1;
            memcpy(y0, y0o, sizeof(double) * d);
            memcpy(Xall, Xo, sizeof(double) * nx * m);
            memcpy(Wall, Wo, sizeof(double) * nw * m);
            memcpy(ll, llo, sizeof(double) * m);
        }
        if (mean) {
            long n_i = ns;
            for (int i = 0; i < m; i++) { n_i = ns; bo_mcnext(N, d, mean + nx * i, m2 + nx * d * i, &n_i, Xall + nx * i); }
            ns = n_i;
        }
    }
    memcpy(y0_out, y0, sizeof(double) * d);
    *acc_out = acc;
    if (nstat) *nstat = ns;
    free(Xo); free(Wo); free(W2); free(llo);
}

/* ctypes wrapper: the m proposals as flat arrays of equal shapes; A1..A4 hold the m guides back to back, tts the m grids */
void bo_smooth_mcmc_flat(int m, int kind, int N, int d, int mp, int mo, int model, const double *par, int aux, const double *apars, int napar,
                         const double *tts, const double *A1, const double *A2, const double *A3, const double *A4,
                         const double *mu, const double *chol, const double *w_old, const double *w_new, int iters, int skip,
                         uint64_t seed, uint32_t path, double *Xall, double *Wall, double *y0_out, double *ll_out, long *acc_out,
                         double *mean, double *m2, long *nstat)
{
    bo_proposal *ps = (bo_proposal *)malloc(sizeof(bo_proposal) * m);
    size_t s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    if (kind == BO_GUIDE_HV) { s1 = (size_t)N * d * d; s2 = (size_t)N * d; }
    else if (kind == BO_GUIDE_LMMU) { s1 = (size_t)N * mo * d; s2 = (size_t)N * mo * mo; s3 = (size_t)N * mo; s4 = mo; }
    else { s1 = (size_t)N * d; s2 = (size_t)N * d * d; }
    for (int i = 0; i < m; i++)
        mk_prop(&ps[i], kind, N, d, mp, mo, model, par, aux, apars + (size_t)napar * i, tts + (size_t)N * i,
                A1 + s1 * i, A2 + s2 * i, A3 ? A3 + s3 * i : NULL, A4 ? A4 + s4 * i : NULL);
    bo_smooth_mcmc(m, ps, mu, chol, w_old, w_new, iters, skip, seed, path, Xall, Wall, y0_out, ll_out, acc_out, mean, m2, nstat);
    free(ps);
}

/* cholupper(Hermitian(A))' for n <= 3: the LOWER factor C (C*C' = A) computed from A's UPPER triangle, as
 * rand(P::Gaussian) = P.mu + cholupper(P.Sigma)'*randn (src/gaussian.jl:54) needs it for pi0 = Gaussian(v, Hermitian(Hd))
 * (supplements/smoothing/smoothing.jl:96,153).  cholesky of a static matrix lives in StaticArrays (a dependency of
 * Bridge.jl that is not part of /root/reference; REQUIRE names it without a pin); its published closed forms are
 *   2x2: a = sqrt(A11); b = A12/a; c = sqrt(A22 - b^2)
 *   3x3: a11 = sqrt(A11); a12 = A12/a11; a22 = sqrt(A22 - a12^2); a13 = A13/a11; a23 = (A23 - a12*a13)/a22;
 *        a33 = sqrt(A33 - a13^2 - a23^2)                                  (U = [a11 a12 a13; 0 a22 a23; 0 0 a33]) */
void bo_chol_lower(int n, const double *A, double *C)
{
    memset(C, 0, sizeof(double) * n * n);
    if (n == 1) { C[0] = sqrt(A[0]); return; }
    if (n == 2) {
        double a = sqrt(A[0]), b = A[2] / a;
        C[0] = a; C[1] = b; C[3] = sqrt(A[3] - b * b);
        return;
    }
    if (n > 3) {
        /* n > 3 (StaticArrays' unrolled factorisation of the upper triangle, column by column): U[r,c] = (A[r,c] - sum_{i<r} U[i,r] U[i,c]) / U[r,r];
           C = U' -- compared at the large-d tolerance only (the closed forms below are the bit-exact ones) */
        for (int c = 0; c < n; c++)
            for (int r = 0; r <= c; r++) {
                double e = A[r + n * c];
                for (int i = 0; i < r; i++) e -= C[r + n * i] * C[c + n * i];   /* U[i,r] = C[r,i] */
                C[c + n * r] = r == c ? sqrt(e) : e / C[r + n * r];
            }
        return;
    }
    double a11 = sqrt(A[0]), a12 = A[3] / a11, a22 = sqrt(A[4] - a12 * a12);
    double a13 = A[6] / a11, a23 = (A[7] - a12 * a13) / a22, a33 = sqrt(A[8] - a13 * a13 - a23 * a23);
    C[0] = a11; C[1] = a12; C[2] = a13; C[4] = a22; C[5] = a23; C[8] = a33;
}

/* ------------------------------------------------------------------------------------------
 * The smoothing loop WITH adaptation: supplements/smoothing/smoothing.jl:75-213 for ONE chain.
 *   set-up (:75-109): H,v <- (HT, vT) [= gpupdate of the prior with the last observation, done by the caller];
 *     for i = m..1: Pt[i] = linearappr(Y0[i], P); Po[i] = GuidedBridge(tt_i, P, Pt[i], v, H); H,v = gpupdate(Po[i], L, Sigma, obs[i])
 *     pi0 = Gaussian(v, Hermitian(H)); y = pi0.mu; first paths.
 *   iteration `it` (:126-213):
 *     if adaptive && it < adaptmax && it % adaptit == 0 (:130-160):  the same backward pass with
 *        Y = mcstate[i] mean (or its moving average over j-hwindow..j+hwindow, :136,142), linearappr!(Pt, Y, P);
 *        pi0 <- Gaussian(v, Hermitian(H)); newblock = true; doaccept = (it == adaptit)
 *     y0o = newblock ? y0 : pi0.mu + w_new*(rand(pi0) - pi0.mu) + w_old*(y0 - pi0.mu)                 (:165-172)
 *     proposals, ll = sum_i llikelihood(XXo[i], Po[i]) - llikelihood(XX[i], Po[i])  -- both under the CURRENT Po (:187-190)
 *     accept if doaccept || log(U) <= ll; on accept newblock = false                                  (:193-202)
 *     mcnext! of every segment                                                                        (:211-213)
 * LinearAppr guides: the index-based Heun solver as restated in bo_gp_hv_heuni (the reference's kerneli cannot run as
 * committed, see there).  The moving average sums left to right and divides by the count (Statistics.mean of a short
 * vector of SVectors).  Noise streams as in bo_smooth_mcmc.
 * lna = 0: LinearAppr auxiliaries (initnu = :foci / :brown style, Y0 given); lna = 1: LinearNoiseAppr auxiliaries whose
 * deterministic paths are Y0; lna = 2: LinearNoiseAppr(tt_i, P, v, a, :backward) as the script builds them (:85), Y0 unused.
 * tts [m][N]; Y0 [m][N][d]; obs [m][mo]: obs[i] is the observation at the LEFT end of segment i (V.yy[i]).
 * Outputs as bo_smooth_mcmc, plus the final pi0 (mu_out [d], H_out [d*d]) and, if rows_out != NULL, the final guides
 * Hd [m][N][d*d], V [m][N][d] of the chain. */
static void la_pack(double *ap, int N, int d, int mp, int model, const double *par, const double *tt, const double *Y, int lna)
{
    ap[0] = (double)N;
    memcpy(ap + 1, tt, sizeof(double) * N);
    double *xx = ap + 1 + N, *B = xx + (size_t)N * d, *b = B + (size_t)N * d * d, *S = b + (size_t)N * d;
    if (lna) { bo_lna_coeffs(model, d, mp, par, tt, N, Y, xx, B, b, S); return; }
    memcpy(xx, Y, sizeof(double) * N * d);
    bo_linearappr(model, d, mp, par, tt, N, Y, B, b, S);
}
static void smooth_build(int m, int N, int d, int mp, int mo, int model, const double *par, const double *tts, const double *Yall,
                         const double *L, const double *Sigma, const double *obs, const double *HT, const double *vT,
                         double *apars, size_t napar, double *Hd, double *V, bo_proposal *props, double *mu, double *H0,
                         int lna /* 0: LinearAppr along Yall; 1: LinearNoiseAppr with Y = Yall; 2: LinearNoiseAppr(tt, P, v, a, :backward), smoothing.jl:85 */)
{
    double H[D2], v[BO_MAXD], Hn[D2], vn[BO_MAXD];
    memcpy(H, HT, sizeof(double) * d * d); memcpy(v, vT, sizeof(double) * d);
    for (int i = m - 1; i >= 0; i--) {
        const double *tt = tts + (size_t)N * i;
        double *ap = apars + napar * i, *Hi = Hd + (size_t)N * d * d * i, *Vi = V + (size_t)N * d * i;
        if (lna == 2) {
            double *Yb = (double *)malloc(sizeof(double) * N * d);
            bo_lna_path(model, d, par, tt, N, v, -1, Yb);
            la_pack(ap, N, d, mp, model, par, tt, Yb, 1);
            free(Yb);
        } else la_pack(ap, N, d, mp, model, par, tt, Yall + (size_t)N * d * i, lna);
        const double *xx = ap + 1 + N, *B = xx + (size_t)N * d, *b = B + (size_t)N * d * d, *S = b + (size_t)N * d;
        bo_gp_hv_heuni(tt, N, d, mp, xx, B, b, S, v, H, Hi, Vi);
        mk_prop(&props[i], BO_GUIDE_HV, N, d, mp, d, model, par, BO_AUX_LINEARAPPR, ap, tt, Hi, Vi, NULL, NULL);
        bo_gpupdate(d, mo, Hi, Vi, L, Sigma, obs + (size_t)mo * i, Hn, vn);
        memcpy(H, Hn, sizeof(double) * d * d); memcpy(v, vn, sizeof(double) * d);
    }
    memcpy(mu, v, sizeof(double) * d); memcpy(H0, H, sizeof(double) * d * d);
}

void bo_smooth_adaptive(int m, int N, int d, int mp, int mo, int model, const double *par, const double *tts, const double *Y0,
                        const double *L, const double *Sigma, const double *obs, const double *HT, const double *vT,
                        const double *w_old, const double *w_new, int iters, int adaptit, int adaptmax, int hwindow, int skip,
                        uint64_t seed, uint32_t path, double *Xall, double *Wall, double *y0_out, double *ll_out, long *acc_out,
                        double *mean, double *m2, double *mu_out, double *H_out, double *Hd_out, double *V_out, int lna)
{
    const size_t nx = (size_t)N * d, nw = (size_t)N * mp;
    const size_t napar = 1 + (size_t)N + nx + nx * d + nx + (size_t)N * d * mp;
    bo_proposal *props = (bo_proposal *)malloc(sizeof(bo_proposal) * m);
    double *apars = (double *)malloc(sizeof(double) * napar * m);
    double *Hd = (double *)malloc(sizeof(double) * nx * d * m), *V = (double *)malloc(sizeof(double) * nx * m);
    double *Ysm = (double *)malloc(sizeof(double) * nx * m);
    double *Xo = (double *)malloc(sizeof(double) * nx * m), *Wo = (double *)malloc(sizeof(double) * nw * m), *W2 = (double *)malloc(sizeof(double) * nw);
    double *llo = (double *)malloc(sizeof(double) * m), *ll = ll_out;
    double mu[BO_MAXD], H0[D2], chol[D2], y0[BO_MAXD], y0o[BO_MAXD], y[BO_MAXD];
    long acc = 0, ns = 0;
    int newblock = 0;
    smooth_build(m, N, d, mp, mo, model, par, tts, Y0, L, Sigma, obs, HT, vT, apars, napar, Hd, V, props, mu, H0, lna);
    bo_chol_lower(d, H0, chol);
    for (int k = 0; k < d; k++) y0[k] = mu[k];
    memcpy(y, y0, sizeof(double) * d);
    for (int i = 0; i < m; i++) {
        wiener_sample_blk(props[i].tt, N, mp, seed, path, 0, (uint32_t)i << 24, Wall + nw * i);
        bo_solve_guided(&props[i], y, Wall + nw * i, Xall + nx * i);
        memcpy(y, Xall + nx * i + (size_t)(N - 1) * d, sizeof(double) * d);
    }
    memset(mean, 0, sizeof(double) * nx * m);
    memset(m2, 0, sizeof(double) * nx * d * m);
    for (int it = 1; it <= iters; it++) {
        int doaccept = 0;
        if (adaptit > 0 && it < adaptmax && it % adaptit == 0) {
            for (int i = 0; i < m; i++)
                for (int j = 0; j < N; j++) {
                    double *dst = Ysm + nx * i + (size_t)j * d;
                    const double *xx = mean + nx * i;
                    if (hwindow <= 0) { memcpy(dst, xx + (size_t)j * d, sizeof(double) * d); continue; }
                    int lo = j - hwindow < 0 ? 0 : j - hwindow, hi = j + hwindow > N - 1 ? N - 1 : j + hwindow;
                    for (int k = 0; k < d; k++) {
                        double sacc = xx[(size_t)lo * d + k];
                        for (int l = lo + 1; l <= hi; l++) sacc += xx[(size_t)l * d + k];
                        dst[k] = sacc / (double)(hi - lo + 1);
                    }
                }
            smooth_build(m, N, d, mp, mo, model, par, tts, Ysm, L, Sigma, obs, HT, vT, apars, napar, Hd, V, props, mu, H0, lna ? 1 : 0);   /* Pt.Y.yy[:] = xx (:136-139) */
            bo_chol_lower(d, H0, chol);
            newblock = 1;
            if (it == adaptit) doaccept = 1;
        }
        const double wo = w_old[it - 1], wn = w_new[it - 1];
        if (newblock) memcpy(y0o, y0, sizeof(double) * d);
        else {
            double xi[BO_MAXD + 1], pr[2];
            for (int k = 0; k < d; k += 2) {
                bo_normal_pair_stream(seed, path, 2u, (uint32_t)it, (uint32_t)(k >> 1), pr);
                xi[k] = pr[0]; xi[k + 1] = pr[1];
            }
            for (int r = 0; r < d; r++) {
                double cz = chol[r] * xi[0];
                for (int c = 1; c < d; c++) cz += chol[r + d * c] * xi[c];
                const double z = mu[r] + cz;
                y0o[r] = mu[r] + wn * (z - mu[r]) + wo * (y0[r] - mu[r]);
            }
        }
        memcpy(y, y0o, sizeof(double) * d);
        double lls = 0.0;
        for (int i = 0; i < m; i++) {
            wiener_sample_blk(props[i].tt, N, mp, seed, path, (uint32_t)it, (uint32_t)i << 24, W2);
            const double *Wc = Wall + nw * i;
            double *Wp = Wo + nw * i;
            for (size_t k = 0; k < nw; k++) Wp[k] = wo * Wc[k] + wn * W2[k];
            bo_solve_guided(&props[i], y, Wp, Xo + nx * i);
            memcpy(y, Xo + nx * i + (size_t)(N - 1) * d, sizeof(double) * d);
            llo[i] = bo_llikelihood(&props[i], Xo + nx * i, skip);
            ll[i] = bo_llikelihood(&props[i], Xall + nx * i, skip);     /* the current path under the CURRENT proposal (:189) */
        }
        for (int i = 0; i < m; i++) lls += llo[i] - ll[i];
        if (doaccept || bo_log(bo_uniform_accept(seed, path, (uint32_t)it)) <= lls) {
            acc += 1;
            memcpy(y0, y0o, sizeof(double) * d);
            memcpy(Xall, Xo, sizeof(double) * nx * m);
            memcpy(Wall, Wo, sizeof(double) * nw * m);
            memcpy(ll, llo, sizeof(double) * m);
            newblock = 0;
        }
        for (int i = 0; i < m; i++) { long n_i = ns; bo_mcnext(N, d, mean + nx * i, m2 + nx * d * i, &n_i, Xall + nx * i); }
        ns += 1;
    }
    memcpy(y0_out, y0, sizeof(double) * d);
    *acc_out = acc;
    memcpy(mu_out, mu, sizeof(double) * d); memcpy(H_out, H0, sizeof(double) * d * d);
    if (Hd_out) memcpy(Hd_out, Hd, sizeof(double) * nx * d * m);
    if (V_out) memcpy(V_out, V, sizeof(double) * nx * m);
    free(props); free(apars); free(Hd); free(V); free(Ysm); free(Xo); free(Wo); free(W2); free(llo);
}

void bo_mcnext(int n_entries, int d, double *mean, double *m2, long *n, const double *x)
{
    long nn = *n;
    for (int i = 0; i < n_entries; i++) {
        double delta[BO_MAXD];
        double *mi = mean + (size_t)i * d, *m2i = m2 + (size_t)i * d * d;
        const double *xi = x + (size_t)i * d;
        for (int k = 0; k < d; k++) { delta[k] = xi[k] - mi[k]; mi[k] += delta[k] / (nn + 1); }
        for (int c = 0; c < d; c++)
            for (int r = 0; r < d; r++) m2i[r + d * c] += delta[r] * (xi[c] - mi[c]);
    }
    *n = nn + 1;
}

/* innovations!(::EulerMaruyama, W, Y, P)  src/euler.jl:358-376: the inverse map X -> W,
 *   w += inv(sigma(t_i, y_i)) * (y_{i+1} - y_i - _b((i,t_i), y_i, P)*(t_{i+1}-t_i)),  square sigma only.
 * P may be the unguided target (kind NONE) or a guided proposal. */
void bo_innovations_flat(int kind, int N, int d, int mp, int m, int model, const double *par,
                         int aux, const double *apar, const double *tt,
                         const double *A1, const double *A2, const double *A3, const double *A4,
                         const double *X, double *W)
{
    bo_proposal P;
    double S[D2], Sinv[D2], w[BO_MAXD], b[BO_MAXD], df[BO_MAXD], inc[BO_MAXD];
    if (kind != BO_GUIDE_NONE) mk_prop(&P, kind, N, d, mp, m, model, par, aux, apar, tt, A1, A2, A3, A4);
    model_sigma_mat(model, d, mp, par, tt[0], X, S);
    if (model == BO_MODEL_LORENZ || model == BO_MODEL_FHN2) {   /* inv(::SDiagonal) = SDiagonal(inv.(diag)) */
        memset(Sinv, 0, sizeof(double) * d * d);
        for (int k = 0; k < d; k++) Sinv[k + d * k] = 1.0 / S[k + d * k];
    } else bo_inv(d, S, Sinv);
    for (int k = 0; k < d; k++) w[k] = 0.0;
    for (int i = 0; i < N - 1; i++) {
        const double *y = X + (size_t)i * d, *yn = X + (size_t)(i + 1) * d;
        memcpy(W + (size_t)i * d, w, sizeof(double) * d);
        if (kind == BO_GUIDE_NONE) bo_b(model, d, par, tt[i], y, b);
        else bo_guided_drift(&P, i, y, b);
        double dt = tt[i + 1] - tt[i];
        for (int k = 0; k < d; k++) df[k] = yn[k] - y[k] - b[k] * dt;
        mv(d, d, Sinv, df, inc);
        for (int k = 0; k < d; k++) w[k] = w[k] + inc[k];
    }
    memcpy(W + (size_t)(N - 1) * d, w, sizeof(double) * d);
}


/* girsanov(X, P, Pt): src/diffusion.jl:109-123.  P, Pt: the same model type with parameter sets
 * par / par_t, or Pt = Wiener (b = 0, src/wiener.jl:143-145) when par_t == NULL (test/guip.jl:72).
 * Gamma(s,x,P) = inv(a(s,x,P)) (src/types.jl:33); for the SDiagonal-sigma models (src/Models.jl:19,57)
 * a stays diagonal and is inverted entry by entry. */
double bo_girsanov(int model, int d, int mp, const double *par, const double *par_t,
                   const double *tt, int N, const double *X)
{
    double A[D2], G[D2], B[BO_MAXD], Bt[BO_MAXD], df[BO_MAXD], g[BO_MAXD];
    bo_a(model, d, mp, par, 0.0, X, A);
    if (model == BO_MODEL_LORENZ || model == BO_MODEL_FHN2) {
        memset(G, 0, sizeof(double) * d * d);
        for (int k = 0; k < d; k++) G[k + d * k] = 1.0 / A[k + d * k];
    } else bo_inv(d, A, G);
    double som = 0.0;
    for (int i = 0; i < N - 1; i++) {
        const double *x = X + (size_t)i * d, *xn = X + (size_t)(i + 1) * d;
        bo_b(model, d, par, tt[i], x, B);
        if (par_t) bo_b(model, d, par_t, tt[i], x, Bt);
        else for (int k = 0; k < d; k++) Bt[k] = 0.0;
        for (int k = 0; k < d; k++) df[k] = B[k] - Bt[k];
        mv(d, d, G, df, g);                                   /* DeltaBG = Gamma*(B - Bt) */
        double dt = tt[i + 1] - tt[i], dot = 0.0;
        for (int k = 0; k < d; k++) {
            double inc = (xn[k] - x[k]) - (0.5 * (B[k] + Bt[k])) * dt;
            dot = k == 0 ? g[0] * inc : dot + g[k] * inc;
        }
        som += dot;
    }
    return som;
}

/* r((i,t),x,Po) and _b((i,t),x,Po) at one point, flat arguments for ctypes */
void bo_guided_terms_flat(int kind, int N, int d, int mp, int m, int model, const double *par,
                          int aux, const double *apar, const double *tt,
                          const double *A1, const double *A2, const double *A3, const double *A4,
                          int i, const double *x, double *r_out, double *drift_out)
{
    bo_proposal P;
    mk_prop(&P, kind, N, d, mp, m, model, par, aux, apar, tt, A1, A2, A3, A4);
    if (r_out) bo_guided_r(&P, i, x, r_out);
    if (drift_out) bo_guided_drift(&P, i, x, drift_out);
}
