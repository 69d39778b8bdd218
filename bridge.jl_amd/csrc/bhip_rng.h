// bhip_rng.h -- RNG specifications "bhip-philox-v4" (the default: inverse distribution function, one normal per 32-bit word),
//               "bhip-philox-v3" (Box-Muller, 40 + 24 bits per pair) and "bhip-philox-v2" (Box-Muller, 53 + 53 bits), host + device.
//
// Replaces the reference's global randn() (src/wiener.jl:31,44,55), which is not reproducible
// outside Julia, with a counter-based generator whose output depends only on
// (seed, global path id, iteration, normal index): results are independent of GPU count and
// launch geometry, and a chain can be resumed from its counters alone.
//
//   Philox4x32-10 (the generator behind rocRAND's default PHILOX4_32_10; Random123 KAT vectors in
//   tests/), key = (seed_lo, seed_hi), counter = (path, stream, iter, call).
//   stream 0: call q -> FOUR standard normals 4q .. 4q+3 by two Box-Muller transforms, one per half of the 128 output
//             bits.  Half s (words a = r[2s], b = r[2s+1]) = "pair" h = 2q + s -> normals 2h (cos branch), 2h+1 (sin):
//               u1 = (K40 + 1)*2^-40 in (0,1],  K40 = (b >> 24)*2^32 + a      (40 bits: radius up to 7.45)
//               u2 = K24*2^-24 in [0,1),        K24 = b & 0xffffff            (24 bits of angle)
//   stream 1: call 0 -> the Metropolis-Hastings uniform U = (bits53(r0,r1)+1)*2^-53
//   stream 2: normals of the pCN move of the starting point (multi-segment chains, bhip_segchains_*), same construction
//   Pairs of stream 0 are offset by segment*2^24 in multi-segment chains (KArgs::blk0): one noise stream per segment.
//
// -2*log(u1) and sin/cos(2*pi*u2) are built from integer operations, +, -, *, fma and two small constant
// tables (bhip_rng_tables.h, generated correctly rounded by scripts/gen_rng_tables.py), so that every host and
// device evaluates bit-identical normals (no libm / ocml dependence).
//
// v4 (round 5, the default): the same generator, counters and call layout -- call q of stream 0 still gives the normals 4q .. 4q+3,
// pair h = half h & 1 of call h >> 1 -- but word r[j] of the call IS normal 4q + j, through a piecewise polynomial inverse of the
// normal distribution function (icdf_normal below, table bhip_icdf_table.h from scripts/gen_icdf_table.py):
//     v = 2*(w mod 2^31) + 1 (odd),  upper-tail probability p = v*2^-33,  d = (double)v,  row R = (highword(d) >> 17) & 255
//     (octave of p and its eighth),  |z| = c0 + d*(c1 + d*(c2 + d*(c3 + d*c4))) in four fma,  sign = bit 31 of w.
// 32 bits per normal (v3: 40 + 24 per pair), |z| <= 6.34 (mass beyond: 2.3e-10), 2^32 equiprobable values; the polynomial is within
// 3.7e-9 of -Phi^-1(p) (Kolmogorov distance of the marginal to N(0,1) <= 1.3e-9 + 2^-33).  ~9 VALU + two LDS reads per normal on top
// of the Philox share instead of ~33: the reference's randn is a ziggurat (a handful of instructions); Box-Muller's log / sqrt /
// sincos were the first limiter of every kernel but the memory-bound headline.  Exact integer -> double conversion
// and fma only: host and device agree bit for bit as before.
//
// v1 -> v2 (round 2): table-driven argument reduction for the log and the rotation instead of a division + 11-term series
// and 7 + 8 Taylor terms: ~147 -> ~119 VALU instructions per Philox call (then: two normals).
// v2 -> v3 (round 3): v2 spent all 128 bits of a Philox call on ONE Box-Muller pair (53 + 53 bits); the ten Philox rounds
// (49 instructions) were the largest single item of every path kernel.  v3 draws two pairs per call -- what rocRAND's own
// single-precision normals do with the four words -- with 40 bits of radius and 24 bits of angle each: ~43 VALU per normal
// instead of ~60.  What the split means for the law: |z| <= sqrt(80 ln 2) = 7.45 (mass beyond: 9e-14 -- v2: 8.6 sigma; a
// launch of 2.6e8 normals expects 2e-5 draws out there), and the angle lives on a 2^24 grid (Kolmogorov distance of the
// marginal to N(0,1) <= 2^-24, invisible below ~1e14 samples).  log / sqrt / sin / cos are the v2 functions, unchanged.
#pragma once
#include <stdint.h>
#include "bhip_rng_tables.h"
#include "bhip_icdf_table.h"
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BHIP_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define BHIP_HD static inline
#endif

namespace bhip {

struct u32x4 { uint32_t x, y, z, w; };

// a ^ b ^ c: ONE instruction on gfx950 (v_bitop3_b32, truth table 0x96) -- the compiler pairs the xors of a Philox round into two
// v_xor_b32 each otherwise (35 instead of 20 per call)
BHIP_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(__has_builtin) && !defined(BHIP_NO_BITOP3)   /* (BHIP_NO_BITOP3: A/B builds, scripts/noise_rate_probe.hip) */
#if __has_builtin(__builtin_amdgcn_bitop3_b32)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
#else
    return a ^ b ^ c;
#endif
}

BHIP_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32x32->64 product per multiplier (v_mad_u64_u32 on gfx950) instead of mul_hi + mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = xor3(hi1, c1, k0), n2 = xor3(hi0, c3, k1);
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

BHIP_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

#if defined(__HIP_DEVICE_COMPILE__)
// Device forms of the fixed-range operations below.  They produce the SAME bits as the portable
// expressions (checked exhaustively-at-random on the GPU by tests/rng_device_forms.hip) with fewer VALU
// instructions: the noise is the largest share of the proposal kernels' issue slots.
//   (a >> 11) * 2^-53 : k = a >> 11 = kh*2^32 + kl; a double whose mantissa field holds kl (kh) under a fixed
//   exponent equals c + kl*2^-53 (c + kh*2^-21) exactly, so two subtractions and one exact addition replace
//   the integer->double conversions.
BHIP_HD double u53_bits(uint32_t lo, uint32_t hi, double lo_magic)
{
    const uint32_t kl = (hi << 21) | (lo >> 11), kh = hi >> 11;
    union { uint64_t u; double d; } a, b;
    a.u = 0x3FE0000000000000ULL | kl;             // 0.5   + kl * 2^-53
    b.u = 0x41E0000000000000ULL | kh;             // 2^31  + kh * 2^-21
    return (b.d - 0x1.0p31) + (a.d - lo_magic);   // both differences and the sum are exact
}
BHIP_HD double u53_open0(uint32_t lo, uint32_t hi) { return u53_bits(lo, hi, 0.5 - 0x1.0p-53); }   // (k+1)*2^-53 in (0,1]
BHIP_HD double u53_open1(uint32_t lo, uint32_t hi) { return u53_bits(lo, hi, 0.5); }               // k*2^-53 in [0,1)
// (K40 + 1)*2^-40, K40 = (b >> 24)*2^32 + a: the 40 bits go to the top of the mantissa field of a double in [1,2), so that
// it reads 1 + K40*2^-40; the subtraction of 1 - 2^-40 is exact (the result has at most 41 significant bits).
BHIP_HD double u40_open0(uint32_t a, uint32_t b)
{
    union { uint64_t u; double d; } v;
    v.u = 0x3FF0000000000000ULL | ((uint64_t)(b >> 24) << 44) | ((uint64_t)a << 12);
    return v.d - (1.0 - 0x1.0p-40);
}
// sqrt(x) as the compiler expands it (reciprocal-square-root seed, Newton steps on fma, final residual
// correction = correctly rounded), WITHOUT the exponent pre-scaling and special-value fix-ups that only matter
// outside the range used here: x in [0, 1500].
BHIP_HD double sqrt_fixed_range(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x * y, h0 = y * 0.5;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, x);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, x);
    const double g3 = __builtin_fma(d1, h1, g2);
    return x == 0.0 ? 0.0 : g3;
}
#else
BHIP_HD double u53_open0(uint32_t lo, uint32_t hi)   // (0,1]
{
    const uint64_t a = ((uint64_t)hi << 32) | lo;
    return (double)((a >> 11) + 1) * 0x1.0p-53;
}
BHIP_HD double u53_open1(uint32_t lo, uint32_t hi)   // [0,1)
{
    const uint64_t a = ((uint64_t)hi << 32) | lo;
    return (double)(a >> 11) * 0x1.0p-53;
}
BHIP_HD double u40_open0(uint32_t a, uint32_t b)   // (0,1]
{
    const uint64_t k = ((uint64_t)(b >> 24) << 32) | a;
    return (double)(k + 1) * 0x1.0p-40;
}
BHIP_HD double sqrt_fixed_range(double x) { return __builtin_sqrt(x); }
#endif

// Where the two constant tables are read from.  TabConst: constant memory on the device (per-lane loads, served by
// the vector L1 / L2 -- used by the kernels that draw few normals), plain arrays on the host.  TabLDS: a copy in
// LDS (the producer waves of bhip_pc_kernel.h and the d = 16/32 tile kernel, which draw normals at full rate).
alignas(16) static const double logtab_host[2 * BHIP_LOGTAB_N] = BHIP_LOGTAB_INIT;
alignas(16) static const double sctab_host[2 * BHIP_SCTAB_N] = BHIP_SCTAB_INIT;
#if defined(__HIPCC__)
alignas(16) static __device__ const double logtab_dev[2 * BHIP_LOGTAB_N] = BHIP_LOGTAB_INIT;
alignas(16) static __device__ const double sctab_dev[2 * BHIP_SCTAB_N] = BHIP_SCTAB_INIT;
typedef double rng_d2v __attribute__((ext_vector_type(2)));
#endif
struct TabConst {
    BHIP_HD void lg(uint32_t k, double &A, double &B) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const rng_d2v v = *reinterpret_cast<const rng_d2v *>(logtab_dev + 2 * k);
        A = v.x; B = v.y;
#else
        A = logtab_host[2 * k]; B = logtab_host[2 * k + 1];
#endif
    }
    BHIP_HD void sc(uint32_t j, double &c, double &s) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const rng_d2v v = *reinterpret_cast<const rng_d2v *>(sctab_dev + 2 * j);
        c = v.x; s = v.y;
#else
        c = sctab_host[2 * j]; s = sctab_host[2 * j + 1];
#endif
    }
};
#if defined(__HIPCC__)
constexpr int RNG_TAB_DOUBLES = 2 * (BHIP_LOGTAB_N + BHIP_SCTAB_N);   // 322 doubles = 2576 bytes of LDS
struct TabLDS {
    typedef const __attribute__((address_space(3))) rng_d2v *lds_t;
    lds_t lgp, scp;
    // `base` points to RNG_TAB_DOUBLES doubles of LDS (16-byte aligned), filled by load() + a barrier
    __device__ __forceinline__ explicit TabLDS(double *base)
        : lgp((lds_t)(__attribute__((address_space(3))) double *)base),
          scp((lds_t)(__attribute__((address_space(3))) double *)(base + 2 * BHIP_LOGTAB_N)) {}
    // cooperative fill by `nthreads` threads (thread index tid); the caller synchronises afterwards
    static __device__ __forceinline__ void load(double *base, int tid, int nthreads)
    {
        for (int q = tid; q < 2 * BHIP_LOGTAB_N; q += nthreads) base[q] = logtab_dev[q];
        for (int q = tid; q < 2 * BHIP_SCTAB_N; q += nthreads) base[2 * BHIP_LOGTAB_N + q] = sctab_dev[q];
    }
    __device__ __forceinline__ void lg(uint32_t k, double &A, double &B) const { const rng_d2v v = lgp[k]; A = v.x; B = v.y; }
    __device__ __forceinline__ void sc(uint32_t j, double &c, double &s) const { const rng_d2v v = scp[j]; c = v.x; s = v.y; }
};
#endif

// ---- specification v4: the table of the piecewise inverse distribution function, BHIP_ICDF_ROWS rows {c0 .. c4}.
// IcdfConst reads it from constant memory (per-lane loads) / the host array; IcdfLDS from a copy in LDS laid out in three planes --
// {c0, c1}[R], {c2, c3}[R], {c4, -}[R], all with a 16-byte stride: ONE address register (row * 16) serves the three reads, and the
// 16-byte reads of neighbouring rows fall into different banks (the sixteen most probable rows, octaves p >= 1/8, hit sixteen
// different bank groups).  The TYPE of the accessor carries the specification.
alignas(16) static const double icdf_host[5 * BHIP_ICDF_ROWS] = BHIP_ICDF_INIT;
#if defined(__HIPCC__)
alignas(16) static __device__ const double icdf_dev[5 * BHIP_ICDF_ROWS] = BHIP_ICDF_INIT;
#endif
struct IcdfConst {
    static constexpr int NOISE_SPEC = 4;
    BHIP_HD void row(uint32_t R, double &c0, double &c1, double &c2, double &c3, double &c4) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const double *c = icdf_dev + 5 * R;
#else
        const double *c = icdf_host + 5 * R;
#endif
        c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4];
    }
};
#if defined(__HIPCC__)
constexpr int ICDF_TAB_DOUBLES = 6 * BHIP_ICDF_ROWS;   // 1536 doubles = 12 288 bytes of LDS (the third plane half empty)
struct IcdfLDS {
    static constexpr int NOISE_SPEC = 4;
    typedef const __attribute__((address_space(3))) rng_d2v *lds2_t;
    typedef const __attribute__((address_space(3))) double *lds1_t;
    lds2_t pa;   // {c0, c1}[R]; {c2, c3}[R] follows at + BHIP_ICDF_ROWS, {c4, unused}[R] at + 2*BHIP_ICDF_ROWS
    // `base` points to ICDF_TAB_DOUBLES doubles of LDS (16-byte aligned), filled by load() + a barrier
    __device__ __forceinline__ explicit IcdfLDS(double *base) : pa((lds2_t)(__attribute__((address_space(3))) double *)base) {}
    static __device__ __forceinline__ void load(double *base, int tid, int nthreads)
    {
        for (int q = tid; q < 5 * BHIP_ICDF_ROWS; q += nthreads) {
            const int R = q / 5, k = q - 5 * R;
            base[(k >> 1) * (2 * BHIP_ICDF_ROWS) + 2 * R + (k & 1)] = icdf_dev[q];
        }
    }
    __device__ __forceinline__ void row(uint32_t R, double &c0, double &c1, double &c2, double &c3, double &c4) const
    {
        const rng_d2v a = pa[R], b = pa[BHIP_ICDF_ROWS + R];
        c4 = ((lds1_t)pa)[2 * (2 * BHIP_ICDF_ROWS + R)];   // an 8-byte read at the same row * 16
        c0 = a.x; c1 = a.y; c2 = b.x; c3 = b.y;
    }
};
// IcdfLDSHot: only the 128 rows of the octaves p >= 2^-17 in LDS (5 KB, the same three planes) -- for the kernels whose LDS has no room
// for the whole table (the d = 16 / 32 tile kernel: fragment matrices + step buffers fill all but 2.7 KB of half a CU).  A word from a
// farther octave comes up once in 2^16 draws: row() then reads a valid but WRONG row and reports the lane `cold`; normal_quad redraws the
// call's four normals from the constant-memory table (IcdfConst) for the whole wave behind ONE wave-uniform branch per call (taken by
// 0.4 % of the calls; every lane gets the value the full table gives, whichever way it went).
constexpr int ICDF_HOT_ROWS = 128, ICDF_HOT_FIRST = 120;   // rows 120 .. 247 = octaves e = 15 .. 0 (row R holds octave (30 - (R >> 3)) mod 32)
constexpr int ICDF_HOT_DOUBLES = 5 * ICDF_HOT_ROWS;       // 640 doubles = 5 120 bytes of LDS
struct IcdfLDSHot {
    static constexpr int NOISE_SPEC = 4;
    static constexpr bool HOT_ONLY = true;
    typedef const __attribute__((address_space(3))) rng_d2v *lds2_t;
    typedef const __attribute__((address_space(3))) double *lds1_t;
    lds2_t pa;
    bool *cold;   // (a register of the caller: set when a lane's word fell outside the rows held here)
    __device__ __forceinline__ IcdfLDSHot(double *base, bool *cold_) : pa((lds2_t)(__attribute__((address_space(3))) double *)base), cold(cold_) {}
    static __device__ __forceinline__ void load(double *base, int tid, int nthreads)
    {
        for (int q = tid; q < 5 * ICDF_HOT_ROWS; q += nthreads) {
            const int R = q / 5, k = q - 5 * R;
            base[k < 4 ? (k >> 1) * (2 * ICDF_HOT_ROWS) + 2 * R + (k & 1) : 4 * ICDF_HOT_ROWS + R] = icdf_dev[5 * ICDF_HOT_FIRST + q];
        }
    }
    __device__ __forceinline__ void row(uint32_t R, double &c0, double &c1, double &c2, double &c3, double &c4) const
    {
        const uint32_t h = R - (uint32_t)ICDF_HOT_FIRST;
        *cold = *cold || h >= (uint32_t)ICDF_HOT_ROWS;
        const uint32_t hc = h & (uint32_t)(ICDF_HOT_ROWS - 1);
        const rng_d2v a = pa[hc], b = pa[ICDF_HOT_ROWS + hc];
        c4 = ((lds1_t)pa)[4 * ICDF_HOT_ROWS + hc];
        c0 = a.x; c1 = a.y; c2 = b.x; c3 = b.y;
    }
};
// the larger of the two LDS tables: what a kernel that can draw under every specification reserves
constexpr int RNG_LDS_DOUBLES = ICDF_TAB_DOUBLES > RNG_TAB_DOUBLES ? ICDF_TAB_DOUBLES : RNG_TAB_DOUBLES;
#endif

// one standard normal from one 32-bit word (specification v4, see the head of this file)
template <class Tab>
BHIP_HD double icdf_normal(const Tab &tab, uint32_t w)
{
    union { double d; uint64_t u; } b;
    b.d = (double)((w << 1) | 1u);                 // exact: an odd integer below 2^32
    const double d = b.d;
    const uint32_t R = ((uint32_t)(b.u >> 32) >> 17) & 255u;
    double c0, c1, c2, c3, c4;
    tab.row(R, c0, c1, c2, c3, c4);
    double q = fma_(c4, d, c3);
    q = fma_(q, d, c2);
    q = fma_(q, d, c1);
    q = fma_(q, d, c0);
    // |q| with the sign bit of w: copysign from a double whose high word is w (one v_bfi_b32 on the device)
    b.u = (uint64_t)w << 32;
    return __builtin_copysign(q, b.d);
}

// L = -2*ln(x) for x in (0,1], normal doubles.  x = 2^e * m0, m0 in [1,2);  k = round(128*m0) - 128 selects the
// table row {A, B}: s = fma(m0, A, 2) = -2*(m/c - 1) with |s| <= 2^-7, and
//     L = e'*(-2 ln2) + B + (s + s^2/4 + s^3/12 + s^4/32 + s^5/80 + s^6/192 + s^7/448)      (= -2*log1p(-s/2))
// The rows k = 0 and k = 128 have c = 1, B = 0: L(1) = 0 exactly and x -> 1 suffers no cancellation (L >= 0 always,
// so the square root below never sees a negative argument).
template <class Tab>
BHIP_HD double det_m2log(double x, const Tab &tab)
{
    union { double d; uint64_t u; } v;
    v.d = x;
    const uint32_t hi = (uint32_t)(v.u >> 32);
    const uint32_t k = (((hi >> 12) & 0xffu) + 1u) >> 1;
    const int e = (int)(hi >> 20) - 1023 + (k > 53u ? 1 : 0);   // rows k > 53 work on m = m0/2
    v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double A, B;
    tab.lg(k, A, B);
    const double s = fma_(v.d, A, 2.0);
    double q = 1.0 / 448.0;
    q = fma_(q, s, 1.0 / 192.0);
    q = fma_(q, s, 1.0 / 80.0);
    q = fma_(q, s, 1.0 / 32.0);
    q = fma_(q, s, 1.0 / 12.0);
    q = fma_(q, s, 0.25);
    const double l1 = fma_(s * s, q, s);
    const double de = (double)e;
    // -2*ln2 split: the high part has 32 significant bits so de*hi is exact
    return fma_(de, -2.0 * 6.93147180369123816490e-01, fma_(de, -2.0 * 1.90821492927058770002e-10, B + l1));
}
template <class Tab> BHIP_HD double det_log(double x, const Tab &tab) { return -0.5 * det_m2log(x, tab); }
BHIP_HD double det_log(double x) { return det_log(x, TabConst()); }

// sin/cos(2*pi*u) from the reduced argument: u = jr/32 + f with jr = round(32u) and x = 2*pi*f, |x| <= pi/32; Taylor sin/cos
// of x, rotated by the table row jr mod 32.
template <class Tab>
BHIP_HD void det_sincos_reduced(double x, uint32_t jr, const Tab &tab, double &sn, double &cs)
{
    double ck, sk;
    tab.sc(jr & 31u, ck, sk);
    const double z = x * x;
    double ps = 1.0 / 362880.0;
    ps = fma_(ps, z, -1.0 / 5040.0);
    ps = fma_(ps, z, 1.0 / 120.0);
    ps = fma_(ps, z, -1.0 / 6.0);
    const double sf = fma_(x * z, ps, x);
    double pc = -1.0 / 3628800.0;
    pc = fma_(pc, z, 1.0 / 40320.0);
    pc = fma_(pc, z, -1.0 / 720.0);
    pc = fma_(pc, z, 1.0 / 24.0);
    pc = fma_(pc, z, -0.5);
    const double cf = fma_(z, pc, 1.0);
    cs = fma_(-sk, sf, ck * cf);
    sn = fma_(ck, sf, sk * cf);
}
// u = K*2^-53 in [0,1) whose top 32 source bits are `w` (53-bit angles; kept for the tests of the function itself)
template <class Tab>
BHIP_HD void det_sincos2pi(double u, uint32_t w, const Tab &tab, double &sn, double &cs)
{
    const uint32_t jr = ((w >> 26) + 1u) >> 1;
    const double f = fma_((double)jr, -0.03125, u);
    det_sincos_reduced(f * 6.283185307179586, jr, tab, sn, cs);
}
// u = K24*2^-24: jr = round(32u) from the top six bits, f = u - jr/32 = (K24 - jr*2^19)*2^-24 exactly (an integer of at most
// 19 bits), x = fl(2 pi)*f -- the same x the 53-bit form computes for this u
template <class Tab>
BHIP_HD void det_sincos2pi_k24(uint32_t k24, const Tab &tab, double &sn, double &cs)
{
    const uint32_t jr = ((k24 >> 18) + 1u) >> 1;
    const int m = (int)k24 - (int)(jr << 19);
    det_sincos_reduced((double)m * (6.283185307179586 * 0x1.0p-24), jr, tab, sn, cs);
}

// one Box-Muller transform from 64 of a Philox call's bits (words a, b): normals z0 (cos branch), z1 (sin branch)
template <class Tab>
BHIP_HD void box_muller_40_24(const Tab &tab, uint32_t a, uint32_t b, double &z0, double &z1)
{
    const double rad = sqrt_fixed_range(det_m2log(u40_open0(a, b), tab));
    double s, c;
    det_sincos2pi_k24(b & 0xffffffu, tab, s, c);
    z0 = rad * c;
    z1 = rad * s;
}

// ---- the FULL-RESOLUTION specification "bhip-philox-v2" (selectable: BHIP_OPT_NOISE_SPEC = 2).  All 128 bits of a Philox call
// go into ONE Box-Muller pair -- 53 bits of radius (|z| <= 8.57), 53 bits of angle -- as the round-2 library drew them: call h
// gives pair h (normals 2h, 2h+1), u1 = (bits53(r0, r1) + 1) 2^-53, u2 = bits53(r2, r3) 2^-53.  Same log / sqrt / sin / cos.
// Twice the Philox calls per normal of v3; golden vectors guided_paths_v2 / _v3.npz are this stream.
template <class Tab>
BHIP_HD void box_muller_53_53(const Tab &tab, const u32x4 &r, double &z0, double &z1)
{
    const double u1 = u53_open0(r.x, r.y);
    const double u2 = u53_open1(r.z, r.w);
    const double rad = sqrt_fixed_range(det_m2log(u1, tab));
    double s, c;
    det_sincos2pi(u2, r.w, tab, s, c);
    z0 = rad * c;
    z1 = rad * s;
}
// The specification travels in the TYPE of the table accessor: FullRes<Tab> reads the tables like Tab and makes normal_quad /
// normal_pair draw the v2 stream.  A kernel holds both code paths and picks one per launch with a wave-uniform branch hoisted out
// of its loops (KArgs::noise_spec), so the default path's schedule is the one it had.  (A branch AT the draw instead breaks the step's
// single basic block -- the generator no longer interleaves with the recurrence: +40 % on the 4..8-dimensional kernels, measured.)
template <class Tab>
struct FullRes : Tab {
    static constexpr int NOISE_SPEC = 2;
    BHIP_HD explicit FullRes(const Tab &t) : Tab(t) {}
};
template <class...> using rng_void_t = void;
template <class Tab, class = void> struct noise_spec_of { static constexpr int value = 3; };
template <class Tab> struct noise_spec_of<Tab, rng_void_t<decltype(Tab::NOISE_SPEC)>> { static constexpr int value = Tab::NOISE_SPEC; };

template <class Tab, class = void> struct is_hot_only { static constexpr bool value = false; };
template <class Tab> struct is_hot_only<Tab, rng_void_t<decltype(Tab::HOT_ONLY)>> { static constexpr bool value = Tab::HOT_ONLY; };

// normals 4q .. 4q+3 of stream 0: v4 -- Philox call q, one normal per word;  v3 -- call q, two 40 + 24-bit pairs;
// v2 -- calls 2q and 2q+1, one 53 + 53-bit pair each
template <class Tab>
BHIP_HD void normal_quad(const Tab &tab, uint32_t k0, uint32_t k1, uint32_t path, uint32_t iter, uint32_t q, double &z0, double &z1, double &z2,
                         double &z3, uint32_t stream = 0u)
{
    if constexpr (noise_spec_of<Tab>::value == 4) {
        const u32x4 r = philox4x32_10(path, stream, iter, q, k0, k1);
        z0 = icdf_normal(tab, r.x); z1 = icdf_normal(tab, r.y); z2 = icdf_normal(tab, r.z); z3 = icdf_normal(tab, r.w);
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (is_hot_only<Tab>::value) {   // a table that holds the near octaves only (IcdfLDSHot): the rare far word redraws the call
            if (__builtin_amdgcn_ballot_w64(*tab.cold) != 0ull) {
                z0 = icdf_normal(IcdfConst(), r.x); z1 = icdf_normal(IcdfConst(), r.y); z2 = icdf_normal(IcdfConst(), r.z); z3 = icdf_normal(IcdfConst(), r.w);
                *tab.cold = false;
            }
        }
#endif
    } else if constexpr (noise_spec_of<Tab>::value == 2) {
        const u32x4 ra = philox4x32_10(path, stream, iter, 2u * q, k0, k1);
        const u32x4 rb = philox4x32_10(path, stream, iter, 2u * q + 1u, k0, k1);
        box_muller_53_53(tab, ra, z0, z1);
        box_muller_53_53(tab, rb, z2, z3);
    } else {
        const u32x4 r = philox4x32_10(path, stream, iter, q, k0, k1);
        box_muller_40_24(tab, r.x, r.y, z0, z1);
        box_muller_40_24(tab, r.z, r.w, z2, z3);
    }
}

// pair `h` of stream 0 -> normals 2h (z0) and 2h+1 (z1): half h & 1 of Philox call h >> 1.  The generic form (host, the
// kernels that draw few normals); the hot kernels draw whole calls with normal_quad.
template <class Tab>
BHIP_HD void normal_pair(const Tab &tab, uint32_t k0, uint32_t k1, uint32_t path, uint32_t iter, uint32_t h, double &z0, double &z1,
                         uint32_t stream = 0u)
{
    if constexpr (noise_spec_of<Tab>::value == 4) {
        // (advisor r5) the hot-rows table answers only through normal_quad, which checks `cold` and redraws from the full table; here a
        // far-octave word would silently take a value from the wrong row
        static_assert(!is_hot_only<Tab>::value, "IcdfLDSHot holds the near octaves only: draw with normal_quad (it handles the cold words)");
        const u32x4 r = philox4x32_10(path, stream, iter, h >> 1, k0, k1);
        const bool second = (h & 1u) != 0u;
        z0 = icdf_normal(tab, second ? r.z : r.x);
        z1 = icdf_normal(tab, second ? r.w : r.y);
    } else if constexpr (noise_spec_of<Tab>::value == 2) {
        box_muller_53_53(tab, philox4x32_10(path, stream, iter, h, k0, k1), z0, z1);
    } else {
        const u32x4 r = philox4x32_10(path, stream, iter, h >> 1, k0, k1);
        const bool second = (h & 1u) != 0u;
        box_muller_40_24(tab, second ? r.z : r.x, second ? r.w : r.y, z0, z1);
    }
}
// the same with the specification chosen at run time (2: full resolution; 3: v3; anything else: v4, the default) -- host code and
// the kernels that draw a handful of normals per launch; the tables from constant memory / the host arrays
BHIP_HD void normal_pair_spec(int spec, uint32_t k0, uint32_t k1, uint32_t path, uint32_t iter, uint32_t h, double &z0, double &z1,
                              uint32_t stream = 0u)
{
    if (spec == 2) normal_pair(FullRes<TabConst>(TabConst()), k0, k1, path, iter, h, z0, z1, stream);
    else if (spec == 3) normal_pair(TabConst(), k0, k1, path, iter, h, z0, z1, stream);
    else normal_pair(IcdfConst(), k0, k1, path, iter, h, z0, z1, stream);
}

BHIP_HD double accept_uniform(uint32_t k0, uint32_t k1, uint32_t path, uint32_t iter)
{
    const u32x4 r = philox4x32_10(path, 1u, iter, 0u, k0, k1);
    return u53_open0(r.x, r.y);
}

}  // namespace bhip
