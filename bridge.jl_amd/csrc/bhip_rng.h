// bhip_rng.h -- RNG specification "bhip-philox-v1" (host + device).
//
// Replaces the reference's global randn() (src/wiener.jl:31,44,55), which is not reproducible
// outside Julia, with a counter-based generator whose output depends only on
// (seed, global path id, iteration, normal index): results are independent of GPU count and
// launch geometry, and a chain can be resumed from its counters alone.
//
//   Philox4x32-10 (the generator behind rocRAND's default PHILOX4_32_10; Random123 KAT vectors in
//   tests/), key = (seed_lo, seed_hi), counter = (path, stream, iter, block).
//   stream 0: block j -> standard normals 2j (cos branch) and 2j+1 (sin branch) by Box-Muller,
//             u1 = (bits53(r0,r1)+1)*2^-53 in (0,1],  u2 = bits53(r2,r3)*2^-53 in [0,1)
//   stream 1: block 0 -> the Metropolis-Hastings uniform U = (bits53(r0,r1)+1)*2^-53
//
// log and sin/cos(2*pi*u) are built from +,-,*,/ and fma only, so that every host and device
// evaluates bit-identical normals (no libm / ocml dependence).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BHIP_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define BHIP_HD static inline
#endif

namespace bhip {

struct u32x4 { uint32_t x, y, z, w; };

BHIP_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32x32->64 product per multiplier (v_mad_u64_u32 on gfx950) instead of mul_hi + mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

BHIP_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

#if defined(__HIP_DEVICE_COMPILE__)
// Device forms of the three fixed-range operations below.  They produce the SAME bits as the portable
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
// a / b and sqrt(x) as the compiler expands them (reciprocal / reciprocal-square-root seed, Newton steps on
// fma, final residual correction = correctly rounded), WITHOUT the exponent pre-scaling and special-value
// fix-ups that only matter outside the ranges used here: b in [1.7, 2.5], |a| < 1;  x in [0, 1500].
BHIP_HD double div_fixed_range(double a, double b)
{
    const double y0 = __builtin_amdgcn_rcp(b);
    const double e0 = __builtin_fma(-b, y0, 1.0);
    const double y1 = __builtin_fma(y0, e0, y0);
    const double e1 = __builtin_fma(-b, y1, 1.0);
    const double y2 = __builtin_fma(y1, e1, y1);
    const double q0 = a * y2;
    const double r = __builtin_fma(-b, q0, a);
    return __builtin_fma(r, y2, q0);
}
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
BHIP_HD double div_fixed_range(double a, double b) { return a / b; }
BHIP_HD double sqrt_fixed_range(double x) { return __builtin_sqrt(x); }
#endif

// natural log for x in (0,1], normal doubles:  x = 2^e m, m in [sqrt(1/2), sqrt(2));
// log m = 2 atanh(s), s = (m-1)/(m+1), odd Taylor series in s up to s^23 (|s| <= 0.1716).
BHIP_HD double det_log(double x)
{
    union { double d; uint64_t u; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7ff) - 1023;
    v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = div_fixed_range(m - 1.0, m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = fma_(p, z, 1.0 / 21.0);
    p = fma_(p, z, 1.0 / 19.0);
    p = fma_(p, z, 1.0 / 17.0);
    p = fma_(p, z, 1.0 / 15.0);
    p = fma_(p, z, 1.0 / 13.0);
    p = fma_(p, z, 1.0 / 11.0);
    p = fma_(p, z, 1.0 / 9.0);
    p = fma_(p, z, 1.0 / 7.0);
    p = fma_(p, z, 1.0 / 5.0);
    p = fma_(p, z, 1.0 / 3.0);
    const double t = s * z;
    double lm = fma_(t, p, s);
    lm = lm + lm;
    const double de = (double)e;
    return fma_(de, 6.93147180369123816490e-01, fma_(de, 1.90821492927058770002e-10, lm));
}

// sin/cos(2*pi*u), u in [0,1): q = round(4u), f = u - q/4 (exact), theta = 2*pi*f, Taylor to
// theta^15 / theta^16, quadrant rotation.
BHIP_HD void det_sincos2pi(double u, double &sn, double &cs)
{
    const double q = __builtin_floor(fma_(u, 4.0, 0.5));
    const double f = fma_(q, -0.25, u);
    const double th = f * 6.283185307179586;
    const double z = th * th;
    double ps = -1.0 / 1307674368000.0;
    ps = fma_(ps, z, 1.0 / 6227020800.0);
    ps = fma_(ps, z, -1.0 / 39916800.0);
    ps = fma_(ps, z, 1.0 / 362880.0);
    ps = fma_(ps, z, -1.0 / 5040.0);
    ps = fma_(ps, z, 1.0 / 120.0);
    ps = fma_(ps, z, -1.0 / 6.0);
    const double s0 = fma_(th * z, ps, th);
    double pc = 1.0 / 20922789888000.0;
    pc = fma_(pc, z, -1.0 / 87178291200.0);
    pc = fma_(pc, z, 1.0 / 479001600.0);
    pc = fma_(pc, z, -1.0 / 3628800.0);
    pc = fma_(pc, z, 1.0 / 40320.0);
    pc = fma_(pc, z, -1.0 / 720.0);
    pc = fma_(pc, z, 1.0 / 24.0);
    pc = fma_(pc, z, -0.5);
    const double c0 = fma_(z, pc, 1.0);
    // quadrant rotation without branches: q odd swaps sin/cos; sin is negated for q in {2,3}, cos for q in {1,2}
    const int qi = (int)q & 3;
    const bool swp = (qi & 1) != 0;
    const double sb = swp ? c0 : s0, cb = swp ? s0 : c0;
    sn = (qi & 2) ? -sb : sb;
    cs = ((qi + 1) & 2) ? -cb : cb;
}

// block `blk` of stream 0 -> normals 2*blk (z0) and 2*blk+1 (z1)
BHIP_HD void normal_pair(uint32_t k0, uint32_t k1, uint32_t path, uint32_t iter, uint32_t blk, double &z0, double &z1)
{
    const u32x4 r = philox4x32_10(path, 0u, iter, blk, k0, k1);
    const double u1 = u53_open0(r.x, r.y);
    const double u2 = u53_open1(r.z, r.w);
    const double rad = sqrt_fixed_range(-2.0 * det_log(u1));
    double s, c;
    det_sincos2pi(u2, s, c);
    z0 = rad * c;
    z1 = rad * s;
}

BHIP_HD double accept_uniform(uint32_t k0, uint32_t k1, uint32_t path, uint32_t iter)
{
    const u32x4 r = philox4x32_10(path, 1u, iter, 0u, k0, k1);
    return u53_open0(r.x, r.y);
}

}  // namespace bhip
