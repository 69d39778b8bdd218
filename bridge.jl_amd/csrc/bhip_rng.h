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
    const double s = (m - 1.0) / (m + 1.0);
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
    const double rad = __builtin_sqrt(-2.0 * det_log(u1));
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
