// The device forms in bridge.jl_amd/csrc/bhip_rng.h (u53_bits, u40_open0, sqrt_fixed_range, det_sincos2pi_k24) against the portable expressions
// they replace -- integer->double conversion, IEEE square root -- on 2^32 pseudo-random inputs from the ranges the
// generator uses, plus the range edges; and the table-driven -2*log / sincos read from LDS (TabLDS, the producer
// waves' and the tile kernel's path) against the constant-memory reads (TabConst).  Any differing bit fails.
// Specification v4 (the inverse distribution function of one 32-bit word): k_icdf below, exhaustive.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I bridge.jl_amd/csrc tests/rng_device_forms.hip -o /tmp/rdf && /tmp/rdf
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bhip_rng.h"

__device__ __forceinline__ bool same(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

__global__ void k(unsigned long long *bad, int rounds)
{
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ __attribute__((aligned(16))) double tab[bhip::RNG_TAB_DOUBLES];
    bhip::TabLDS::load(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const bhip::TabLDS lds(tab);
    unsigned long long nb = 0;
    for (int r = 0; r < rounds; r++) {
        const bhip::u32x4 v = bhip::philox4x32_10(t, 7u, (unsigned)r, 0u, 0x1234u, 0x5678u);
        // uniforms
        const unsigned long long a = ((unsigned long long)v.y << 32) | v.x;
        const double p0 = (double)((a >> 11) + 1) * 0x1.0p-53, p1 = (double)(a >> 11) * 0x1.0p-53;
        nb += !same(p0, bhip::u53_open0(v.x, v.y)) + !same(p1, bhip::u53_open1(v.x, v.y));
        // specification v3: the 40-bit radius uniform (mantissa-field construction vs integer -> double conversion), the 24-bit
        // angle (integer reduction vs the 53-bit routine on the same u), a whole call == its two pairs, LDS tables == constant tables
        {
            const unsigned long long k40 = ((unsigned long long)(v.y >> 24) << 32) | v.x;
            nb += !same((double)(k40 + 1) * 0x1.0p-40, bhip::u40_open0(v.x, v.y));
            const unsigned k24 = v.w & 0xffffffu;
            double s0, c0, s1, c1;
            bhip::det_sincos2pi_k24(k24, lds, s0, c0);
            bhip::det_sincos2pi((double)k24 * 0x1.0p-24, k24 << 8, bhip::TabConst(), s1, c1);
            nb += !same(s0, s1) + !same(c0, c1);
            double z0, z1, z2, z3, y0, y1, y2, y3;
            bhip::normal_quad(lds, 0x1234u, 0x5678u, t, (unsigned)r, 3u, z0, z1, z2, z3);
            bhip::normal_pair(bhip::TabConst(), 0x1234u, 0x5678u, t, (unsigned)r, 6u, y0, y1);
            bhip::normal_pair(bhip::TabConst(), 0x1234u, 0x5678u, t, (unsigned)r, 7u, y2, y3);
            nb += !same(z0, y0) + !same(z1, y1) + !same(z2, y2) + !same(z3, y3);
            nb += !(bhip::det_m2log(p0, lds) >= 0.0);
        }
        // square root: -2 log(u) with u spread over (0,1] including values next to 0 and next to 1
        const double u = bhip::u53_open0(v.z, v.w);
        const double xs[3] = {bhip::det_m2log(u, lds), bhip::det_m2log(u * 0x1.0p-40 + 0x1.0p-53, lds), (double)(v.z >> 1) * 0x1.0p-21 * 0.75};
        for (int j = 0; j < 3; j++) nb += !same(__builtin_sqrt(xs[j]), bhip::sqrt_fixed_range(xs[j]));
    }
    if (t == 0) {   // range edges
        const double e[6] = {0.0, 0x1.0p-53 * 2.0, 0x1.0p-52, 1.0, 1500.0, 73.47};
        for (int j = 0; j < 6; j++) nb += !same(__builtin_sqrt(e[j]), bhip::sqrt_fixed_range(e[j]));
        nb += !same(bhip::u53_open0(0xFFFFFFFFu, 0xFFFFFFFFu), 1.0) + !same(bhip::u53_open1(0u, 0u), 0.0) + !same(bhip::u53_open0(0u, 0u), 0x1.0p-53);
        nb += !same(bhip::u53_open1(0xFFFFFFFFu, 0xFFFFFFFFu), 1.0 - 0x1.0p-53);
        nb += !same(bhip::det_m2log(1.0, lds), 0.0) + !(bhip::det_m2log(1.0 - 0x1.0p-53, lds) > 0.0);
        nb += !same(bhip::u40_open0(0xFFFFFFFFu, 0xFFFFFFFFu), 1.0) + !same(bhip::u40_open0(0u, 0x00FFFFFFu), 0x1.0p-40) + !same(bhip::u40_open0(0u, 0x01000000u), 0x1.0p-8 + 0x1.0p-40);
    }
    if (nb) atomicAdd(bad, nb);
}

// Specification v4 (icdf_normal): ALL 2^32 words.  On the device the table read from LDS (IcdfLDS, three planes) against the constant-
// memory read (IcdfConst), symmetry z(w ^ 0x80000000) = -z(w), monotonicity inside the thread's run of consecutive words (|z| never
// increases with the word by more than the polynomial's error at a segment boundary), range; and a 64-bit checksum of the result bits
// per block of 2^16 words, which the HOST recomputes with its own icdf_normal for 514 of the 65 536 blocks (first, last, 512 spread).
__global__ void k_icdf(unsigned long long *bad, unsigned long long *sums)
{
    __shared__ __attribute__((aligned(16))) double tab[bhip::ICDF_TAB_DOUBLES];
    __shared__ unsigned long long blocksum;
    bhip::IcdfLDS::load(tab, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) blocksum = 0ull;
    __syncthreads();
    const bhip::IcdfLDS lds(tab);
    const bhip::IcdfConst cst;
    unsigned long long nb = 0, sum = 0;
    const unsigned w0 = blockIdx.x * 65536u + threadIdx.x * 256u;   // 256 threads x 256 consecutive words
    double prev = 0.0;
    for (unsigned j = 0; j < 256u; j++) {
        const unsigned w = w0 + j;
        const double a = bhip::icdf_normal(lds, w), b = bhip::icdf_normal(cst, w);
        nb += !same(a, b);
        sum += (unsigned long long)__double_as_longlong(a);
        const double m = bhip::icdf_normal(lds, w ^ 0x80000000u);
        nb += !same(m, -a);
        const double aa = fabs(a);
        nb += !(aa > 0.0 && aa < 6.3380);
        nb += !(((w >> 31) != 0u) == (a < 0.0));
        if (j > 0) nb += !(aa <= prev + 8e-9);   // the upper-tail probability grows with the word: |z| falls (up to twice the fit error at a row change)
        prev = aa;
    }
    atomicAdd(&blocksum, sum);
    if (nb) atomicAdd(bad, nb);
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = blocksum;
}

static int check_icdf()
{
    unsigned long long *d, *ds, h = 0;
    if (hipMalloc(&d, 8) != hipSuccess || hipMalloc(&ds, 8 * 65536) != hipSuccess) { printf("FAIL alloc\n"); return 2; }
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k_icdf, dim3(65536), dim3(256), 0, 0, d, ds);
    unsigned long long *hs = new unsigned long long[65536];
    if (hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hs, ds, 8 * 65536, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL copy\n"); return 2; }
    if (h) { printf("FAIL icdf: %llu mismatching results on the device\n", h); return 1; }
    int nbad = 0;
    for (int q = 0; q < 514; q++) {
        const unsigned blk = q == 0 ? 0u : q == 1 ? 65535u : (unsigned)(((unsigned long long)(q - 2) * 2654435761ull) % 65536ull);
        unsigned long long sum = 0;
        for (unsigned j = 0; j < 65536u; j++) {
            union { double d; unsigned long long u; } b;
            b.d = bhip::icdf_normal(bhip::IcdfConst(), blk * 65536u + j);
            sum += b.u;
        }
        nbad += sum != hs[blk];
    }
    if (nbad) { printf("FAIL icdf: host and device disagree on %d of 514 blocks of 2^16 words\n", nbad); return 1; }
    // the ends of the range on the host
    const double zmax = bhip::icdf_normal(bhip::IcdfConst(), 0u), zmin = bhip::icdf_normal(bhip::IcdfConst(), 0x7fffffffu);
    if (!(zmax > 6.33 && zmax < 6.34 && zmin > 0.0 && zmin < 1e-9)) { printf("FAIL icdf range %g %g\n", zmax, zmin); return 1; }
    delete[] hs;
    return 0;
}

int main()
{
    if (int rc = check_icdf()) return rc;
    unsigned long long *d, h = 0;
    if (hipMalloc(&d, 8) != hipSuccess) { printf("FAIL alloc\n"); return 2; }
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k, dim3(16384), dim3(256), 0, 0, d, 1024);   // 2^32 samples
    if (hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL copy\n"); return 2; }
    if (h) printf("FAIL %llu mismatching results\n", h);
    else printf("OK device forms == portable expressions on 2^32 samples (uniforms, sqrt, LDS tables == constant tables); "
                "v4: all 2^32 words, LDS == constant table, host == device on 514 blocks of 2^16\n");
    return h ? 1 : 0;
}
