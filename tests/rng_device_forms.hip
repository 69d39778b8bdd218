// The device forms in bridge.jl_amd/csrc/bhip_rng.h (u53_bits, sqrt_fixed_range) against the portable expressions
// they replace -- integer->double conversion, IEEE square root -- on 2^32 pseudo-random inputs from the ranges the
// generator uses, plus the range edges; and the table-driven -2*log / sincos read from LDS (TabLDS, the producer
// waves' and the tile kernel's path) against the constant-memory reads (TabConst).  Any differing bit fails.
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
        // the two table homes give the same normals
        {
            double z0, z1, y0, y1;
            bhip::normal_pair(bhip::TabConst(), 0x1234u, 0x5678u, t, (unsigned)r, 3u, z0, z1);
            bhip::normal_pair(lds, 0x1234u, 0x5678u, t, (unsigned)r, 3u, y0, y1);
            nb += !same(z0, y0) + !same(z1, y1);
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
    }
    if (nb) atomicAdd(bad, nb);
}

int main()
{
    unsigned long long *d, h = 0;
    if (hipMalloc(&d, 8) != hipSuccess) { printf("FAIL alloc\n"); return 2; }
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k, dim3(16384), dim3(256), 0, 0, d, 1024);   // 2^32 samples
    if (hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL copy\n"); return 2; }
    if (h) printf("FAIL %llu mismatching results\n", h);
    else printf("OK device forms == portable expressions on 2^32 samples (uniforms, sqrt, LDS tables == constant tables)\n");
    return h ? 1 : 0;
}
