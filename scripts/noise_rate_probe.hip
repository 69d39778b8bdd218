// Issue cost of one standard normal under the noise specifications of bhip_rng.h, outside any path kernel: every lane draws Philox
// calls back to back (four normals each) and sums them.  Waves per SIMD as a parameter (1, 2, 4): the path kernels run the generator at
// 1-2 producer waves per SIMD (k_pc) or inside 4 waves per SIMD (k_paths).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I bridge.jl_amd/csrc scripts/noise_rate_probe.hip -o /tmp/nrp && /tmp/nrp
//   (-DBHIP_NO_BITOP3: the Philox xors as the compiler pairs them, for the A/B of v_bitop3_b32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bhip_rng.h"

template <int SPEC>
__global__ __launch_bounds__(256) void kq(double *out, uint32_t k0, uint32_t k1, int n)
{
    __shared__ __attribute__((aligned(16))) double tab[bhip::RNG_LDS_DOUBLES];
    if (SPEC == 4) bhip::IcdfLDS::load(tab, threadIdx.x, blockDim.x);
    else bhip::TabLDS::load(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    double acc = 0.0;
    const uint32_t path = blockIdx.x * blockDim.x + threadIdx.x;
    auto run = [&](const auto &t) {
        for (int q = 0; q < n; q++) {
            double z0, z1, z2, z3;
            bhip::normal_quad(t, k0, k1, path, 5u, (uint32_t)q, z0, z1, z2, z3);
            acc += z0; acc += z1; acc += z2; acc += z3;
        }
    };
    if constexpr (SPEC == 4) run(bhip::IcdfLDS(tab));
    else if constexpr (SPEC == 3) run(bhip::TabLDS(tab));
    else run(bhip::FullRes<bhip::TabLDS>(bhip::TabLDS(tab)));
    out[path] = acc;
}

template <int SPEC>
static void bench(double *d, int waves_per_simd, int n)
{
    const int blocks = 256 * waves_per_simd;   // 256 CUs x 4 SIMDs x waves / 4 waves per block
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kq<SPEC>, dim3(blocks), dim3(256), 0, 0, d, 1u, 2u, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double normals = 4.0 * n * blocks * 256.0;
    // cycles per normal and wave at a nominal 2.1 GHz: a SIMD with w waves issues for all of them
    const double cyc = ms * 1e-3 * 2.1e9 / (4.0 * n) / waves_per_simd;
    printf("spec v%d  %d wave(s)/SIMD  %8.3f ms  %7.1f Gnormals/s  %6.1f SIMD-cycles per wave-normal (2.1 GHz nominal)\n", SPEC, waves_per_simd, ms, normals / ms * 1e-6, cyc);
}

int main()
{
    double *d;
    if (hipMalloc(&d, 8 * 256 * 256 * 8) != hipSuccess) return 2;
    for (int w : {1, 2, 4, 8}) {
        bench<4>(d, w, 20000);
        bench<3>(d, w, 20000);
        bench<2>(d, w, 10000);
    }
    return 0;
}
