// Micro-benchmark (gfx950): do fp64 MFMA and VALU work overlap on one SIMD?
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
// mode 0: MFMA only   1: fp64 VALU FMA only   2: int VALU (mad_u64_u32) only
// mode 3: even waves MFMA, odd waves fp64 VALU   4: even waves MFMA, odd waves int VALU
// mode 5: every wave interleaves MFMA + fp64 VALU   6: every wave interleaves MFMA + int VALU
// Grid: 256 CUs x 4 SIMDs x `wps` waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters, double seed)
{
    const int lane = threadIdx.x;
    const int wave = blockIdx.x;
    const bool do_m = MODE == 0 || MODE >= 5 || ((MODE == 3 || MODE == 4) && (wave & 1) == 0);
    const bool do_f = MODE == 1 || MODE == 5 || (MODE == 3 && (wave & 1) == 1);
    const bool do_i = MODE == 2 || MODE == 6 || (MODE == 4 && (wave & 1) == 1);
    double4v a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    double x0 = seed + lane, x1 = seed * 2 + lane, x2 = seed * 3, x3 = seed * 4, x4 = 1.0, x5 = 2.0, x6 = 3.0, x7 = 4.0;
    uint32_t c0 = lane, c1 = lane * 3 + 1, c2 = lane * 5 + 2, c3 = lane * 7 + 3;
    const double av = seed * 1e-3 + lane * 1e-6, bv = seed * 1e-4;
    for (int it = 0; it < iters; it++) {
        if (do_m) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, a1, 0, 0, 0);
            }
        }
        if (do_f) {
#pragma unroll
            for (int u = 0; u < 32; u++) {   // 8 independent chains x 32 = 256 v_fma_f64
                x0 = __builtin_fma(x0, 0.999999, 1e-9); x1 = __builtin_fma(x1, 0.999998, 1e-9);
                x2 = __builtin_fma(x2, 0.999997, 1e-9); x3 = __builtin_fma(x3, 0.999996, 1e-9);
                x4 = __builtin_fma(x4, 0.999995, 1e-9); x5 = __builtin_fma(x5, 0.999994, 1e-9);
                x6 = __builtin_fma(x6, 0.999993, 1e-9); x7 = __builtin_fma(x7, 0.999992, 1e-9);
            }
        }
        if (do_i) {
#pragma unroll
            for (int u = 0; u < 64; u++) {   // 128 v_mad_u64_u32 + xors
                const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
                const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ u, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ it;
                c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
            }
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0 + c1 + c2 + c3;
}

template <int MODE>
float run(double *out, int waves, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, out, 10, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, out, iters, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double *out;
    hipMalloc(&out, sizeof(double) * 64 * 1024 * 8);
    const int iters = 4000;
    for (int wps : {1, 2, 4}) {
        const int waves = 256 * 4 * wps;
        const float t[7] = {run<0>(out, waves, iters), run<1>(out, waves, iters), run<2>(out, waves, iters), run<3>(out, waves, iters),
                            run<4>(out, waves, iters), run<5>(out, waves, iters), run<6>(out, waves, iters)};
        const double mf = (double)waves * iters * 16 * 2048.0, vf = (double)waves * iters * 256 * 64 * 2.0;
        printf("waves/SIMD %d: MFMA %.2f ms (%.1f TF)  f64VALU %.2f ms (%.1f TF)  intVALU %.2f ms | split MFMA|f64 %.2f  split MFMA|int %.2f | "
               "fused MFMA+f64 %.2f (sum %.2f)  fused MFMA+int %.2f (sum %.2f)\n",
               wps, t[0], mf / t[0] / 1e9, t[1], vf / t[1] / 1e9, t[2], t[3], t[4], t[5], t[0] + t[1], t[6], t[0] + t[2]);
    }
    return 0;
}
