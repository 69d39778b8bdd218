// Micro-benchmark 4: chain state as full 128-byte lines per (parity half, 16-step chunk, chain), moved by the
// wave cooperatively (8 lanes per line) and transposed through LDS so that each chain's lane gets its own 16
// values.  Only the CURRENT half is read and only the OTHER half is written: 8 + 8 B/path-step for W instead of
// the slot layout's 16 + 16, i.e. the algorithmic 32 B/path-step with X.  Costs: 17 KB of LDS per wave (double
// buffered) => 2 waves per SIMD, 32 staging VGPRs, 2 LDS accesses per step.  `spin` emulates the step's VALU work.
//   hipcc --offload-arch=gfx950 -O3 scripts/layout_probe4.hip -o /tmp/lp4 && /tmp/lp4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));

constexpr int CH = 16;          // steps per chunk = doubles per 128-byte line
constexpr int LDW = CH + 1;     // padded row of the per-wave LDS tile [64 chains][17]

template <int SPIN>
__global__ __launch_bounds__(256, 2) void k_lines(double *Wh, double *X, const unsigned char *cur, long P, int nch)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *buf0 = lds + (size_t)wave * 2 * 64 * LDW, *buf1 = buf0 + 64 * LDW;
    const long c0 = ((long)blockIdx.x * 4 + wave) * 64;
    const long p = c0 + lane;
    // cooperative mapping: instruction q moves lines of chains 8q + lane/8, lane%8 = 16-byte part of the line
    int par[8];
#pragma unroll
    for (int q = 0; q < 8; q++) par[q] = cur[c0 + 8 * q + (lane >> 3)];
    const size_t half_stride = (size_t)nch * P * CH;   // doubles between the two parity halves
    auto line = [&](int h, int chunk, long chain) { return Wh + (size_t)h * half_stride + ((size_t)chunk * P + chain) * CH; };
    d2v stage[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
        stage[q] = __builtin_nontemporal_load((const d2v *)(line(par[q], 0, c0 + 8 * q + (lane >> 3)) + 2 * (lane & 7)));
    double *x = X + p;
    double acc = 0.0;
    for (int k = 0; k < nch; k++) {
        double *buf = (k & 1) ? buf1 : buf0;
        // staged chunk k -> LDS tile
#pragma unroll
        for (int q = 0; q < 8; q++) {
            double *d = buf + (8 * q + (lane >> 3)) * LDW + 2 * (lane & 7);
            d[0] = stage[q].x; d[1] = stage[q].y;
        }
        __builtin_amdgcn_wave_barrier();
        // prefetch chunk k+1 (held in registers while this chunk is computed)
        const int kn = k + 1 < nch ? k + 1 : k;
#pragma unroll
        for (int q = 0; q < 8; q++)
            stage[q] = __builtin_nontemporal_load((const d2v *)(line(par[q], kn, c0 + 8 * q + (lane >> 3)) + 2 * (lane & 7)));
        // 16 steps: read own value, "compute", write the proposal value in place, store X
#pragma unroll 4
        for (int s = 0; s < CH; s++) {
            double v = buf[lane * LDW + s];
            for (int r = 0; r < SPIN; r++) v = __builtin_fma(v, 0.999, 0.001);
            acc += v;
            buf[lane * LDW + s] = v;
            const long i = (long)k * CH + s;
            __builtin_nontemporal_store(v, &x[(i * 2 + 0) * P]);
            __builtin_nontemporal_store(acc, &x[(i * 2 + 1) * P]);
        }
        __builtin_amdgcn_wave_barrier();
        // LDS tile -> the OTHER half's lines
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const double *d = buf + (8 * q + (lane >> 3)) * LDW + 2 * (lane & 7);
            __builtin_nontemporal_store(d2v{d[0], d[1]}, (d2v *)(line(par[q] ^ 1, k, c0 + 8 * q + (lane >> 3)) + 2 * (lane & 7)));
        }
    }
}

template <int SPIN>
float run(double *Wh, double *X, unsigned char *cur, long P, int nch)
{
    const size_t lds = sizeof(double) * 4 * 2 * 64 * LDW;
    (void)hipFuncSetAttribute((const void *)k_lines<SPIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_lines<SPIN>, dim3(P / 256), dim3(256), lds, 0, Wh, X, cur, P, nch);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_lines<SPIN>, dim3(P / 256), dim3(256), lds, 0, Wh, X, cur, P, nch);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms / 3;
}

int main()
{
    const long P = 262144;
    const int nch = 62;   // 992 steps
    double *Wh, *X; unsigned char *cur;
    (void)hipMalloc(&Wh, sizeof(double) * 2 * nch * P * CH);
    (void)hipMalloc(&X, sizeof(double) * (size_t)nch * CH * 2 * P);
    (void)hipMalloc(&cur, P);
    (void)hipMemset(Wh, 0, sizeof(double) * 2 * nch * P * CH);
    unsigned char *h = new unsigned char[P];
    for (long p = 0; p < P; p++) h[p] = (unsigned char)((p * 2654435761u >> 7) & 1);
    (void)hipMemcpy(cur, h, P, hipMemcpyHostToDevice);
    const double gb = 32.0 * P * nch * CH / 1e9;
    const float t0 = run<0>(Wh, X, cur, P, nch), t1 = run<60>(Wh, X, cur, P, nch), t2 = run<140>(Wh, X, cur, P, nch);
    printf("full-line layout + LDS transpose, 2 waves/SIMD, 992 steps:  no VALU work %.3f ms (%.0f GB/s of 32 B/path-step)\n", t0, gb / t0 * 1e3);
    printf("   with 60 dependent fp64 FMAs per step %.3f ms;  with 140 per step %.3f ms      [slot layout, real kernel: 2.2-2.5 ms per 1000 steps]\n", t1, t2);
    return 0;
}
