// Micro-benchmark 2: can the MCMC kernel's W traffic drop from 32 to 16 B/path-step with a TIME-CHUNKED
// per-chain layout?  Wt[p][chunk][half][8 doubles]: a lane reads the 64-byte chunk of its current half
// and writes the 64-byte chunk of the other half (no redundant bytes, but every lane touches its own
// line); X stays plain SoA (16 B/path-step).  Compared with the slot layout of layout_probe.hip
// (48 B/path-step, fully coalesced).
//   hipcc --offload-arch=gfx950 -O3 scripts/layout_probe2.hip -o /tmp/lp2 && /tmp/lp2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));

template <int CH>   // CH doubles per chunk (8 -> 64 B, 16 -> 128 B)
__global__ __launch_bounds__(256) void k_chunk(double *Wt, double *X, const unsigned char *cur, long P, int N, int spin)
{
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = cur[p];
    const int nch = N / CH;
    double *w = Wt + (size_t)p * nch * 2 * CH;
    double *x = X + p;
    d2v nb[CH / 2], cb[CH / 2];
#pragma unroll
    for (int q = 0; q < CH / 2; q++) nb[q] = __builtin_nontemporal_load((const d2v *)(w + (size_t)(0 * 2 + c) * CH) + q);
    double acc = 0;
    for (int ch = 0; ch < nch; ch++) {
#pragma unroll
        for (int q = 0; q < CH / 2; q++) cb[q] = nb[q];
        const int nx = ch + 1 < nch ? ch + 1 : ch;
#pragma unroll
        for (int q = 0; q < CH / 2; q++) nb[q] = __builtin_nontemporal_load((const d2v *)(w + (size_t)(nx * 2 + c) * CH) + q);
        d2v ob[CH / 2];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            double v = (j & 1) ? cb[j >> 1].y : cb[j >> 1].x;
            for (int s = 0; s < spin; s++) v = __builtin_fma(v, 0.999, 0.001);
            acc += v;
            if (j & 1) ob[j >> 1].y = v; else ob[j >> 1].x = v;
            const long i = (long)ch * CH + j;
            __builtin_nontemporal_store(v, &x[(i * 2 + 0) * P]);
            __builtin_nontemporal_store(acc, &x[(i * 2 + 1) * P]);
        }
#pragma unroll
        for (int q = 0; q < CH / 2; q++) __builtin_nontemporal_store(ob[q], (d2v *)(w + (size_t)(ch * 2 + (c ^ 1)) * CH) + q);
    }
}

template <int CH>
float run(double *Wt, double *X, unsigned char *cur, long P, int N, int spin)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_chunk<CH>, dim3(P / 256), dim3(256), 0, 0, Wt, X, cur, P, N, spin);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_chunk<CH>, dim3(P / 256), dim3(256), 0, 0, Wt, X, cur, P, N, spin);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main()
{
    const long P = 262144;
    const int N = 992;   // multiple of 8 and 16
    double *Wt, *X; unsigned char *cur;
    (void)hipMalloc(&Wt, sizeof(double) * P * N * 2);
    (void)hipMalloc(&X, sizeof(double) * P * N * 2);
    (void)hipMalloc(&cur, P);
    (void)hipMemset(Wt, 0, sizeof(double) * P * N * 2);
    unsigned char *h = new unsigned char[P];
    for (long p = 0; p < P; p++) h[p] = (unsigned char)((p * 2654435761u >> 7) & 1);
    (void)hipMemcpy(cur, h, P, hipMemcpyHostToDevice);
    const double gb = 32.0 * P * N / 1e9;
    for (int spin : {0, 40}) {
        const float a = run<8>(Wt, X, cur, P, N, spin), b = run<16>(Wt, X, cur, P, N, spin);
        printf("spin %3d: time-chunked 64 B %.3f ms (%.0f GB/s of 32 B/path-step) | 128 B chunks %.3f ms (%.0f GB/s)   [slot layout reference: ~2.18 ms per 1000 steps]\n",
               spin, a, gb / a * 1e3, b, gb / b * 1e3);
    }
    return 0;
}
