// Micro-benchmark: the MCMC kernel's memory pattern without its arithmetic.
// Each lane walks N steps; per step it loads its 16-byte slot, stores it back and stores two 8-byte
// X components (48 B/path-step of traffic), all non-temporal.  layout 0: plain SoA (row stride = P);
// layout 1: path-blocked SoA (each 256-path block owns a contiguous region, row stride = 256).
//   hipcc --offload-arch=gfx950 -O3 scripts/layout_probe.hip -o /tmp/lp && /tmp/lp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));

template <int LAYOUT, int PF>
__global__ __launch_bounds__(256) void k(d2v *W, double *X, long P, int N, int spin)
{
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    d2v *w; double *x; long ld;
    if (LAYOUT == 0) { w = W + p; x = X + p; ld = P; }
    else { w = W + (long)blockIdx.x * N * 256 + threadIdx.x; x = X + (long)blockIdx.x * N * 2 * 256 + threadIdx.x; ld = 256; }
    d2v pf[PF];
#pragma unroll
    for (int j = 0; j < PF; j++) pf[j] = __builtin_nontemporal_load(&w[(long)j * ld]);
    double acc = 0;
    for (int i = 0; i < N; i++) {
        d2v cur = pf[0];
#pragma unroll
        for (int j = 0; j + 1 < PF; j++) pf[j] = pf[j + 1];
        const int ii = i + PF < N ? i + PF : N - 1;
        pf[PF - 1] = __builtin_nontemporal_load(&w[(long)ii * ld]);
        double v = cur.x + cur.y;
        for (int s = 0; s < spin; s++) v = __builtin_fma(v, 0.999, 0.001);   // stand-in for the arithmetic
        acc += v;
        __builtin_nontemporal_store(d2v{v, cur.y}, &w[(long)i * ld]);
        __builtin_nontemporal_store(v, &x[((long)i * 2 + 0) * ld]);
        __builtin_nontemporal_store(acc, &x[((long)i * 2 + 1) * ld]);
    }
}

template <int LAYOUT, int PF>
float run(d2v *W, double *X, long P, int N, int spin)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<LAYOUT, PF>), dim3(P / 256), dim3(256), 0, 0, W, X, P, N, spin);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<LAYOUT, PF>), dim3(P / 256), dim3(256), 0, 0, W, X, P, N, spin);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main()
{
    const long P = 262144;
    const int N = 1000;
    d2v *W; double *X;
    (void)hipMalloc(&W, sizeof(d2v) * P * N);
    (void)hipMalloc(&X, sizeof(double) * P * N * 2);
    (void)hipMemset(W, 0, sizeof(d2v) * P * N);
    const double gb = 48.0 * P * N / 1e9;
    for (int spin : {0, 40, 120}) {
        const float a = run<0, 4>(W, X, P, N, spin), b = run<1, 4>(W, X, P, N, spin), c = run<0, 8>(W, X, P, N, spin), d = run<1, 8>(W, X, P, N, spin);
        printf("spin %3d: SoA pf4 %.3f ms (%.0f GB/s) | blocked pf4 %.3f ms (%.0f GB/s) | SoA pf8 %.3f ms (%.0f GB/s) | blocked pf8 %.3f ms (%.0f GB/s)\n",
               spin, a, gb / a * 1e3, b, gb / b * 1e3, c, gb / c * 1e3, d, gb / d * 1e3);
    }
    return 0;
}
