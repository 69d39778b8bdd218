// HBM streaming rates by access mix on MI355X: read-only, write-only, copy (1:1) and the 1:2 read:write mix of
// the MCMC kernel, 4.2 GB per stream, fully coalesced, plain and non-temporal.  Context for the roofline
// fractions: a write-only kernel (independent proposals: 16 B written per path-step, nothing read) cannot reach
// the 8 TB/s headline, which needs reads and writes together.
//   hipcc --offload-arch=gfx950 -O3 scripts/hbm_rw_probe.hip -o /tmp/hrw && /tmp/hrw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));

template <int NR, int NW, bool NT>
__global__ __launch_bounds__(256) void k(const d2v *__restrict__ in, d2v *__restrict__ out, size_t n, size_t stride)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) {
        d2v acc = {1.0, 2.0};
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const d2v v = NT ? __builtin_nontemporal_load(in + r * stride + i) : in[r * stride + i];
            acc += v;
        }
#pragma unroll
        for (int w = 0; w < NW; w++) {
            d2v o = acc; o.x += w;
            if (NT) __builtin_nontemporal_store(o, out + w * stride + i); else out[w * stride + i] = o;
        }
    }
}

template <int NR, int NW, bool NT>
void run(const char *name, d2v *a, d2v *b, size_t n)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * 16;
    hipLaunchKernelGGL((k<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, a, b, n, n);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, a, b, n, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double gb = 16.0 * n * (NR + NW) / 1e9;
    printf("%-34s %s  %7.3f ms  %6.0f GB/s\n", name, NT ? "non-temporal" : "plain       ", ms, gb / ms * 1e3);
}

int main()
{
    const size_t n = 262144ull * 1000;   // 16-byte elements: 4.19 GB per stream
    d2v *a, *b;
    (void)hipMalloc(&a, 16 * n * 2);
    (void)hipMalloc(&b, 16 * n * 2);
    (void)hipMemset(a, 0, 16 * n * 2);
    (void)hipMemset(b, 0, 16 * n * 2);
    run<1, 0, false>("read only", a, b, n);   run<1, 0, true>("read only", a, b, n);
    run<0, 1, false>("write only", a, b, n);  run<0, 1, true>("write only", a, b, n);
    run<1, 1, false>("copy 1 read : 1 write", a, b, n);  run<1, 1, true>("copy 1 read : 1 write", a, b, n);
    run<1, 2, false>("1 read : 2 writes (MCMC mix)", a, b, n);  run<1, 2, true>("1 read : 2 writes (MCMC mix)", a, b, n);
    run<2, 1, false>("2 reads : 1 write", a, b, n);  run<2, 1, true>("2 reads : 1 write", a, b, n);
    return 0;
}
