// Cross-check of the library's inline Philox4x32-10 (bridge.jl_amd/csrc/bhip_rng.h) against rocRAND's own
// device generator (rocrand_state_philox4x32_10): the specification "bhip-philox-v1" is rocRAND's default
// generator with an explicit counter layout.
//   counter = (path, stream, iter, block), key = seed   <=>   rocrand_init(seed, subsequence = iter | block<<32,
//                                                                          offset = 4*(path | stream<<32))
// Build + run (GPU):  hipcc --offload-arch=gfx950 -O2 -I bridge.jl_amd/csrc tests/rocrand_check.hip -o /tmp/rc && /tmp/rc
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <cstdio>
#include "bhip_rng.h"

__global__ void k(unsigned long long seed, unsigned int *out_mine, unsigned int *out_rr, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned int path = 1000003u * t + 7u, iter = 31u * t, block = 17u * t + 1u, stream = t & 1u;
    const bhip::u32x4 a = bhip::philox4x32_10(path, stream, iter, block, (unsigned int)seed, (unsigned int)(seed >> 32));
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, (unsigned long long)iter | ((unsigned long long)block << 32), 4ull * ((unsigned long long)path | ((unsigned long long)stream << 32)), &st);
    const uint4 b = rocrand4(&st);
    out_mine[4 * t + 0] = a.x; out_mine[4 * t + 1] = a.y; out_mine[4 * t + 2] = a.z; out_mine[4 * t + 3] = a.w;
    out_rr[4 * t + 0] = b.x; out_rr[4 * t + 1] = b.y; out_rr[4 * t + 2] = b.z; out_rr[4 * t + 3] = b.w;
}

int main()
{
    const int n = 4096;
    unsigned int *dm, *dr;
    if (hipMalloc(&dm, 16 * n) != hipSuccess || hipMalloc(&dr, 16 * n) != hipSuccess) { printf("FAIL alloc\n"); return 2; }
    static unsigned int hm[4 * n], hr[4 * n];
    int bad = 0;
    for (unsigned long long seed : {0ull, 1ull, 0xDEADBEEFCAFEF00Dull}) {
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, seed, dm, dr, n);
        if (hipMemcpy(hm, dm, 16 * n, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hr, dr, 16 * n, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL copy\n"); return 2; }
        for (int i = 0; i < 4 * n; i++) bad += hm[i] != hr[i];
    }
    printf(bad ? "FAIL %d mismatches\n" : "OK rocRAND philox4x32_10 == bhip philox4x32_10 (%d words x 3 seeds)\n", bad ? bad : 4 * n);
    return bad ? 1 : 0;
}
