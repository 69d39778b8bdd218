// hbm_rank_probe.hip -- round 4: what plain streaming kernels reach on MI355X when their streams lie in the SAME 96-GiB piece of the
// physical memory and when they lie in DIFFERENT pieces (profiles/r4_placement_regions.txt).  The ceiling the pCN kernel (1 read : 3
// writes of 2.1 GB each: read W, write Wo next to it, write Xo twice as large) is to be measured against.
// One physically contiguous 200-GiB block; the first cut is located by bisection with a copy kernel (src fixed at offset 0).
//   hipcc --offload-arch=gfx950 -O3 -o ab/hbm_rank_probe scripts/hbm_rank_probe.hip && ./ab/hbm_rank_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));

// streams: read r0 (n elements), write w0 (n), write w1 (2n) -- any of them may be null
__global__ __launch_bounds__(256) void k_mix(const d2v *__restrict__ r0, d2v *__restrict__ w0, d2v *__restrict__ w1, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) {
        d2v acc = {1.0, 2.0};
        if (r0) acc += __builtin_nontemporal_load(r0 + i);
        if (w0) __builtin_nontemporal_store(acc, w0 + i);
        if (w1) { __builtin_nontemporal_store(acc, w1 + i); acc.x += 1.0; __builtin_nontemporal_store(acc, w1 + n + i); }
    }
}
// two write streams of m elements each, interleaved per thread
__global__ __launch_bounds__(256) void k_two(d2v *__restrict__ a, d2v *__restrict__ b, size_t m)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < m; i += step) {
        const d2v v = {(double)i, 1.0};
        __builtin_nontemporal_store(v, a + i);
        __builtin_nontemporal_store(v, b + i);
    }
}
static float run(const void *r0, void *w0, void *w1, size_t n, int reps = 4)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mix, dim3(4096), dim3(256), 0, 0, (const d2v *)r0, (d2v *)w0, (d2v *)w1, n);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_mix, dim3(4096), dim3(256), 0, 0, (const d2v *)r0, (d2v *)w0, (d2v *)w1, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t GiB = (size_t)1 << 30, n = 131072ull * 1000;   // 16-byte elements: 2.1 GB per unit stream
    const double unit = 16.0 * n / 1e9;
    char *big = nullptr;
    CK(hipExtMallocWithFlags((void **)&big, 200 * GiB, hipDeviceMallocContiguous));
    auto at = [&](double gib) { return big + (size_t)(gib * 1024) * ((size_t)1 << 20); };
    // locate the first cut: copy from offset 0 to offset x
    const float k_near = run(at(0), at(3), nullptr, n), k_far = run(at(0), at(150), nullptr, n);
    printf("copy 2.1 GB -> 2.1 GB: neighbours %.4f ms, 150 GiB apart %.4f ms (a copy does not care)\n", k_near, k_far);
    // two write streams DO: locate the first cut with them
    const float c_near = run(nullptr, at(0), at(3), n), c_far = run(nullptr, at(0), at(150), n);
    printf("two writes (2.1 + 4.2 GB): neighbours %.4f ms (%.0f GB/s), 150 GiB apart %.4f ms (%.0f GB/s)\n", c_near, 3 * unit / c_near * 1e3, c_far, 3 * unit / c_far * 1e3);
    double lo = 3, hi = 150;
    const float thr = 0.5f * (c_near + c_far);
    const bool usable = std::max(c_near, c_far) > 1.03f * std::min(c_near, c_far);
    if (usable) {
        const bool near_slow = c_near > c_far;
        // first, find any point in the next piece within 100 GiB, then bisect
        for (double x = 8; x < 150; x += 8) { const bool s = run(nullptr, at(0), at(x), n) > thr; if (s != near_slow) { hi = x; break; } lo = x; }
        while (hi - lo > 0.26) { const double mid = (int)((lo + hi) / 2 * 4) / 4.0; const bool s = run(nullptr, at(0), at(mid), n) > thr; if (s == near_slow) lo = mid; else hi = mid; }
        printf("first cut seen by the two write streams: second stream at %.2f GiB still like a neighbour, at %.2f GiB not (cut ~ %.1f GiB into the block)\n", lo, hi, lo + 1.0);
    } else printf("the probe does not tell the pieces apart (%.1f %%)\n", 100.0 * (c_near / c_far - 1));
    const double cut = usable ? lo + 1.0 : 1e9;
    double A = 0.0, B = usable ? cut + 8 : 60.0, C = usable ? cut + 96 + 8 : 160.0;   // one offset inside each of three pieces
    if (C > 190.0) { C = cut - 96 + 8 > 0 ? cut - 96 + 8 : 190.0; printf("(third piece taken below the first cut / clamped: C = %.1f GiB)\n", C); }
    if (B > 190.0) B = 190.0;
    printf("offsets: A %.1f  B %.1f  C %.1f GiB\n", A, B, C);
    auto line = [&](const char *what, const void *r0, void *w0, void *w1, double units) {
        const float ms = run(r0, w0, w1, n);
        printf("%-86s %8.4f ms  %6.0f GB/s\n", what, ms, units * unit / ms * 1e3);
    };
    line("read 2.1 GB", at(A), nullptr, nullptr, 1);
    line("write 2.1 GB", nullptr, at(A), nullptr, 1);
    line("write 4.2 GB", nullptr, nullptr, at(A), 2);
    line("copy, source and destination in ONE piece", at(A), at(A + 3), nullptr, 2);
    line("copy, source and destination in two pieces", at(A), at(B), nullptr, 2);
    line("pCN mix (read 2.1, write 2.1 beside it, write 4.2), all in ONE piece", at(A), at(A + 2), at(A + 5), 4);
    line("pCN mix, the 4.2-GB write stream in a second piece", at(A), at(A + 2), at(B), 4);
    line("pCN mix, three pieces (read | write 2.1 | write 4.2)", at(A), at(B), at(C), 4);
    line("pCN mix, read + write 4.2 in one piece, write 2.1 in another", at(A), at(B), at(A + 3), 4);
    line("two writes (2.1 + 4.2) in ONE piece", nullptr, at(A), at(A + 3), 3);
    line("two writes (2.1 + 4.2) in two pieces", nullptr, at(A), at(B), 3);
    // how small can a two-write-stream probe be and still tell the pieces apart?  (both streams `m` 16-byte elements)
    printf("two equal write streams of S each: ONE piece vs two pieces\n");
    for (size_t mb : {(size_t)32, (size_t)64, (size_t)128, (size_t)256, (size_t)512, (size_t)1024}) {
        const size_t m = mb * ((size_t)1 << 20) / 16;
        float s1 = 1e30f, s2 = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto t = [&](void *a, void *b) {
                hipLaunchKernelGGL(k_mix, dim3(4096), dim3(256), 0, 0, (const d2v *)nullptr, (d2v *)a, (d2v *)nullptr, m);   // warm
                CK(hipEventRecord(e0));
                for (int r = 0; r < 4; r++) {
                    hipLaunchKernelGGL(k_two, dim3(4096), dim3(256), 0, 0, (d2v *)a, (d2v *)b, m);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 4;
            };
            s1 = std::min(s1, t(at(A), at(A + 3))); s2 = std::min(s2, t(at(A), at(B)));
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        printf("  S = %4zu MiB: one piece %.4f ms (%5.0f GB/s)   two pieces %.4f ms (%5.0f GB/s)   ratio %.3f\n", mb, s1, 2.0 * mb * 1.048576 / s1, s2, 2.0 * mb * 1.048576 / s2, s1 / s2);
    }
    CK(hipFree(big));
    return 0;
}
