// rank_probe.hip -- round 4: does a cheap two-buffer kernel see what the pCN kernel sees?
//
// The pCN kernel runs at 1.53 ms when its chain lines W and its proposal paths Xo lie in different 96-GiB regions of the MI355X's physical
// memory and at 1.78 ms when they share one (profiles/r4_placement_regions.txt).  To PLACE buffers by that rule the library has to
// learn which region a piece of physical memory belongs to; nothing reports physical addresses, so it has to be measured.  This probe:
// workgroups read 128-byte lines of window A and write 128-byte lines of window B at scattered positions (every access a DRAM row miss:
// the rate is set by how many banks serve the two windows together).
//   part 1: one contiguous 200-GiB block (hipDeviceMallocContiguous): probe(A at a, B at b) over a grid of offsets -> contrast, period
//   part 2: chunks from hipMemCreate (the virtual-memory API): cost of creating them, and their classes against chunk 0
//
//   hipcc --offload-arch=gfx950 -O3 -o rank_probe scripts/rank_probe.hip && ./rank_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// window = `lines` 128-byte lines; thread group of 8 lanes moves one line (16 bytes per lane); line index scattered by a multiplicative hash
__global__ __launch_bounds__(256) void k_probe(const d2 *__restrict__ A, d2 *__restrict__ B, unsigned lines, unsigned iters, unsigned mode)
{
    const unsigned g = (blockIdx.x * 256u + threadIdx.x) >> 3, part = threadIdx.x & 7u;
    const unsigned ngroups = gridDim.x * 32u;
    d2 acc = {0.0, 0.0};
    for (unsigned it = 0; it < iters; it++) {
        const unsigned q = it * ngroups + g;
        const unsigned la = (q * 2654435761u) % lines, lb = (q * 2246822519u + 977u) % lines;
        if (mode & 1u) { const d2 v = __builtin_nontemporal_load(A + (size_t)la * 8 + part); acc += v; }
        if (mode & 2u) __builtin_nontemporal_store(d2{(double)q, acc.x}, B + (size_t)lb * 8 + part);
        if (mode & 4u) __builtin_nontemporal_store(d2{acc.y, (double)q}, B + (size_t)((lb * 7u + 3u) % lines) * 8 + part);
    }
    if (acc.x == 1.2345e300) B[0] = acc;
}

static float probe(const void *A, void *B, size_t window, unsigned mode = 7u, int reps = 3)
{
    const unsigned lines = (unsigned)(window / 128);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps + 1; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe, dim3(2048), dim3(256), 0, 0, (const d2 *)A, (d2 *)B, lines, 16u, mode);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) best = std::min(best, ms);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}

int main(int argc, char **argv)
{
    const size_t GiB = (size_t)1 << 30, MiB = (size_t)1 << 20;
    const size_t window = (argc > 1 ? atol(argv[1]) : 256) * MiB;
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("free %zu MiB of %zu MiB; probe window %zu MiB, 2048 x 256 threads x 16 iterations = 1 Mi lines read + 2 Mi lines written per launch\n", fr >> 20, tot >> 20, window >> 20);
    {   // ---- part 1: one contiguous block
        const size_t S = 200 * GiB;
        char *big = nullptr;
        CK(hipExtMallocWithFlags((void **)&big, S, hipDeviceMallocContiguous));
        CK(hipMemset(big, 0, 8 * GiB));
        printf("== part 1: contiguous %zu GiB.  A at offset a, B at offset b; ms per launch (read A + 2 x write B | read only | write only)\n", S >> 30);
        for (size_t a : {(size_t)0, 40 * GiB, 130 * GiB})
            for (size_t b = 1; b < 199; b += (b < 40 ? 3 : 6)) {
                if (b * GiB == a) continue;
                const float t = probe(big + a, big + b * GiB, window);
                printf("a %3zu GiB  b %3zu GiB: %.4f ms", a >> 30, b, t);
                if (b % 24 == 1) printf("   (read only %.4f, writes only %.4f)", probe(big + a, big + b * GiB, window, 1u), probe(big + a, big + b * GiB, window, 6u));
                printf("\n");
            }
        // locate the boundaries seen from a = 0 by bisection on b (1 GiB resolution, then 64 MiB)
        const float t_same = probe(big, big + 2 * GiB, window), t_far = probe(big, big + 60 * GiB, window);
        printf("reference: same neighbourhood %.4f ms, 60 GiB away %.4f ms\n", t_same, t_far);
        auto same = [&](size_t boff) { return probe(big, big + boff, window) > 0.5f * (t_same + t_far); };
        if (t_same > 1.03f * t_far || t_far > 1.03f * t_same) {
            size_t lo = 2 * GiB, hi = 60 * GiB;   // same(lo) true-ish, same(hi) false-ish
            const bool slo = same(lo);
            while (hi - lo > 64 * MiB) { const size_t mid = (lo + hi) / 2 / (64 * MiB) * (64 * MiB); if (same(mid) == slo) lo = mid; else hi = mid; }
            printf("first boundary seen from offset 0: between %.3f and %.3f GiB\n", (double)lo / GiB, (double)hi / GiB);
            size_t lo2 = hi + 8 * GiB, hi2 = 199 * GiB;
            const bool s2 = same(lo2);
            if (same(hi2) != s2) {
                while (hi2 - lo2 > 64 * MiB) { const size_t mid = (lo2 + hi2) / 2 / (64 * MiB) * (64 * MiB); if (same(mid) == s2) lo2 = mid; else hi2 = mid; }
                printf("second boundary: between %.3f and %.3f GiB (distance %.3f GiB)\n", (double)lo2 / GiB, (double)hi2 / GiB, (double)(lo2 - lo) / GiB);
            } else printf("no second boundary below 199 GiB\n");
        } else printf("the probe does not separate the two cases (contrast < 3 %%)\n");
        CK(hipFree(big));
    }
    {   // ---- part 2: chunks from the virtual-memory API
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        size_t gmin = 0; CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
        const size_t C = std::max(window, gran);
        const int N = argc > 2 ? atoi(argv[2]) : 96;
        printf("== part 2: hipMemCreate chunks of %zu MiB (granularity: recommended %zu KiB, minimum %zu KiB), %d chunks\n", C >> 20, gran >> 10, gmin >> 10, N);
        char *va = nullptr; CK(hipMemAddressReserve((void **)&va, C * N, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> h(N);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < N; k++) { CK(hipMemCreate(&h[k], C, &prop, 0)); CK(hipMemMap(va + k * C, C, 0, h[k], 0)); }
        hipMemAccessDesc desc = {}; desc.location = prop.location; desc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, C * N, &desc, 1));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("create + map + set access: %.3f ms per chunk\n", dt / N * 1e3);
        CK(hipMemset(va, 0, C));
        printf("probe(chunk 0, chunk k), ms:");
        for (int k = 1; k < N; k++) printf("%s %.3f", k % 16 == 1 ? "\n " : "", probe(va, va + (size_t)k * C, C));
        printf("\nprobe(chunk k, chunk k) [same chunk], ms: %.3f %.3f\n", probe(va, va, C), probe(va + C, va + C, C));
        for (int k = 0; k < N; k++) { CK(hipMemUnmap(va + k * C, C)); CK(hipMemRelease(h[k])); }
        CK(hipMemAddressFree(va, C * N));
    }
    return 0;
}
