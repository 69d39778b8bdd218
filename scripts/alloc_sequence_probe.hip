// alloc_sequence_probe.hip -- round 4: in WHICH 96-GiB piece of the device memory do consecutive allocations of a fresh process land?
// N blocks of B GiB each (hipExtMallocWithFlags(hipDeviceMallocContiguous), all held), and for every block the time of two write
// streams, one into block 0 and one into block i: 4.7 TB/s when they share a piece, 6.1 TB/s when they do not
// (profiles/r4_placement_streams.txt).  argv: N B [plain]   (plain: hipMalloc instead of contiguous runs)
//   hipcc --offload-arch=gfx950 -O3 -o ab/alloc_sequence_probe scripts/alloc_sequence_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));
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
static float two(void *a, void *b, size_t m)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_two, dim3(4096), dim3(256), 0, 0, (d2v *)a, (d2v *)b, m);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_two, dim3(4096), dim3(256), 0, 0, (d2v *)a, (d2v *)b, m);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / 3;
}
int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int N = argc > 1 ? atoi(argv[1]) : 24;
    const double B = argc > 2 ? atof(argv[2]) : 4.0;
    const bool plain = argc > 3 && !strcmp(argv[3], "plain");
    const size_t bytes = (size_t)(B * 1024) << 20, m = ((size_t)1 << 30) / 16;   // streams of 1 GiB each
    std::vector<char *> blk;
    for (int i = 0; i < N; i++) {
        char *p = nullptr;
        hipError_t e = plain ? hipMalloc((void **)&p, bytes) : hipExtMallocWithFlags((void **)&p, bytes, hipDeviceMallocContiguous);
        if (e != hipSuccess) { (void)hipGetLastError(); printf("block %d: allocation failed\n", i); break; }
        blk.push_back(p);
    }
    printf("%zu blocks of %.1f GiB (%s): two 1-GiB write streams, one into block 0, one into block i -- GB/s\n", blk.size(), B, plain ? "hipMalloc" : "contiguous");
    for (size_t i = 1; i < blk.size(); i++) {
        const float ms = two(blk[0], blk[i], m);
        printf("  block %2zu at %p (%+8.2f GiB from block 0): %6.0f\n", i, (void *)blk[i], ((double)(blk[i] - blk[0])) / (1 << 30), 2.0 * (1 << 30) / ms / 1e6);
    }
    const float ms = two(blk[0], blk[0] + bytes / 2, m);
    printf("  block 0, its two halves: %6.0f\n", 2.0 * (1 << 30) / ms / 1e6);
    return 0;
}
