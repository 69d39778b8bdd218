// Reproducer attempt for the round-5 side finding: buffers of 0.8-2.5 MB allocated with
// hipExtMallocWithFlags(hipDeviceMallocContiguous) and freed between other work made later downloads of OTHER buffers of the process read
// page-sized stretches of zeros (seen inside the whole test suite only).  This program replays the pattern standalone: a population of
// plain hipMalloc "victim" buffers filled with a known non-zero pattern by a kernel; rounds of small contiguous allocations that are
// written by a kernel and freed, interleaved with plain allocations / frees of similar sizes; after every round every victim is copied to
// the host and compared.  Prints the first mismatches (offset, length of the zero run) or "no corruption in R rounds".
//     hipcc --offload-arch=gfx950 -O2 scripts/contig_small_probe.hip -o /tmp/contig_probe && /tmp/contig_probe [rounds] [flag 0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void fill(double *p, size_t n, double tag)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag + (double)(i % 4093) + 1.0;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    const bool flag = argc > 2 ? atoi(argv[2]) != 0 : true;
    const size_t sizes[] = {800 << 10, 1200 << 10, 1700 << 10, 2500 << 10, 96 << 10, 3 << 20};
    struct Buf { double *d; size_t n; double tag; };
    std::vector<Buf> victims;
    std::vector<double> host;
    srand(7);
    long bad = 0;
    for (int r = 0; r < rounds; r++) {
        // a few new victims per round (plain allocations, the sizes of the suite's ensembles), the oldest ones retire
        for (int k = 0; k < 3; k++) {
            Buf b; b.n = (sizes[rand() % 6] + (size_t)(rand() % 4096) * 8) / 8; b.tag = 1000.0 * (r * 3 + k + 1);
            if (hipMalloc((void **)&b.d, b.n * 8) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
            fill<<<64, 256>>>(b.d, b.n, b.tag);
            victims.push_back(b);
        }
        while (victims.size() > 24) { (void)hipFree(victims.front().d); victims.erase(victims.begin()); }
        // the suspected pattern: small contiguous runs, written, freed -- between plain allocations of similar size
        void *c[3] = {nullptr, nullptr, nullptr}, *pl = nullptr;
        for (int k = 0; k < 3; k++) {
            const size_t bytes = sizes[rand() % 4];
            hipError_t e = flag ? hipExtMallocWithFlags(&c[k], bytes, hipDeviceMallocContiguous) : hipMalloc(&c[k], bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); c[k] = nullptr; continue; }
            fill<<<64, 256>>>((double *)c[k], bytes / 8, -5.0);
            if (k == 1) { (void)hipMalloc(&pl, sizes[rand() % 6]); if (pl) fill<<<64, 256>>>((double *)pl, 1000, -7.0); }
        }
        (void)hipDeviceSynchronize();
        for (int k = 0; k < 3; k++) if (c[k]) (void)hipFree(c[k]);
        if (pl) (void)hipFree(pl);
        // every victim read back and checked
        for (const Buf &b : victims) {
            host.resize(b.n);
            if (hipMemcpy(host.data(), b.d, b.n * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("memcpy failed\n"); return 2; }
            size_t run = 0, start = 0;
            for (size_t i = 0; i < b.n; i++) {
                const double want = b.tag + (double)(i % 4093) + 1.0;
                if (host[i] != want) { if (!run) start = i; run++; }
                else if (run) { if (bad < 10) printf("round %d: victim of %zu B: %zu wrong doubles from offset %zu B (first value read %g)\n", r, b.n * 8, run, start * 8, host[start]); bad++; run = 0; }
            }
            if (run) { if (bad < 10) printf("round %d: victim of %zu B: %zu wrong doubles at its tail from offset %zu B\n", r, b.n * 8, run, start * 8); bad++; }
        }
    }
    if (!bad) printf("no corruption in %d rounds (%s small allocations, 24 live victims of 0.1-3 MB checked after every round)\n", rounds, flag ? "hipDeviceMallocContiguous" : "plain");
    else printf("%ld corrupted stretches in %d rounds (%s small allocations)\n", bad, rounds, flag ? "hipDeviceMallocContiguous" : "plain");
    return bad ? 1 : 0;
}
