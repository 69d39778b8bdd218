// What would spreading the pCN kernel's streams over ALL THREE 96-GiB pieces of the device memory buy (today: W in one piece, Xo in another)?
// Walks the allocator with contiguous 4-GiB runs until it holds one run in each of three pieces (classified with the library's
// four-run two-stream test), then times plain streaming kernels with the traffic mixes of the path kernels:
//   write-only (fresh proposals: X), read + write in place + write (the pCN chain step: W, Xo), with the streams in 1, 2 or 3 pieces.
//   hipcc --offload-arch=gfx950 -O2 scripts/three_piece_probe.hip -o /tmp/tpp && /tmp/tpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));
struct Trip { const d2v *r; d2v *w1, *w2; };   // r: read (may be null), w1: written (null: none), w2: written (null: none)
struct Trips { Trip t[3]; int n; int wide; };   // wide: w2 takes 32 bytes per element (read 16, write 16, write 32 = the FHN chain step's 8 : 8 : 16)
constexpr int GRID = 4104;   // divisible by 1, 2, 3

// block j works on triple j % n: reads r[i], writes w1[i] (= r when the update is in place) and w2[i]
__global__ __launch_bounds__(256) void k_mix(Trips T, size_t m)
{
    const Trip t = T.t[blockIdx.x % T.n];
    size_t i = (size_t)(blockIdx.x / T.n) * 256 + threadIdx.x;
    const size_t step = (size_t)(gridDim.x / T.n) * 256;
    for (; i < m; i += step) {
        d2v v = {(double)i, 1.0};
        if (t.r) { const d2v u = __builtin_nontemporal_load(t.r + i); v.x += 0.9 * u.x; v.y += 0.9 * u.y; }
        if (t.w1) __builtin_nontemporal_store(v, t.w1 + i);
        if (t.w2) {
            if (T.wide) { __builtin_nontemporal_store(v, t.w2 + i); __builtin_nontemporal_store(v, t.w2 + m + i); }   // the headline's mix: Xo is twice W (two coalesced rows)
            else __builtin_nontemporal_store(v, t.w2 + i);
        }
    }
}
static float run_ms(const Trips &T, size_t m)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_mix<<<GRID, 256>>>(T, m);
    std::vector<float> ms;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0); k_mix<<<GRID, 256>>>(T, m); hipEventRecord(e1); hipEventSynchronize(e1);
        float t = 0; hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    std::sort(ms.begin(), ms.end());
    return ms[2];
}
static float two_stream(void *a, void *b, size_t bytes)
{
    Trips T{}; T.n = 1; T.t[0] = Trip{nullptr, (d2v *)a, (d2v *)b};
    return (float)(2.0 * bytes / (run_ms(T, bytes / 16) * 1e6));
}
int main()
{
    const size_t BLK = (size_t)4 << 30, SP = (size_t)512 << 20;
    std::vector<char *> blk;
    std::vector<int> piece;
    std::vector<char *> rep;
    float r_same = 0;
    for (int k = 0; k < 64 && rep.size() < 3; k++) {
        size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot);
        if (fr < 3 * BLK) break;
        void *p = nullptr;
        if (hipExtMallocWithFlags(&p, BLK, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); break; }
        char *q = (char *)p;
        blk.push_back(q);
        if (k == 0) {
            std::vector<float> rr = {two_stream(q, q + BLK - SP, SP), two_stream(q, q + BLK / 2 - SP / 2, SP), two_stream(q + BLK / 2 - SP / 2, q + BLK - SP, SP)};
            std::sort(rr.begin(), rr.end()); r_same = rr[1];
            printf("one-piece two-stream rate %.0f GB/s (median of %.0f %.0f %.0f)\n", r_same, rr[0], rr[1], rr[2]);
            rep.push_back(q); piece.push_back(0); continue;
        }
        int pc = -1; bool all_apart = true;
        for (size_t j = 0; j < rep.size(); j++) {
            const float mean = 0.25f * (two_stream(rep[j], q, SP) + two_stream(rep[j], q + BLK - SP, SP) + two_stream(rep[j] + BLK - SP, q, SP) + two_stream(rep[j] + BLK - SP, q + BLK - SP, SP)) / r_same;
            if (mean <= 1.11f) { pc = (int)j; break; }
            if (mean < 1.14f) all_apart = false;
        }
        if (pc < 0 && all_apart) { pc = (int)rep.size(); rep.push_back(q); }
        piece.push_back(pc);
        printf("block %3d at %p: piece %d\n", k, (void *)q, pc);
    }
    if (rep.size() < 3) { printf("only %zu pieces found in %zu blocks\n", rep.size(), blk.size()); }
    for (char *q : blk) if (std::find(rep.begin(), rep.end(), q) == rep.end()) (void)hipFree(q);
    if (rep.size() < 2) return 0;
    char *A = rep[0], *B = rep[1], *C = rep.size() > 2 ? rep[2] : nullptr;
    const size_t TOT = (size_t)1536 << 20;   // bytes per logical stream (split over the triples of a run)
    auto gbs = [&](const Trips &T, int streams) { const size_t m = TOT / 16 / T.n; const float ms = run_ms(T, m); return (double)streams * (double)(m * 16) * T.n / (ms * 1e6); };
    auto T1 = [](Trip a) { Trips T{}; T.n = 1; T.t[0] = a; return T; };
    auto T2 = [](Trip a, Trip b) { Trips T{}; T.n = 2; T.t[0] = a; T.t[1] = b; return T; };
    auto T3 = [](Trip a, Trip b, Trip c) { Trips T{}; T.n = 3; T.t[0] = a; T.t[1] = b; T.t[2] = c; return T; };
    auto D = [](char *p, size_t off) { return (d2v *)(p + off); };
    const size_t H = BLK / 2;   // a 4-GiB run holds two streams of 1.5 GiB side by side
    printf("\n== write-only, 1.5 GiB in all (fresh proposals: X)\n");
    printf("one stream, one piece                         %7.0f GB/s\n", gbs(T1(Trip{nullptr, D(A, 0), nullptr}), 1));
    printf("halves to two pieces                          %7.0f GB/s\n", gbs(T2(Trip{nullptr, D(A, 0), nullptr}, Trip{nullptr, D(B, 0), nullptr}), 1));
    if (C) printf("thirds to three pieces                        %7.0f GB/s\n", gbs(T3(Trip{nullptr, D(A, 0), nullptr}, Trip{nullptr, D(B, 0), nullptr}, Trip{nullptr, D(C, 0), nullptr}), 1));
    printf("\n== two write streams of 1.5 GiB each (the placement's test)\n");
    printf("both in one piece                             %7.0f GB/s\n", gbs(T1(Trip{nullptr, D(A, 0), D(A, H)}), 2));
    printf("one per piece (two pieces)                    %7.0f GB/s\n", gbs(T1(Trip{nullptr, D(A, 0), D(B, 0)}), 2));
    if (C) printf("each striped over three pieces                %7.0f GB/s\n", gbs(T3(Trip{nullptr, D(A, 0), D(B, 0)}, Trip{nullptr, D(B, H), D(C, 0)}, Trip{nullptr, D(C, H), D(A, H)}), 2));
    printf("\n== the pCN chain step: read W, write W in place, write Xo (three streams of 1.5 GiB; bytes counted: 3 x)\n");
    printf("W and Xo in one piece                         %7.0f GB/s\n", gbs(T1(Trip{D(A, 0), D(A, 0), D(A, H)}), 3));
    printf("W in one piece, Xo in another (today)         %7.0f GB/s\n", gbs(T1(Trip{D(A, 0), D(A, 0), D(B, 0)}), 3));
    printf("halves crossed over two pieces                %7.0f GB/s\n", gbs(T2(Trip{D(A, 0), D(A, 0), D(B, 0)}, Trip{D(B, H), D(B, H), D(A, H)}), 3));
    if (C) printf("thirds rotated over three pieces              %7.0f GB/s\n", gbs(T3(Trip{D(A, 0), D(A, 0), D(B, 0)}, Trip{D(B, H), D(B, H), D(C, 0)}, Trip{D(C, H), D(C, H), D(A, H)}), 3));
    if (C) printf("W, W', Xo each in its own piece (W out of place) %4.0f GB/s\n", gbs(T1(Trip{D(A, 0), D(C, 0), D(B, 0)}), 3));
    {   // the headline's own mix: per element read 16 B of W, write 16 B of W, write 32 B of Xo (1 GiB of W, 2 GiB of Xo per run; bytes counted: 4 x 1 GiB)
        const size_t WB = (size_t)1 << 30;
        auto g4 = [&](Trips T) { T.wide = 1; const size_t m = WB / 16 / T.n; const float ms = run_ms(T, m); return 4.0 * (double)(m * 16) * T.n / (ms * 1e6); };
        printf("\n== the FitzHugh-Nagumo chain step's mix 8 : 8 : 16 (read W, write W, write Xo)\n");
        printf("W and Xo in one piece                         %7.0f GB/s\n", g4(T1(Trip{D(A, 0), D(A, 0), D(A, H)})));
        printf("W in one piece, Xo in another (today)         %7.0f GB/s\n", g4(T1(Trip{D(A, 0), D(A, 0), D(B, 0)})));
        if (C) printf("W in one piece, Xo's halves in the two others %7.0f GB/s\n", g4(T2(Trip{D(A, 0), D(A, 0), D(B, 0)}, Trip{D(A, H), D(A, H), D(C, 0)})));
        if (C) printf("thirds rotated over three pieces              %7.0f GB/s\n", g4(T3(Trip{D(A, 0), D(A, 0), D(B, 0)}, Trip{D(B, H), D(B, H), D(C, 0)}, Trip{D(C, H), D(C, H), D(A, H)})));
    }
    printf("\n== copy (read one, write another), bytes counted 2 x\n");
    printf("inside one piece                              %7.0f GB/s\n", gbs(T1(Trip{D(A, 0), D(A, H), nullptr}), 2));
    printf("across two pieces                             %7.0f GB/s\n", gbs(T1(Trip{D(A, 0), D(B, 0), nullptr}), 2));
    return 0;
}
