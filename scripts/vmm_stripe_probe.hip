// Can a buffer be CONSTRUCTED that lies in all three 96-GiB pieces of the device memory at once?  hipMemMap takes no offset into a
// physical handle (scripts/vmm_offset_probe.hip), but a virtual range can be assembled from many small handles.  This probe
//   1. finds one contiguous 4-GiB run in each of three pieces (as scripts/three_piece_probe.hip),
//   2. creates groups of 256 physical handles of 2 MiB (hipMemCreate), maps each group as 512 MiB and classifies the GROUP against the
//      three runs with two write streams (consecutive creations come from consecutive physical memory),
//   3. assembles two virtual ranges of 1.5 GiB whose 2-MiB chunks rotate over the three pieces,
//   4. times the write-only stream and the pCN mix (read W, write W, write Xo) on them against the contiguous runs.
//   hipcc --offload-arch=gfx950 -O2 scripts/vmm_stripe_probe.hip -o /tmp/vsp && /tmp/vsp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_mix(const d2v *r, d2v *w1, d2v *w2, size_t m)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < m; i += step) {
        d2v v = {(double)i, 1.0};
        if (r) { const d2v u = __builtin_nontemporal_load(r + i); v.x += 0.9 * u.x; v.y += 0.9 * u.y; }
        if (w1) __builtin_nontemporal_store(v, w1 + i);
        if (w2) __builtin_nontemporal_store(v, w2 + i);
    }
}
static double rate(const void *r, void *w1, void *w2, size_t bytes)   // GB/s over all streams, median of 5
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t m = bytes / 16;
    k_mix<<<4096, 256>>>((const d2v *)r, (d2v *)w1, (d2v *)w2, m);
    std::vector<float> ms;
    for (int k = 0; k < 5; k++) {
        (void)hipEventRecord(e0); k_mix<<<4096, 256>>>((const d2v *)r, (d2v *)w1, (d2v *)w2, m); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float t = 0; (void)hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    std::sort(ms.begin(), ms.end());
    const int n = (r ? 1 : 0) + (w1 ? 1 : 0) + (w2 ? 1 : 0);
    return (double)n * (double)bytes / (ms[2] * 1e6);
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t BLK = (size_t)4 << 30, SP = (size_t)512 << 20, CH = (size_t)2 << 20;
    // 1. one contiguous run per piece
    std::vector<char *> blk, rep, donor(3, nullptr);
    double r_same = 0;
    auto donors = [&]() { return (donor[0] != nullptr) + (donor[1] != nullptr) + (donor[2] != nullptr); };
    for (int k = 0; k < 64 && (rep.size() < 3 || donors() < 3); k++) {
        void *p = nullptr;
        if (hipExtMallocWithFlags(&p, BLK, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); break; }
        char *q = (char *)p; blk.push_back(q);
        if (k == 0) {
            std::vector<double> rr = {rate(nullptr, q, q + BLK - SP, SP), rate(nullptr, q, q + BLK / 2, SP), rate(nullptr, q + BLK / 2 - SP, q + BLK - SP, SP)};
            std::sort(rr.begin(), rr.end()); r_same = rr[1]; rep.push_back(q); continue;
        }
        int same = -1; bool apart = true;
        for (size_t j = 0; j < rep.size(); j++) {
            char *r = rep[j];
            const double mean = 0.25 * (rate(nullptr, r, q, SP) + rate(nullptr, r, q + BLK - SP, SP) + rate(nullptr, r + BLK - SP, q, SP) + rate(nullptr, r + BLK - SP, q + BLK - SP, SP)) / r_same;
            if (mean <= 1.11) { same = (int)j; break; }
            if (mean < 1.14) apart = false;
        }
        if (same < 0 && apart && rep.size() < 3) rep.push_back(q);
        else if (same >= 0 && !donor[same]) donor[same] = q;
    }
    for (char *q : blk) if (std::find(rep.begin(), rep.end(), q) == rep.end() && std::find(donor.begin(), donor.end(), q) == donor.end()) (void)hipFree(q);
    printf("one-piece two-stream rate %.0f GB/s; runs in %zu pieces after %zu allocations\n", r_same, rep.size(), blk.size());
    if (rep.size() < 3 || donors() < 3) { printf("donors %d\n", donors()); return 0; }
    // 2. groups of small handles
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const int PER = (int)(SP / CH);
    std::vector<std::vector<hipMemGenericAllocationHandle_t>> pool(3);
    int want = -1;
    for (int g = 0; g < 40; g++) {
        if (pool[0].size() >= 2u * PER && pool[1].size() >= 2u * PER && pool[2].size() >= 2u * PER) break;
        // steering: free the donor run of a piece whose pool is short -- do the next small creations fill the hole it leaves?
        if (want < 0 || pool[want].size() >= 2u * PER) {
            want = -1;
            for (int j = 2; j >= 0; j--) if (pool[j].size() < 2u * PER && donor[j]) { want = j; break; }
            if (want >= 0) { (void)hipFree(donor[want]); donor[want] = nullptr; printf("freed the donor run of piece %d\n", want); }
        }
        std::vector<hipMemGenericAllocationHandle_t> hs(PER);
        const double t0 = now_ms();
        for (int k = 0; k < PER; k++) CK(hipMemCreate(&hs[k], CH, &prop, 0));
        const double t1 = now_ms();
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, SP, 0, nullptr, 0));
        for (int k = 0; k < PER; k++) CK(hipMemMap((char *)va + (size_t)k * CH, CH, 0, hs[k], 0));
        CK(hipMemSetAccess(va, SP, &acc, 1));
        const double t2 = now_ms();
        double m[3]; int pc = -1;
        for (int j = 0; j < 3; j++) {
            m[j] = 0.5 * (rate(nullptr, rep[j], va, SP) + rate(nullptr, rep[j] + BLK - SP, va, SP)) / r_same;
        }
        const int lo = (int)(std::min_element(m, m + 3) - m);
        if (m[lo] <= 1.11 && m[(lo + 1) % 3] >= 1.14 && m[(lo + 2) % 3] >= 1.14) pc = lo;
        printf("group %2d: create %.1f ms, map %.1f ms; against the runs %.3f %.3f %.3f -> piece %d\n", g, t1 - t0, t2 - t1, m[0], m[1], m[2], pc);
        CK(hipMemUnmap(va, SP)); CK(hipMemAddressFree(va, SP));
        if (pc >= 0 && pool[pc].size() < 2u * PER) pool[pc].insert(pool[pc].end(), hs.begin(), hs.end());
        else for (auto h : hs) (void)hipMemRelease(h);
    }
    printf("pool: %zu %zu %zu chunks\n", pool[0].size(), pool[1].size(), pool[2].size());
    if (pool[0].size() < 2u * PER || pool[1].size() < 2u * PER || pool[2].size() < 2u * PER) return 0;
    // 3. two striped ranges of 1.5 GiB: chunk c of range s comes from piece (c + s) % 3
    const size_t TOT = (size_t)1536 << 20; const int NCH = (int)(TOT / CH);
    char *S[2]; size_t used[3] = {0, 0, 0};
    for (int s = 0; s < 2; s++) {
        void *va = nullptr;
        const double t0 = now_ms();
        CK(hipMemAddressReserve(&va, TOT, 0, nullptr, 0));
        for (int c = 0; c < NCH; c++) { const int pc = (c + s) % 3; CK(hipMemMap((char *)va + (size_t)c * CH, CH, 0, pool[pc][used[pc]++], 0)); }
        CK(hipMemSetAccess(va, TOT, &acc, 1));
        printf("striped range %d: %d chunks mapped in %.1f ms\n", s, NCH, now_ms() - t0);
        S[s] = (char *)va;
    }
    char *A = rep[0], *B = rep[1];
    printf("\n== write-only, 1.5 GiB\ncontiguous run (one piece)      %7.0f GB/s\nstriped over three pieces       %7.0f GB/s\n", rate(nullptr, A, nullptr, TOT), rate(nullptr, S[0], nullptr, TOT));
    printf("== two write streams\nboth in one piece               %7.0f GB/s\none per piece                   %7.0f GB/s\nboth striped                    %7.0f GB/s\n",
           rate(nullptr, A, A + BLK / 2, TOT), rate(nullptr, A, B, TOT), rate(nullptr, S[0], S[1], TOT));
    printf("== pCN mix: read W, write W, write Xo\nW and Xo in one piece           %7.0f GB/s\nW in one piece, Xo in another   %7.0f GB/s\nW and Xo striped                %7.0f GB/s\nW contiguous, Xo striped        %7.0f GB/s\n",
           rate(A, A, A + BLK / 2, TOT), rate(A, A, B, TOT), rate(S[0], S[0], S[1], TOT), rate(A, A, S[1], TOT));
    printf("== copy\nacross two pieces               %7.0f GB/s\nstriped to striped              %7.0f GB/s\n", rate(A, B, nullptr, TOT), rate(S[0], S[1], nullptr, TOT));
    return 0;
}
