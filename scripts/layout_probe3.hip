// Micro-benchmark 3 (memory pattern only, values are garbage): wave-blocked, time-chunked W layout
//   Wb[block of 64 chains][chunk][chain in block][half][8 doubles]
// accessed with 4 lanes per chain (lane l -> chain 16j + l/4, 16-byte piece l%4), i.e. every load/store
// instruction touches 16 lines and uses one 64-byte half of each.  A real kernel would transpose through
// LDS afterwards.  Question: does the memory system move only the touched 64-byte halves (32 B/path-step
// in total), or whole 128-byte lines (48 B, no better than the slot layout)?
//   hipcc --offload-arch=gfx950 -O3 scripts/layout_probe3.hip -o /tmp/lp3 && /tmp/lp3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_blk(double *Wb, double *X, const unsigned char *cur, long P, int N, int nt)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long p = wave * 64 + lane;
    const int nch = N / 8;
    char *base = (char *)Wb + (size_t)wave * nch * 8192;
    int off_r[4], off_w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int cl = 16 * j + (lane >> 2);
        const int c = cur[wave * 64 + cl];
        off_r[j] = (cl * 2 + c) * 64 + (lane & 3) * 16;
        off_w[j] = (cl * 2 + (c ^ 1)) * 64 + (lane & 3) * 16;
    }
    double *x = X + p;
    d2v nb[4], cb[4];
#pragma unroll
    for (int j = 0; j < 4; j++) nb[j] = nt ? __builtin_nontemporal_load((const d2v *)(base + off_r[j])) : *(const d2v *)(base + off_r[j]);
    double acc = 0;
    for (int ch = 0; ch < nch; ch++) {
#pragma unroll
        for (int j = 0; j < 4; j++) cb[j] = nb[j];
        const int nx = ch + 1 < nch ? ch + 1 : ch;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const d2v *a = (const d2v *)(base + (size_t)nx * 8192 + off_r[j]);
            nb[j] = nt ? __builtin_nontemporal_load(a) : *a;
        }
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const double v = (s & 1) ? cb[s >> 1].y : cb[s >> 1].x;
            acc += v;
            const long i = (long)ch * 8 + s;
            __builtin_nontemporal_store(v, &x[(i * 2 + 0) * P]);
            __builtin_nontemporal_store(acc, &x[(i * 2 + 1) * P]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d2v o = cb[j]; o.x += 1.0;
            d2v *a = (d2v *)(base + (size_t)ch * 8192 + off_w[j]);
            if (nt) __builtin_nontemporal_store(o, a); else *a = o;
        }
    }
}

int main()
{
    const long P = 262144;
    const int N = 992;
    double *Wb, *X; unsigned char *cur;
    (void)hipMalloc(&Wb, sizeof(double) * P * N * 2);
    (void)hipMalloc(&X, sizeof(double) * P * N * 2);
    (void)hipMalloc(&cur, P);
    (void)hipMemset(Wb, 0, sizeof(double) * P * N * 2);
    unsigned char *h = new unsigned char[P];
    for (long p = 0; p < P; p++) h[p] = (unsigned char)((p * 2654435761u >> 7) & 1);
    (void)hipMemcpy(cur, h, P, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int nt = 0; nt < 2; nt++) {
        hipLaunchKernelGGL(k_blk, dim3(P / 256), dim3(256), 0, 0, Wb, X, cur, P, N, nt);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_blk, dim3(P / 256), dim3(256), 0, 0, Wb, X, cur, P, N, nt);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 3;
        printf("wave-blocked half-line layout, nontemporal=%d: %.3f ms  (%.0f GB/s of 32 B/path-step; slot layout reference ~2.18 ms)\n",
               nt, ms, 32.0 * P * N / 1e9 / ms * 1e3);
    }
    return 0;
}
