// bhip_util_kernels.h -- small device kernels around the fused path kernel:
// layout conversion (AoS host order <-> SoA ensemble), stand-alone Wiener sampling (LOOP A,
// src/wiener.jl:24-58), ensemble statistics (acceptance / log-weight block that is all-gathered,
// and the pointwise mcstart/mcnext! mean & covariance of src/mclog.jl:22-56).
#pragma once
#include "bhip_rng.h"
#include <hip/hip_runtime.h>

namespace bhip {

// aos[p][e] (e = i*dim+k, E entries per path)  ->  soa[e*ld + p0+p]
__global__ void k_aos_to_soa(const double *__restrict__ aos, double *__restrict__ soa, long E, long ld, long p0, long np)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * np) return;
    const long p = idx % np, e = idx / np;
    soa[e * ld + p0 + p] = aos[p * E + e];
}
__global__ void k_soa_to_aos(const double *__restrict__ soa, double *__restrict__ aos, long E, long ld, long p0, long np)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * np) return;
    const long p = idx % np, e = idx / np;
    aos[p * E + e] = soa[e * ld + p0 + p];
}
// same, but each chain reads from the buffer its parity bit selects
__global__ void k_soa2_to_aos(const double *__restrict__ b0, const double *__restrict__ b1, const unsigned char *__restrict__ cur,
                              double *__restrict__ aos, long E, long ld, long p0, long np)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * np) return;
    const long p = idx % np, e = idx / np;
    const double *src = cur[p0 + p] ? b1 : b0;
    aos[p * E + e] = src[e * ld + p0 + p];
}

// sample!(W, Wiener{SVector{mp}}()):  W[0] = 0; W[i+1] = W[i] + rootdt[i]*xi   (time-major,
// component-minor normals, src/wiener.jl:24-35; test/with_srand.jl)
__global__ __launch_bounds__(256) void k_wiener(const double *__restrict__ rootdt, int N, int mp, double *__restrict__ W, long ld, long P,
                                                uint32_t k0, uint32_t k1, uint32_t iter, uint32_t path0)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const uint32_t path = path0 + (uint32_t)p;
    double w[4] = {0.0, 0.0, 0.0, 0.0};
    double zc = 0.0;
    for (int k = 0; k < mp; k++) W[(size_t)k * ld + p] = 0.0;
    for (int i = 0; i < N - 1; i++) {
        const double rdt = rootdt[i];
        for (int k = 0; k < mp; k++) {
            const int n = i * mp + k;
            double z;
            if ((n & 1) == 0) normal_pair(k0, k1, path, iter, (uint32_t)(n >> 1), z, zc);
            else z = zc;
            const double wn = w[k & 3] + rdt * z;
            w[k & 3] = wn;
            W[((size_t)(i + 1) * mp + k) * ld + p] = wn;
        }
    }
}
// mp > 4 (large-d Wiener): state kept in memory instead of registers
__global__ __launch_bounds__(256) void k_wiener_big(const double *__restrict__ rootdt, int N, int mp, double *__restrict__ W, long ld, long P,
                                                    uint32_t k0, uint32_t k1, uint32_t iter, uint32_t path0)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const uint32_t path = path0 + (uint32_t)p;
    double zc = 0.0;
    for (int k = 0; k < mp; k++) W[(size_t)k * ld + p] = 0.0;
    for (int i = 0; i < N - 1; i++) {
        const double rdt = rootdt[i];
        for (int k = 0; k < mp; k++) {
            const int n = i * mp + k;
            double z;
            if ((n & 1) == 0) normal_pair(k0, k1, path, iter, (uint32_t)(n >> 1), z, zc);
            else z = zc;
            W[((size_t)(i + 1) * mp + k) * ld + p] = W[((size_t)i * mp + k) * ld + p] + rdt * z;
        }
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}

// stats[8] = {n, iters, sum acc, sum ll, sum ll^2, min ll, max ll, sum acc^2}; one 256-thread block
__global__ __launch_bounds__(256) void k_chain_stats(const double *__restrict__ ll, const unsigned int *__restrict__ acc, long P, double iters,
                                                     double *__restrict__ stats)
{
    __shared__ double sh[4][5];
    double sa = 0, sl = 0, sl2 = 0, mn = INFINITY, mx = -INFINITY, sa2 = 0;
    for (long p = threadIdx.x; p < P; p += blockDim.x) {
        const double l = ll[p], a = (double)acc[p];
        sa += a; sa2 += a * a; sl += l; sl2 += l * l; mn = fmin(mn, l); mx = fmax(mx, l);
    }
    sa = wave_sum(sa); sl = wave_sum(sl); sl2 = wave_sum(sl2); sa2 = wave_sum(sa2); mn = wave_min(mn); mx = wave_max(mx);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ double sh2[4];
    if (lane == 0) { sh[w][0] = sa; sh[w][1] = sl; sh[w][2] = sl2; sh[w][3] = mn; sh[w][4] = mx; sh2[w] = sa2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) { sa += sh[k][0]; sl += sh[k][1]; sl2 += sh[k][2]; mn = fmin(mn, sh[k][3]); mx = fmax(mx, sh[k][4]); sa2 += sh2[k]; }
        stats[0] = (double)P; stats[1] = iters; stats[2] = sa; stats[3] = sl; stats[4] = sl2; stats[5] = mn; stats[6] = mx; stats[7] = sa2;
    }
}

// pointwise ensemble mean and M2 = sum outer(x - mean) of the current X at every grid point
// (what mcnext! accumulates, src/mclog.jl:48-56).  One block per grid index i; two passes.
__global__ __launch_bounds__(256) void k_path_stats(const double *__restrict__ b0, const double *__restrict__ b1, const unsigned char *__restrict__ cur,
                                                    int d, long ld, long P, double *__restrict__ mean, double *__restrict__ m2)
{
    const int i = blockIdx.x;
    __shared__ double sh[4];
    __shared__ double mu[32];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < d; k++) {
        double s = 0;
        for (long p = threadIdx.x; p < P; p += blockDim.x) {
            const double *src = (cur && cur[p]) ? b1 : b0;
            s += src[((size_t)i * d + k) * ld + p];
        }
        s = wave_sum(s);
        if (lane == 0) sh[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) { mu[k] = (sh[0] + sh[1] + sh[2] + sh[3]) / (double)P; mean[(size_t)i * d + k] = mu[k]; }
        __syncthreads();
    }
    for (int c = 0; c < d; c++)
        for (int r = 0; r < d; r++) {
            double s = 0;
            for (long p = threadIdx.x; p < P; p += blockDim.x) {
                const double *src = (cur && cur[p]) ? b1 : b0;
                s += (src[((size_t)i * d + r) * ld + p] - mu[r]) * (src[((size_t)i * d + c) * ld + p] - mu[c]);
            }
            s = wave_sum(s);
            if (lane == 0) sh[w] = s;
            __syncthreads();
            if (threadIdx.x == 0) m2[(size_t)i * d * d + r + d * c] = sh[0] + sh[1] + sh[2] + sh[3];
            __syncthreads();
        }
}

}  // namespace bhip
