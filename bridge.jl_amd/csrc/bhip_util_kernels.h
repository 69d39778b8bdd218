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
// current W of the chains p0..p0+np out of the 16-byte slots (half cur[p]) into plain SoA [E][np]
__global__ void k_slots_to_soa(const double *__restrict__ slots, const unsigned char *__restrict__ cur, double *__restrict__ soa,
                               long E, long ld, long p0, long np)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * np) return;
    const long p = idx % np, e = idx / np;
    soa[e * np + p] = slots[(e * ld + p0 + p) * 2 + cur[p0 + p]];
}

// plain SoA [E][np] -> half 0 of the slots of all np = n chains (restoring a saved chain state: parity 0 everywhere)
__global__ void k_soa_to_slots(const double *__restrict__ soa, double *__restrict__ slots, long E, long ld, long np)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * np) return;
    const long p = idx % np, e = idx / np;
    slots[(e * ld + p) * 2] = soa[e * np + p];
}

// The tile-line layout of the d = 16/32 chains (bhip_tile_kernel.h): Wl[(((h*N + i)*T + t)*ld + p)*16 + pos], row = 16t + pos.
//   k_tlines_to_soa: the CURRENT halves of chains p0..p0+np -> plain SoA [N][d][np]  (rows d..16T-1 are padding, not copied)
//   k_soa_to_tlines: plain SoA [N][d][np] -> half 0 of all np = n chains, padding rows zeroed
// position of component q = 4r + kq of a row group inside its 128-byte tile line (bhip_tile_kernel.h)
#ifndef BHIP_TL_PAIR
#define BHIP_TL_PAIR 0      // (as in bhip_tile_kernel.h)
#endif
__device__ __forceinline__ int tl_pos(int q) { const int r = q >> 2, kq = q & 3; return 8 * (r >> 1) + 2 * kq + (r & 1); }
// offset of the line (half h, grid point i, row group t, chain p) of the tile-line layout
__device__ __forceinline__ size_t tl_line(int h, int i, int t, long p, int N, int T, long ld)
{
#if BHIP_TL_PAIR
    return ((((size_t)i * T + t) * ld + p) * 2 + h) * 16;
#else
    return ((((size_t)h * N + i) * T + t) * ld + p) * 16;
#endif
}
__global__ void k_tlines_to_soa(const double *__restrict__ Wl, const unsigned char *__restrict__ cur, double *__restrict__ soa,
                                int N, int d, int T, long ld, long p0, long np)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * d * np) return;
    const long p = idx % np, e = idx / np;
    const int i = (int)(e / d), row = (int)(e % d);
    soa[e * np + p] = Wl[tl_line(cur[p0 + p], i, row / 16, p0 + p, N, T, ld) + tl_pos(row % 16)];
}
__global__ void k_soa_to_tlines(const double *__restrict__ soa, double *__restrict__ Wl, int N, int d, int T, long ld, long np)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * 16 * T * np) return;
    const long p = idx % np, e = idx / np;
    const int i = (int)(e / (16 * T)), row = (int)(e % (16 * T));
    Wl[tl_line(0, i, row / 16, p, N, T, ld) + tl_pos(row % 16)] = row < d ? soa[((size_t)i * d + row) * np + p] : 0.0;
}

// sample!(W, Wiener{SVector{mp}}()):  W[0] = 0; W[i+1] = W[i] + rootdt[i]*xi   (time-major,
// component-minor normals, src/wiener.jl:24-35; test/with_srand.jl)
template <int MP>
__global__ __launch_bounds__(256) void k_wiener(const double *__restrict__ rootdt, int N, double *__restrict__ W, long ld, long P,
                                                uint32_t k0, uint32_t k1, uint32_t iter, uint32_t path0, int noise_spec,
                                                double *__restrict__ W1 = nullptr, double *__restrict__ W2 = nullptr, long wpart = 0 /* W in parts: paths [j*wpart, (j+1)*wpart) in buffer j */)
{
    // the generator's table in LDS like in the path kernels (round 6: the stand-alone sample! read it from constant memory with a different
    // index per lane -- 0.72 ms for 262 144 x 1000 normals where the fused proposal kernel, which draws the same normals AND solves, takes 0.75)
    __shared__ __attribute__((aligned(16))) double rng_tab[RNG_LDS_DOUBLES];
    if (noise_spec == 2 || noise_spec == 3) TabLDS::load(rng_tab, threadIdx.x, blockDim.x);
    else IcdfLDS::load(rng_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const uint32_t path = path0 + (uint32_t)p;
    // the noise specification (bhip_rng.h) travels in the table accessor's type; one wave-uniform branch per launch picks the body
    auto run = [&](const auto &tab) {
        double w[MP];
        const int part = wpart ? (int)(p / wpart) : 0;
        double *out = (part == 0 ? W : part == 1 ? W1 : W2) + (p - (long)part * wpart);
#pragma unroll
        for (int k = 0; k < MP; k++) { w[k] = 0.0; out[(size_t)k * ld] = 0.0; }
        out += (size_t)MP * ld;
        // four grid steps per iteration = 4*MP normals = MP whole quads (bhip_rng.h: normals 4q .. 4q+3): normal n = i*MP + k is
        // number n & 3 of quad n >> 2, so a quad is drawn once
        int i = 0;
        for (; i + 3 < N - 1; i += 4) {
            double z[4 * MP];
#pragma unroll
            for (int b = 0; b < MP; b++)
                normal_quad(tab, k0, k1, path, iter, (uint32_t)(i / 4 * MP + b), z[4 * b], z[4 * b + 1], z[4 * b + 2], z[4 * b + 3]);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const double rdt = rootdt[i + s];
#pragma unroll
                for (int k = 0; k < MP; k++) { w[k] = w[k] + rdt * z[s * MP + k]; __builtin_nontemporal_store(w[k], out + (size_t)k * ld); }
                out += (size_t)MP * ld;
            }
        }
        for (; i < N - 1; i++) {   // the last steps (fewer than four): single normals, the pair that holds them
            const double rdt0 = rootdt[i];
#pragma unroll
            for (int k = 0; k < MP; k++) {
                const int n = i * MP + k;
                double z0, z1;
                normal_pair(tab, k0, k1, path, iter, (uint32_t)(n >> 1), z0, z1);
                w[k] = w[k] + rdt0 * ((n & 1) ? z1 : z0);
                out[(size_t)k * ld] = w[k];
            }
            out += (size_t)MP * ld;
        }
    };
    if (noise_spec == 2) run(FullRes<TabLDS>(TabLDS(rng_tab)));
    else if (noise_spec == 3) run(TabLDS(rng_tab));
    else run(IcdfLDS(rng_tab));
}
// mp > 4 with mp % 4 == 0 (round 6): the components in blocks of four -- a block's normals (i*mp + 4kb .. + 3) are ONE Philox quad, number i*(mp/4) + kb --,
// the whole time loop per block with the block's four running sums in registers: every normal drawn once, W written once, nothing read back
// (the generic kernel below re-reads W[i] for every W[i+1] and draws by pairs from the constant-memory tables)
__global__ __launch_bounds__(256) void k_wiener_blocks(const double *__restrict__ rootdt, int N, int mp, double *__restrict__ W, long ld, long P,
                                                       uint32_t k0, uint32_t k1, uint32_t iter, uint32_t path0, int noise_spec)
{
    __shared__ __attribute__((aligned(16))) double rng_tab[RNG_LDS_DOUBLES];
    if (noise_spec == 2 || noise_spec == 3) TabLDS::load(rng_tab, threadIdx.x, blockDim.x);
    else IcdfLDS::load(rng_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const uint32_t path = path0 + (uint32_t)p;
    const int nq = mp >> 2;
    auto run = [&](const auto &tab) {
        for (int kb = 0; kb < nq; kb++) {
            double w[4] = {0.0, 0.0, 0.0, 0.0};
            double *out = W + (size_t)(4 * kb) * ld + p;
#pragma unroll
            for (int k = 0; k < 4; k++) out[(size_t)k * ld] = 0.0;
            for (int i = 0; i < N - 1; i++) {
                double z[4];
                normal_quad(tab, k0, k1, path, iter, (uint32_t)i * (uint32_t)nq + (uint32_t)kb, z[0], z[1], z[2], z[3]);
                const double rdt = rootdt[i];
                out += (size_t)mp * ld;
#pragma unroll
                for (int k = 0; k < 4; k++) { w[k] = w[k] + rdt * z[k]; __builtin_nontemporal_store(w[k], out + (size_t)k * ld); }
            }
        }
    };
    if (noise_spec == 2) run(FullRes<TabLDS>(TabLDS(rng_tab)));
    else if (noise_spec == 3) run(TabLDS(rng_tab));
    else run(IcdfLDS(rng_tab));
}
// mp > 4 (large-d Wiener), any mp: state kept in memory instead of registers
__global__ __launch_bounds__(256) void k_wiener_big(const double *__restrict__ rootdt, int N, int mp, double *__restrict__ W, long ld, long P,
                                                    uint32_t k0, uint32_t k1, uint32_t iter, uint32_t path0, int noise_spec)
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
            if ((n & 1) == 0) normal_pair_spec(noise_spec, k0, k1, path, iter, (uint32_t)(n >> 1), z, zc);
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

// Ensemble statistics {n, iters, sum acc, sum ll, sum ll^2, min ll, max ll, sum acc^2}: per-path values are
// reduced with wavefront shuffles, then across the block's waves through LDS; stage 1 leaves one
// partial per block, stage 2 (one block) folds the partials in a fixed order (deterministic sums).
struct StatAcc { double sa, sl, sl2, mn, mx, sa2; };
__device__ __forceinline__ void block_fold(StatAcc &v)
{
    __shared__ double sh[4][6];
    v.sa = wave_sum(v.sa); v.sl = wave_sum(v.sl); v.sl2 = wave_sum(v.sl2); v.sa2 = wave_sum(v.sa2); v.mn = wave_min(v.mn); v.mx = wave_max(v.mx);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh[w][0] = v.sa; sh[w][1] = v.sl; sh[w][2] = v.sl2; sh[w][3] = v.mn; sh[w][4] = v.mx; sh[w][5] = v.sa2; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 1; k < 4; k++) {
            v.sa += sh[k][0]; v.sl += sh[k][1]; v.sl2 += sh[k][2]; v.mn = fmin(v.mn, sh[k][3]); v.mx = fmax(v.mx, sh[k][4]); v.sa2 += sh[k][5];
        }
}
__global__ __launch_bounds__(256) void k_chain_stats_partial(const double *__restrict__ ll, const unsigned int *__restrict__ acc, long P,
                                                             double *__restrict__ part)
{
    StatAcc v = {0, 0, 0, INFINITY, -INFINITY, 0};
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
        const double l = ll[p], a = (double)acc[p];
        v.sa += a; v.sa2 += a * a; v.sl += l; v.sl2 += l * l; v.mn = fmin(v.mn, l); v.mx = fmax(v.mx, l);
    }
    block_fold(v);
    if (threadIdx.x == 0) {
        double *o = part + (size_t)blockIdx.x * 6;
        o[0] = v.sa; o[1] = v.sl; o[2] = v.sl2; o[3] = v.mn; o[4] = v.mx; o[5] = v.sa2;
    }
}
__global__ __launch_bounds__(256) void k_chain_stats_final(const double *__restrict__ part, int nparts, long P, double iters, double *__restrict__ stats)
{
    StatAcc v = {0, 0, 0, INFINITY, -INFINITY, 0};
    for (int b = threadIdx.x; b < nparts; b += blockDim.x) {
        const double *o = part + (size_t)b * 6;
        v.sa += o[0]; v.sl += o[1]; v.sl2 += o[2]; v.mn = fmin(v.mn, o[3]); v.mx = fmax(v.mx, o[4]); v.sa2 += o[5];
    }
    block_fold(v);
    if (threadIdx.x == 0) {
        stats[0] = (double)P; stats[1] = iters; stats[2] = v.sa; stats[3] = v.sl; stats[4] = v.sl2; stats[5] = v.mn; stats[6] = v.mx; stats[7] = v.sa2;
    }
}

// pointwise ensemble mean and M2 = sum outer(x - mean) of the current X at every grid point
// (what mcnext! accumulates, src/mclog.jl:48-56).  One block per grid index i; two passes.
__global__ __launch_bounds__(256) void k_path_stats(const double *__restrict__ X, int d, long ld, long P, double *__restrict__ mean, double *__restrict__ m2)
{
    const int i = blockIdx.x;
    __shared__ double sh[4];
    __shared__ double mu[32];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < d; k++) {
        double s = 0;
        for (long p = threadIdx.x; p < P; p += blockDim.x) s += X[((size_t)i * d + k) * ld + p];
        s = wave_sum(s);
        if (lane == 0) sh[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) { mu[k] = (sh[0] + sh[1] + sh[2] + sh[3]) / (double)P; mean[(size_t)i * d + k] = mu[k]; }
        __syncthreads();
    }
    for (int c = 0; c < d; c++)
        for (int r = 0; r < d; r++) {
            double s = 0;
            for (long p = threadIdx.x; p < P; p += blockDim.x)
                s += (X[((size_t)i * d + r) * ld + p] - mu[r]) * (X[((size_t)i * d + c) * ld + p] - mu[c]);
            s = wave_sum(s);
            if (lane == 0) sh[w] = s;
            __syncthreads();
            if (threadIdx.x == 0) m2[(size_t)i * d * d + r + d * c] = sh[0] + sh[1] + sh[2] + sh[3];
            __syncthreads();
        }
}

// merge a batch (nb samples: mean_b, m2_b) into the running pooled Welford state (na, mean_a, m2_a), per grid point:
// the parallel form of mcnext (src/mclog.jl:31-38), the same formula as bhip_welford_merge on the host
static __global__ void k_welford_merge(long entries, int d, double na, double *__restrict__ mean_a, double *__restrict__ m2_a, double nb,
                                       const double *__restrict__ mean_b, const double *__restrict__ m2_b)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= entries) return;
    const double n = na + nb;
    double delta[3];
    for (int k = 0; k < d; k++) delta[k] = mean_b[e * d + k] - mean_a[e * d + k];
    for (int c = 0; c < d; c++)
        for (int r = 0; r < d; r++) {
            const long q = e * d * d + r + d * c;
            m2_a[q] = m2_a[q] + m2_b[q] + delta[r] * delta[c] * (na * nb / n);
        }
    for (int k = 0; k < d; k++) mean_a[e * d + k] = mean_a[e * d + k] + delta[k] * (nb / n);
}

// ---- joint MH over chained segments (bhip_segchains.inc): start proposal, joint accept, commit + mcnext!
template <int D>
__global__ void k_seg_y0(long n, long ld, double w_old, double w_new, const double *__restrict__ y0, double *__restrict__ y0o,
                         uint32_t k0, uint32_t k1, uint32_t iter, uint32_t path0, KArgs geo /* x0 = mu, mpar[0..D*D) = chol (col-major) */,
                         const double *__restrict__ mu_pc, const double *__restrict__ chol_pc /* per-chain pi0 ([D][ld], [D*D][ld]) or null */,
                         const unsigned char *__restrict__ newblock /* smoothing.jl:166-167: after an adaptation y0o = y0 until the first accept */)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (newblock && newblock[p]) {
#pragma unroll
        for (int r = 0; r < D; r++) y0o[r * ld + p] = y0[r * ld + p];
        return;
    }
    double xi[D + 1], mu[D], ch[D * D];
#pragma unroll
    for (int k = 0; k < D; k++) mu[k] = mu_pc ? mu_pc[(size_t)k * ld + p] : geo.x0[k];
#pragma unroll
    for (int k = 0; k < D * D; k++) ch[k] = chol_pc ? chol_pc[(size_t)k * ld + p] : geo.mpar[k];
#pragma unroll
    for (int k = 0; k < D; k += 2) normal_pair_spec(geo.noise_spec, k0, k1, path0 + (uint32_t)p, iter, (uint32_t)(k >> 1), xi[k], xi[k + 1], 2u);
#pragma unroll
    for (int r = 0; r < D; r++) {
        double cz = ch[r] * xi[0];
#pragma unroll
        for (int c = 1; c < D; c++) cz += ch[r + D * c] * xi[c];
        const double z = mu[r] + cz;                                       // rand(pi0) = mu + C*randn   src/gaussian.jl:54
        y0o[r * ld + p] = mu[r] + w_new * (z - mu[r]) + w_old * (y0[r * ld + p] - mu[r]);
    }
}

// the same at any state dimension (d > 3: the MFMA tile kernel's chains): mu [d], chol [d*d] (column-major) in device memory
static __global__ void k_seg_y0_big(long n, long ld, int d, double w_old, double w_new, const double *__restrict__ y0, double *__restrict__ y0o,
                                    uint32_t k0, uint32_t k1, uint32_t iter, uint32_t path0, const double *__restrict__ mu,
                                    const double *__restrict__ chol, const unsigned char *__restrict__ newblock, int noise_spec)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (newblock && newblock[p]) {
        for (int r = 0; r < d; r++) y0o[r * ld + p] = y0[r * ld + p];
        return;
    }
    double xi[34];   // d <= 32
    for (int k = 0; k < d; k += 2) normal_pair_spec(noise_spec, k0, k1, path0 + (uint32_t)p, iter, (uint32_t)(k >> 1), xi[k], xi[k + 1], 2u);
    for (int r = 0; r < d; r++) {
        double cz = chol[r] * xi[0];
        for (int c = 1; c < d; c++) cz += chol[r + d * c] * xi[c];
        const double z = mu[r] + cz;                                       // rand(pi0) = mu + C*randn   src/gaussian.jl:54
        y0o[r * ld + p] = mu[r] + w_new * (z - mu[r]) + w_old * (y0[r * ld + p] - mu[r]);
    }
}

// commit (+ the running means of mcnext!) at any state dimension: blockIdx.y = grid point, blockIdx.z = segment
static __global__ void k_seg_commit_big(long n, long ld, int N, int m, int d, double *const *__restrict__ tab, int xo /* 0 or 4m: this iteration's Xo */,
                                        const unsigned char *__restrict__ accflag, int want_stats, double count)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, sg = blockIdx.z;
    if (p >= n) return;
    const double *__restrict__ Xo = tab[xo + sg];
    double *__restrict__ Xc = tab[m + sg];
    double *__restrict__ mean = want_stats ? tab[2 * m + sg] : nullptr;
    const bool a = accflag[p] != 0;
    for (int k = 0; k < d; k++) {
        const size_t e = ((size_t)i * d + k) * ld + p;
        const double x = a ? Xo[e] : Xc[e];
        if (a) Xc[e] = x;
        if (mean) { const double mk = mean[e]; mean[e] = mk + (x - mk) / (count + 1.0); }   // m += delta/(n+1)   src/mclog.jl:52
    }
}

// the same with the full per-chain mcnext! state (mean, m2) at any state dimension d <= 32: 64 chains per workgroup; a chain's
// delta = x - m and x - m_new (src/mclog.jl:50-54) wait in LDS ([k][lane]: conflict free) while the d*d entries of m2 stream
// through -- 16 d*d bytes per chain and grid point and iteration, the pass is as long as that state is large.
static __global__ void __launch_bounds__(64) k_seg_commit_big_m2(long n, long ld, int N, int m, int d, double *const *__restrict__ tab, int xo,
                                                                 const unsigned char *__restrict__ accflag, double count)
{
    __shared__ double sh_delta[32 * 64], sh_xm[32 * 64];
    const long p = (long)blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y, sg = blockIdx.z, t = threadIdx.x;
    if (p >= n) return;
    const double *__restrict__ Xo = tab[xo + sg];
    double *__restrict__ Xc = tab[m + sg];
    double *__restrict__ mean = tab[2 * m + sg];
    double *__restrict__ m2 = tab[3 * m + sg];
    const bool a = accflag[p] != 0;
    for (int k = 0; k < d; k++) {
        const size_t e = ((size_t)i * d + k) * ld + p;
        const double x = a ? Xo[e] : Xc[e];
        if (a) Xc[e] = x;
        const double mk = mean[e];
        const double delta = x - mk;
        const double mn = mk + delta / (count + 1.0);
        mean[e] = mn;
        sh_delta[k * 64 + t] = delta;
        sh_xm[k * 64 + t] = x - mn;
    }
    // a lane only reads what it wrote: no barrier
    for (int c = 0; c < d; c++) {
        const double xc = sh_xm[c * 64 + t];
        for (int r = 0; r < d; r++) {
            const size_t e = ((size_t)i * d * d + r + (size_t)d * c) * ld + p;
            m2[e] = m2[e] + sh_delta[r * 64 + t] * xc;
        }
    }
}

static __global__ void k_seg_accept(long n, long ld, int m, int d, const double *__restrict__ llo, double *__restrict__ ll,
                                    unsigned char *__restrict__ cur, unsigned int *__restrict__ acc, unsigned char *__restrict__ accflag,
                                    double *__restrict__ y0, const double *__restrict__ y0o, uint32_t k0, uint32_t k1, uint32_t iter, uint32_t path0,
                                    unsigned char *__restrict__ newblock, int doaccept /* smoothing.jl:156-158,193: the first adaptive proposal is accepted */,
                                    // time-blocked paths in a ring of buffers (bhip_segchains.inc): xcur[p] = the buffer that holds chain p's current
                                    // path, xtgt[p] = the one its proposal was written to; hist_cur [..][ld]: the current buffers after the iterations
                                    // whose mcnext! is still pending (this one is entry j); hist_prev [K][ld]: those of the batch whose statistics
                                    // pass may still be running (or null).  The next proposal goes to the lowest buffer none of them names.
                                    unsigned char *__restrict__ xcur, unsigned char *__restrict__ xtgt, unsigned char *__restrict__ hist_cur,
                                    const unsigned char *__restrict__ hist_prev, int j, int K)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double lls = 0.0;
    for (int i = 0; i < m; i++) lls += llo[i * ld + p] - ll[i * ld + p];   // ll += llikelihood(XXo[i]) - llikelihood(XX[i])
    const double u = accept_uniform(k0, k1, path0 + (uint32_t)p, iter);
    const bool ok = doaccept || det_log(u) <= lls;
    accflag[p] = ok ? 1 : 0;
    if (xcur) {
        const unsigned b = ok ? xtgt[p] : xcur[p];
        xcur[p] = (unsigned char)b;
        hist_cur[(size_t)j * ld + p] = (unsigned char)b;
        unsigned mask = 1u << b;
        for (int u = 0; u < j; u++) mask |= 1u << hist_cur[(size_t)u * ld + p];
        if (hist_prev)
            for (int u = 0; u < K; u++) mask |= 1u << hist_prev[(size_t)u * ld + p];
        xtgt[p] = (unsigned char)(__ffs((int)~mask) - 1);
    }
    if (ok) {
        if (newblock) newblock[p] = 0;
        cur[p] ^= 1;
        acc[p] += 1u;
        for (int i = 0; i < m; i++) ll[i * ld + p] = llo[i * ld + p];
        for (int k = 0; k < d; k++) y0[k * ld + p] = y0o[k * ld + p];
    }
}

// accepted chains: Xc <- Xo;  then (optionally) mcnext! of every chain with its current path:
//   delta = x - m; m += delta/(n+1); m2 += outer(delta, x - m)        src/mclog.jl:48-56
// (all m segments in ONE launch: blockIdx.z = segment, the per-segment arrays come from a device table
//  tab[0..m) = Xo of the even iterations, tab[m..2m) = Xc, tab[2m..3m) = mean, tab[3m..4m) = m2, tab[4m..5m) = Xo of the odd ones)
template <int D>
__global__ void k_seg_commit(long n, long ld, int N, int m, double *const *__restrict__ tab, int xo, const unsigned char *__restrict__ accflag, int want_stats, double count)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, sg = blockIdx.z;
    if (p >= n) return;
    const double *__restrict__ Xo = tab[xo + sg];
    double *__restrict__ Xc = tab[m + sg];
    double *__restrict__ mean = want_stats ? tab[2 * m + sg] : nullptr;
    double *__restrict__ m2 = want_stats ? tab[3 * m + sg] : nullptr;   // null: the means only (BHIP_SEGCHAINS_MCNEXT_MEAN)
    double x[D];
    const bool a = accflag[p] != 0;
#pragma unroll
    for (int k = 0; k < D; k++) {
        const size_t e = ((size_t)i * D + k) * ld + p;
        x[k] = a ? Xo[e] : Xc[e];
        if (a) Xc[e] = x[k];
    }
    if (mean) {
        double delta[D], xm[D];
#pragma unroll
        for (int k = 0; k < D; k++) {
            const size_t e = ((size_t)i * D + k) * ld + p;
            const double mk = mean[e];
            delta[k] = x[k] - mk;
            const double mn = mk + delta[k] / (count + 1.0);
            mean[e] = mn;
            xm[k] = x[k] - mn;
        }
        if (m2) {
#pragma unroll
            for (int c = 0; c < D; c++)
#pragma unroll
                for (int r = 0; r < D; r++) {
                    const size_t e = ((size_t)i * D * D + r + D * c) * ld + p;
                    m2[e] = m2[e] + delta[r] * xm[c];
                }
        }
    }
}

// ---- the same loop on TIME-BLOCKED paths (KArgs::Xtb, bhip_path_kernel.h): sixteen grid points of one chain are one 128-byte line and the
// chain's current path is the half its parity names -- an accept copies nothing and mcnext! reads exactly the current paths (with plain
// SoA paths and one lane per chain every line of the proposal AND of the current paths is fetched, and the lines of the accepted ones
// written back: 72 bytes per grid point at d = 3 where 24 are needed).
// mcnext! of every chain with its current paths of the last kk iterations (kk <= K; hist [K][ld]: the buffer that held chain p's current
// path after each of them), all segments in one launch -- DEFERRED statistics: the state of a grid point (mean, m2: 16 D (D + 1) bytes read
// and written) travels once per kk iterations instead of once per iteration, and an iteration in which the chain did not move reads no
// path (its line is still in the tile).  The updates are those of src/mclog.jl:48-56, one after the other in registers: same operations,
// same order, same bits as a pass per iteration.
// A work item = 64 chains x one block of sixteen grid points of one segment; the grid is a fixed number of workgroups that walk the items
// (gridDim.x apart): beside the wave-specialised proposal kernels, whose workgroups need most of a CU's LDS, only as many workgroups as
// fit next to them may be resident -- a grid of one workgroup per item would keep every CU's LDS in small pieces and the proposals waiting
// for the whole pass.  The 64 x D lines of an item and history entry come in whole -- eight lanes per line, fetched one entry ahead --
// into an LDS tile; thread (c, jq) updates grid points 4jq .. 4jq+3 of chain c, all loads of the state (plain SoA [N][D(*D)][ld],
// contiguous across the 64 chains) before the first use.   tab[0..m) = Xtb (buffer b at + b*half), tab[2m..3m) = mean, tab[3m..4m) = m2
template <int D>
__global__ __launch_bounds__(256) void k_seg_mcnext_tb(long n, long ld, int N, int m, double *const *__restrict__ tab, long half,
                                                       const unsigned char *__restrict__ hist, int kk, double count, long ngroups, int nblk)
{
    constexpr int ROW = 16 * D + 1;   // odd: conflict-free 8-byte accesses
    __shared__ double tile[64 * ROW];
    const int t = threadIdx.x;
    const int sub = t >> 3, part = 2 * (t & 7);   // 32 lines per instruction: chains sub, sub + 32 of component k
    const long items = ngroups * nblk * m;
    d2v nx[D][2];
    unsigned skn = 0u;   // bit q: the fetch of nx[.][q] was skipped (the chain's buffer is that of the entry before: the tile row stays)
    auto fetch = [&](long w, int u) {
        const long g = w % ngroups;
        const int blk = (int)((w / ngroups) % nblk), sg = (int)(w / (ngroups * nblk));
        const double *X = tab[sg];
        skn = 0u;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const long p = g * 64 + 32 * q + sub;   // (hist and the lines are allocated up to ld, a multiple of 64)
            const unsigned b = hist[(size_t)u * ld + p];
            if (u > 0 && hist[(size_t)(u - 1) * ld + p] == b) { skn |= 1u << q; continue; }
#pragma unroll
            for (int k = 0; k < D; k++) nx[k][q] = ld_stream((const d2v *)(X + (size_t)b * half + (((size_t)blk * D + k) * ld + p) * 16 + part));
        }
    };
    long w = blockIdx.x;
    if (w < items) fetch(w, 0);
    for (; w < items; w += gridDim.x) {
        const long g = w % ngroups;
        const int blk = (int)((w / ngroups) % nblk), sg = (int)(w / (ngroups * nblk));
        const int c = t & 63, jq = t >> 6;
        const long p = g * 64 + c;
        const bool mine = p < n;
        double *__restrict__ mean = tab[2 * m + sg];
        double *__restrict__ m2 = tab[3 * m + sg];   // null: the means only (BHIP_SEGCHAINS_MCNEXT_MEAN)
        double mk[4][D], q2[4][D * D];
        if (mine) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = min(blk * 16 + 4 * jq + u, N - 1);
#pragma unroll
                for (int k = 0; k < D; k++) mk[u][k] = mean[((size_t)i * D + k) * ld + p];
                if (m2) {
#pragma unroll
                    for (int e = 0; e < D * D; e++) q2[u][e] = m2[((size_t)i * D * D + e) * ld + p];
                }
            }
        }
        for (int h = 0; h < kk; h++) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (skn & (1u << q)) continue;
#pragma unroll
                for (int k = 0; k < D; k++) {
                    double *d = tile + (32 * q + sub) * ROW + k * 16 + part;
                    d[0] = nx[k][q].x; d[1] = nx[k][q].y;
                }
            }
            __syncthreads();
            if (h + 1 < kk) fetch(w, h + 1);
            else if (w + gridDim.x < items) fetch(w + gridDim.x, 0);
            if (mine) {
                const double cnt = count + (double)h;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = 4 * jq + u;
                    double x[D], delta[D], xm[D];
#pragma unroll
                    for (int k = 0; k < D; k++) x[k] = tile[c * ROW + k * 16 + j];
#pragma unroll
                    for (int k = 0; k < D; k++) {
                        delta[k] = x[k] - mk[u][k];
                        mk[u][k] = mk[u][k] + delta[k] / (cnt + 1.0);
                        xm[k] = x[k] - mk[u][k];
                    }
                    if (m2) {
#pragma unroll
                        for (int cc = 0; cc < D; cc++)
#pragma unroll
                            for (int r = 0; r < D; r++) q2[u][r + D * cc] = q2[u][r + D * cc] + delta[r] * xm[cc];
                    }
                }
            }
            __syncthreads();   // the tile is rewritten next
        }
        if (mine) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = blk * 16 + 4 * jq + u;
                if (i < N) {
#pragma unroll
                    for (int k = 0; k < D; k++) mean[((size_t)i * D + k) * ld + p] = mk[u][k];
                    if (m2) {
#pragma unroll
                        for (int e = 0; e < D * D; e++) m2[((size_t)i * D * D + e) * ld + p] = q2[u][e];
                    }
                }
            }
        }
    }
}

// time-blocked buffer `h` (per chain: cur[p] -- a parity or a ring index --, or 0 without cur) <-> plain SoA [N][d][ld]: the chains' current paths for whoever reads
// them in the library's common layout (getters, llikelihood under new proposals), and the initial paths the other way
static __global__ void k_tb_to_soa(long n, long ld, int N, int d, const double *__restrict__ Xtb, long half, const unsigned char *__restrict__ cur, double *__restrict__ Xs)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int blk = blockIdx.y;
    if (p >= n) return;
    const double *X = Xtb + (size_t)(cur ? cur[p] : 0) * half;
    for (int k = 0; k < d; k++)
        for (int j = 0; j < 16; j++) {
            const int i = blk * 16 + j;
            if (i < N) Xs[((size_t)i * d + k) * ld + p] = X[(((size_t)blk * d + k) * ld + p) * 16 + j];
        }
}
static __global__ void k_soa_to_tb(long n, long ld, int N, int d, const double *__restrict__ Xs, double *__restrict__ Xtb, long half, const unsigned char *__restrict__ cur)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int blk = blockIdx.y;
    if (p >= n) return;
    double *X = Xtb + (size_t)(cur ? cur[p] : 0) * half;
    for (int k = 0; k < d; k++)
        for (int j = 0; j < 16; j++) {
            const int i = blk * 16 + j;
            X[(((size_t)blk * d + k) * ld + p) * 16 + j] = i < N ? Xs[((size_t)i * d + k) * ld + p] : 0.0;
        }
}

// two write streams of m 16-byte elements each (placement of large chain ensembles, bhip_api.hip chains_place): on MI355X they run at
// ~4.7 TB/s when a and b lie in the same 96-GiB piece of the device memory and at ~6.1 TB/s when they do not
// (profiles/r4_placement_streams.txt) -- the quickest way to tell where an allocation lies
static __global__ __launch_bounds__(256) void k_two_write_streams(d2v *__restrict__ a, d2v *__restrict__ b, size_t m)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < m; i += step) {
        const d2v v = {(double)i, 1.0};
        __builtin_nontemporal_store(v, a + i);
        __builtin_nontemporal_store(v, b + i);
    }
}

}  // namespace bhip
