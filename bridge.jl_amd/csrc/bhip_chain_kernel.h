// bhip_chain_kernel.h -- the pCN Metropolis-Hastings iteration on the LINE layout (noise dimension m' = 1 or 2).
//
// Same arithmetic as k_paths<..., NOISE_PCN, ...> (it calls the same path_step), different chain-state layout.
// The slot layout of bhip_path_kernel.h keeps a chain's current W and its proposal Wo side by side (16 bytes per
// grid point) because neighbouring chains have different accept/reject parities; every iteration then moves the
// unchanged half as well: 16 B read + 16 B written per path-step for 8 + 8 algorithmic bytes.  This memory system
// moves whole 128-byte lines (profiles/r1_microbench.txt), so the waste can only be avoided if a chain's values of
// ONE parity half fill whole lines:
//
//     Wl[((h*nch + k)*ld + p)*16 + s*m' + c] = component c of W[(16/m')k + s] of chain p in half h
//                                                                      (a 128-byte line per (h, k, p): 16/m' grid points)
//
// A wave owns 64 chains.  Per chunk of 16/m' steps it reads, for every chain, only the line of that chain's CURRENT half
// -- cooperatively, 8 lanes per line, 8 instructions for the 64 lines, exactly as coalesced as a plain stream --
// transposes through a padded LDS tile so that each chain's lane finds its own 16 values, overwrites them in
// place with the proposal values as the steps are computed, and writes the tile cooperatively to the lines of
// the OTHER halves.  An accept flips the chain's parity bit as before.  HBM traffic per path-step: 8 B read +
// 8 B written for W (+ 8d for the proposal path) = the algorithmic bytes of SURVEY 8(d) mode M.
// Cost: a 64 x 17 double LDS tile per wave and 32 staging registers => 2 waves per SIMD (VGPRs are then plentiful).
#pragma once
#include "bhip_path_kernel.h"

namespace bhip {

constexpr int LINE_DOUBLES = 16;              // one 128-byte line
// components per grid point INSIDE a line: m' itself when it divides 16, else padded (m' = 3 -> 4: a line holds 4 grid points,
// the fourth slot of each is never read; 64 instead of 48 bytes of W per path-step each way -- still less than the 96 of the
// 16-byte slots, and the layout the wave-specialised kernels need)
constexpr int line_mpp(int mp) { return mp == 3 ? 4 : mp; }
constexpr int LINE_ROW = LINE_DOUBLES + 1;    // padded LDS row: lane L reads column s of row L without bank conflicts

// the two parity halves of a chain's chunk are NEIGHBOURS (a 256-byte pair; round 2 kept them 2 GB apart: every allocation 3-4 %
// slower, profiles/r3_alloc_placement.txt).  line_index(h, 0, j, ..) of a group's j-th chain is also the offset inside the
// group's contiguous block of chunk k, and flipping the half is `^ LINE_DOUBLES` (bhip_pc_kernel.h relies on both).
BHIP_DEV size_t line_index(int h, int k, long chain, int /*nch*/, long ld)
{
    return ((((size_t)k * ld + chain) * 2 + h)) * LINE_DOUBLES;
}

// BHIP_LINES_STAGE 1: the next chunk's lines are fetched into 32 staging registers while the current chunk is
// computed (2 waves per SIMD, latency hidden inside the wave); 0: fetched at the chunk boundary (fits the 128
// registers of 4 waves per SIMD, latency hidden by the other waves).
#ifndef BHIP_LINES_STAGE
#define BHIP_LINES_STAGE 1
#endif

template <class M, int GK, int MO, int FL, bool PPR = false /* per-chain coefficient rows */>
__global__ __launch_bounds__(256, BHIP_LINES_STAGE ? 2 : 4) void k_chain_lines(const KArgs a)
{
    constexpr int D = M::D, MP = M::MP;
    static_assert(MP >= 1 && MP <= 3, "a line holds 16/m' grid points (m' = 3 padded to 4)");
    constexpr int MPP = line_mpp(MP);
    constexpr int SPC = LINE_DOUBLES / MPP;  // grid points (steps) per chunk
    using RL = RowLayout<GK, D, MO, is_constdiff<M>::value>;
    extern __shared__ __attribute__((aligned(16))) double lds_tiles[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long c0 = ((long)blockIdx.x * 4 + wave) * 64;
    if (c0 >= a.P) return;                         // no block-level synchronisation anywhere below
    const bool live = c0 + lane < a.P;
    // lanes beyond the ensemble replicate chain P-1 exactly (same row of the tile, same stream): their stores
    // write identical values to identical addresses and need no execution mask
    const int row = live ? lane : (int)(a.P - 1 - c0);
    const long p = c0 + row;
    double *tile = lds_tiles + (size_t)wave * 64 * LINE_ROW;
    double *mine = tile + row * LINE_ROW;

    const M model(a.mpar);
    const int N = a.N, nsteps = N - 1, nll = N - 1 - a.skip;
    const int nch = (N + SPC - 1) / SPC;
    const cptr_t rows = (cptr_t)(uintptr_t)a.rows;
    LaneState<D, MP> st;
#pragma unroll
    for (int k = 0; k < D; k++) st.y[k] = a.x0_dev ? a.x0_dev[k * a.ldx0 + p] : a.x0[k];
    st.ll = 0.0; st.zq[0] = st.zq[1] = st.zq[2] = 0.0;
#pragma unroll
    for (int k = 0; k < MP; k++) { st.wprev[k] = 0.0; st.w2prev[k] = 0.0; }
    double *xout = nullptr;
    long ldx = 0;
    if constexpr ((FL & 1) != 0) { xout = a.Xo + p; ldx = a.ldC; }
    const uint32_t path = a.path0 + (uint32_t)p;
    const int c = a.cur[p];

    // cooperative mapping: instruction q moves the lines of chains c0 + 8q + lane/8; lane%8 selects 16 bytes of a line
    const int sub = lane >> 3, part = 2 * (lane & 7);
    int par[8];
#pragma unroll
    for (int q = 0; q < 8; q++) par[q] = a.cur[c0 + 8 * q + sub];   // cur[] is allocated (and zeroed) up to ld
    d2v stage[8];
    auto fetch = [&](int k) {
#pragma unroll
        for (int q = 0; q < 8; q++)
            stage[q] = ld_stream((const d2v *)(a.Wc + line_index(par[q], k, c0 + 8 * q + sub, nch, a.ldC) + part));
    };
    if (BHIP_LINES_STAGE) fetch(0);

    // the chunk loop, generic in the table accessor that carries the noise specification (bhip_rng.h)
    auto run_chunks = [&](const auto &tb) {
        using TB = typename bhip_unref<decltype(tb)>::type;
        // one Euler step i (grid point j = i + 1 = SPC*k + s): the chain's current W[j] comes from the tile, the proposal goes back
        auto step = [&](int i, int s, int i4 = -1 /* i & 3 where static */) {
            double wc[MP];
#pragma unroll
            for (int cc = 0; cc < MP; cc++) wc[cc] = mine[s * MPP + cc];
            if constexpr (PPR) {
                constexpr int NPP = pp_row_len<D>();
                double cr[NPP];
                const double *src = a.prows + (size_t)i * NPP * a.ldr + p;
#pragma unroll
                for (int q = 0; q < NPP; q++) cr[q] = src[(size_t)q * a.ldr];
                ExpRow<RL::LEN - 3> x;
                x.sh = rows + (size_t)i * RL::RS;
                expand_pp_row<M>(model, a.lna, cr, x.e);
                path_step<M, GK, MO, NOISE_PCN, FL, ExpRow<RL::LEN - 3>, TB>(model, a, x, i, nll, path, wc, nullptr, 0, xout, ldx, st, tb, 0u, i4);
            } else
                path_step<M, GK, MO, NOISE_PCN, FL, cptr_t, TB>(model, a, rows + (size_t)i * RL::RS, i, nll, path, wc, nullptr, 0, xout, ldx, st, tb, 0u, i4);
#pragma unroll
            for (int cc = 0; cc < MP; cc++) mine[s * MPP + cc] = st.wprev[cc];
        };

        for (int k = 0; k < nch; k++) {
            if (!BHIP_LINES_STAGE) fetch(k);
#pragma unroll
            for (int q = 0; q < 8; q++) {   // staged lines -> tile
                double *d = tile + (8 * q + sub) * LINE_ROW + part;
                d[0] = stage[q].x; d[1] = stage[q].y;
            }
            __builtin_amdgcn_wave_barrier();
            if (BHIP_LINES_STAGE && k + 1 < nch) fetch(k + 1);   // in flight while this chunk is computed
            const int j0 = k * SPC;
            if (k > 0 && j0 + SPC <= N) {
                // interior chunk: SPC valid steps, four at a time (j0 is a multiple of 4) so that the position of a step's normals
                // inside their Philox call is static
#ifndef BHIP_LINES_UNROLL
#define BHIP_LINES_UNROLL 1
#endif
#pragma unroll BHIP_LINES_UNROLL
                for (int s = 0; s < SPC; s += 4) {
                    step(j0 + s - 1, s, 3);
                    step(j0 + s, s + 1, 0);
                    step(j0 + s + 1, s + 2, 1);
                    step(j0 + s + 2, s + 3, 2);
                }
            } else {
                // first chunk (grid point 0 is W[0] = Wo[0] = 0, no step) and the ragged last chunk
#pragma unroll 1
                for (int s = 0; s < SPC; s += 2) {
                    const int i0 = j0 + s - 1;
                    if (i0 < 0) {
#pragma unroll
                        for (int cc = 0; cc < MP; cc++) mine[cc] = 0.0;
                    } else if (i0 < nsteps) step(i0, s);
                    if (i0 + 1 < nsteps) step(i0 + 1, s + 1);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 8; q++) {   // tile -> the lines of the other halves
                const double *d = tile + (8 * q + sub) * LINE_ROW + part;
                st_stream((d2v *)(a.Wc + line_index(par[q] ^ 1, k, c0 + 8 * q + sub, nch, a.ldC) + part), d2v{d[0], d[1]});
            }
            __builtin_amdgcn_wave_barrier();   // the tile is overwritten by the next chunk only after these reads
        }
    };
    run_chunks(IcdfConst());   // (the default stream only: under BHIP_OPT_NOISE_SPEC = 2 / 3 the host launches the wave-specialised kernel, do_launch)

    if constexpr (PPR) {
        if (a.uv_pc[p]) {
#pragma unroll
            for (int k = 0; k < D; k++) st.y[k] = a.vend_pc[(size_t)k * a.ldr + p];
        }
    } else if (a.use_vend) {   // endpoint(y, P::GuidedBridge) src/euler.jl:241-242
#pragma unroll
        for (int k = 0; k < D; k++) st.y[k] = a.vend[k];
    }
    if constexpr ((FL & 1) != 0) {
#pragma unroll
        for (int k = 0; k < D; k++) st_stream(&xout[((size_t)(N - 1) * D + k) * ldx], st.y[k]);
    }
    // if log(rand()) <= llo - ll: W <- Wo (parity flip), ll <- llo, acc += 1
    if (live) {
        if (!a.defer_accept) {
            const double u = accept_uniform(a.k0, a.k1, path, a.iter);
            if (det_log(u) <= st.ll - a.llcur[p]) {
                a.cur[p] = (unsigned char)(c ^ 1);
                a.llcur[p] = st.ll;
                a.acc[p] += 1u;
            }
        }
        if (a.ll) a.ll[p] = st.ll;
    }
}

// ---- BHIP_RTC_END  (above: device code, also embedded for hipRTC user models; below: host launch + layout conversion)

constexpr size_t CHAIN_LINES_LDS = sizeof(double) * 4 * 64 * LINE_ROW;   // one tile per wave, 4 waves per block: 34 KiB

template <class M, int GK, int MO, int FL, bool PPR>
hipError_t launch_chain_lines(const KArgs &a, hipStream_t st)
{
    const long grid = (a.P + 255) / 256;
    hipLaunchKernelGGL((k_chain_lines<M, GK, MO, FL, PPR>), dim3((unsigned)grid), dim3(256), CHAIN_LINES_LDS, st, a);
    return hipGetLastError();
}

// Layout conversion, both directions as a tiled transpose: a block owns 64 chains x one 16-value chunk, so that the
// SoA side is accessed 64 chains (512 bytes) at a time and the line side a whole line (8 lanes x 16 bytes) at a time.
//   k_soa_to_lines: plain SoA W [N][m'][ldW] -> half 0 of the line layout (chain initialisation)
//   k_lines_to_soa: the CURRENT halves of chains p0..p0+np -> plain SoA [N][m'][np]
// (line position sl of line k <-> grid point j = (16/mpp)*k + sl/mpp, component c = sl%mpp, mpp = line_mpp(m'); SoA row j*m' + c)
static __global__ __launch_bounds__(256) void k_soa_to_lines(const double *__restrict__ W, long ldW, int N, int mp, int nch, double *__restrict__ Wl, long ld, long P)
{
    __shared__ double tile[64 * LINE_ROW];
    const int k = blockIdx.y, t = threadIdx.x;
    const long c0 = (long)blockIdx.x * 64;
    const int mpp = line_mpp(mp), spc = LINE_DOUBLES / mpp;
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {   // line position sl = 4*rep + t/64 <-> SoA row (grid point, component), chain c0 + t%64
        const int sl = 4 * rep + (t >> 6), j = k * spc + sl / mpp, c = sl % mpp;
        const long p = c0 + (t & 63);
        tile[(t & 63) * LINE_ROW + sl] = (p < P && c < mp && j < N) ? W[((size_t)j * mp + c) * ldW + p] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {   // 16-byte pieces: chain (rep*256 + t)/8, part (rep*256 + t)%8
        const int idx = rep * 256 + t, cr = idx >> 3, part = 2 * (idx & 7);
        const double *d = tile + cr * LINE_ROW + part;
        *(d2v *)(Wl + line_index(0, k, c0 + cr, nch, ld) + part) = d2v{d[0], d[1]};
    }
}
static __global__ __launch_bounds__(256) void k_lines_to_soa(const double *__restrict__ Wl, const unsigned char *__restrict__ cur, int N, int mp, int nch, long ld,
                                                             long p0, long np, double *__restrict__ W)
{
    __shared__ double tile[64 * LINE_ROW];
    const int k = blockIdx.y, t = threadIdx.x;
    const long q0 = (long)blockIdx.x * 64;   // local chain index of the block's first chain
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        const int idx = rep * 256 + t, cr = idx >> 3, part = 2 * (idx & 7);
        if (q0 + cr < np) {
            const long p = p0 + q0 + cr;
            const d2v v = *(const d2v *)(Wl + line_index(cur[p], k, p, nch, ld) + part);
            tile[cr * LINE_ROW + part] = v.x; tile[cr * LINE_ROW + part + 1] = v.y;
        }
    }
    __syncthreads();
    const int mpp = line_mpp(mp), spc = LINE_DOUBLES / mpp;
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
        const int sl = 4 * rep + (t >> 6), j = k * spc + sl / mpp, c = sl % mpp;
        const long q = q0 + (t & 63);
        if (q < np && c < mp && j < N) W[((size_t)j * mp + c) * np + q] = tile[(t & 63) * LINE_ROW + sl];
    }
}

}  // namespace bhip
