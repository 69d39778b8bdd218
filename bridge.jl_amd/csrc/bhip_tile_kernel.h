// bhip_tile_kernel.h -- guided Euler-Maruyama + log-likelihood for LARGE state dimension
// (config C5: LinPro d = 32, SVector{32}; SURVEY 8(a) rows a4/a5/a7/a8 at d > 3).
//
// For d <= 3 a lane owns a path (bhip_path_kernel.h).  At d = 32 the state no longer fits a lane and
// the reference's work per step is five dense d x d mat-vecs
//     r  = Hm_i (nu_i - x)      guide: Hm = inv(Hdiamond_i) (GuidedBridge, src/guip.jl:192-193) or H_i (nuH)
//     bT = B (x - mu)           target LinPro drift            src/linpro.jl:80
//     bA = B~ (x - mu~) + beta~ auxiliary drift                src/guip.jl:434
//     g  = a r                  guiding term a*(...)           src/guip.jl:192
//     s  = sigma dW             _scale(dW, sigma)              src/euler.jl:264
// batched over paths they are a dense contraction, so a WAVE owns a 16-path tile and runs them on the
// fp64 matrix cores:  Y(d x 16) = M(d x d) X(d x 16) as v_mfma_f64_16x16x4_f64 chains.
//
// Round 5: on gfx950 fp64 MFMA and fp64 VALU work of a SIMD do not overlap (profiles/r1_mfma_valu_overlap.txt), so every vector
// instruction of the step costs MFMA time.  Both the target and the auxiliary are affine and everything but x and dW is
// path-independent, so the step is regrouped ON THE HOST (build_tile_data) into FOUR products whose accumulators start from
// path-independent vectors -- no vector arithmetic is left but the Wiener increment and the dot product of the log-likelihood:
//     r       = hnu_i + (-Hm_i) x                          hnu_i = Hm_i nu_i
//     bT - bA = c + (B - B~) x                             c = B~ mu~ - B mu - beta~                      (only this difference enters ll)
//     x_{i+1} = q_i + P_i x + sigma dW                     P_i = I + dt_i (B - a Hm_i),  q_i = dt_i (a Hm_i nu_i - B mu)
// and, r and bT - bA entering nothing but their dot product, that dot product as ONE quadratic form:
//     dot(bT - bA, r) = c0_i + x . (b_i + A_i x)           A_i = -(B - B~)' Hm_i,  b_i = (B - B~)' hnu_i - Hm_i' c,  c0_i = c . hnu_i
// THREE products -- A_i x, P_i x, sigma dW: 48 instead of 80 MFMAs -- and ~45 instead of ~165 vector instructions per wave and step
// besides the noise; two per-step matrices (A_i, P_i) travel to LDS instead of one.  Same algebra, another association: parity with the
// oracle stays tolerance-based here (the cancellation inside the quadratic form costs three digits at the stiff end of a GuidedBridge:
// 1e-13 relative, against the stated 1e-8 on ll).
// A component-wise user drift b(t, x) (hipRTC, UD::ON) takes the place of B (x - mu) as a vector term -- B = 0 in the formulas, + b in the
// difference, + dt b in the update -- and needs r itself for b . r: that instantiation keeps the two products r and bT - bA (four in all).
//
// Tile layout of every d x 16 operand (T = d/16 row tiles): lane (kq = lane>>4, j = lane&15) holds
// v[t][r] = element (row 16t + 4r + kq, path j).  This is at once the MFMA C/D layout of a result and
// the B-operand layout of the next mat-vec (K-slice ks = 4t + r), so chained products need no shuffles.
// Matrices are pre-arranged on the host in A-operand fragment order
//     Mf[(t'*4T + ks)*64 + lane] = M[16t' + (lane&15)][4ks + (lane>>4)]
// The path-independent per-step matrix Hm_i (8 KiB at d = 32) is streamed global -> LDS once per block
// and step (double-buffered, one barrier per step) and shared by the block's 4 waves; the four
// constant matrices live in LDS for the whole kernel.  Noise: every lane needs 8 normals per step; a
// Philox call yields the normals of four consecutive rows, which sit in the four lanes of the path's
// column, so each lane draws a quarter of the path's calls and they exchange through a small LDS
// transpose.  The per-path log-weight is reduced over the 4 row groups of a path with two wavefront shuffles.
//
// Numerics: MFMA accumulates with fused multiply-adds in k order and the guide solve is replaced by a
// product with the pre-inverted matrix, so parity with the oracle is tolerance-based here
// (tests: 1e-9 relative on paths, 1e-8 on ll), not bit-exact as for d <= 3.
#pragma once
#include "bhip_rng.h"
#include <hip/hip_runtime.h>

namespace bhip {

typedef double double4v __attribute__((ext_vector_type(4)));
constexpr int TILE_RNG_DOUBLES = ICDF_HOT_DOUBLES > RNG_TAB_DOUBLES ? ICDF_HOT_DOUBLES : RNG_TAB_DOUBLES;   // 640
constexpr int TILE_ZB = 16 * 18;   // doubles of noise-exchange buffer per wave (k_tile): 16 columns x 16 normals, row stride 18

// Timing experiments only (results become WRONG): bit0 no per-step barrier / matrix staging, bit1 no normal
// generation, bit2 no MFMA products, bit3 no chain-state loads, bit4 no chain-state stores (pCN instantiation).  scripts/gpu_tile_probe.py runs the variants (profiles/r1_tile_breakdown.txt).
#ifndef BHIP_TILE_EXP
#define BHIP_TILE_EXP 0
#endif

struct TArgs {
    const double *steps;   // [N-1][2*D*D + 2*D + 4]: A_i (user drift: -Hm_i), P_i in fragment order, b_i (hnu_i), q_i (natural order), dt_i, sqrt(dt_i), c0_i, 0
    const double *hdr;     // [N-1][2]: dt_i, sqrt(dt_i) (the same values, for the instantiations that load them directly)
    const double *cst;     // 2 fragment matrices (sigma, B - B~ [user drift only]), then vend and c = B~ mu~ - B mu - beta~ [user drift only] (D each)
    double x0[32];         // shared starting point (zero padded), passed by value: launches on one proposal do not interfere
    int dtrue;             // state dimension of the process (<= the kernel's D; the rest is zero padding, template PAD)
    int N, skip, use_vend, noise;   // noise: 0 = external W, 1 = fresh Philox, 2 = pCN chain step, 3 = llikelihood of a stored X (Win = X)
    long P;
    const double *Win; long ldWin;
    double *Wout; long ldWout;
    int wstride;           // fresh W store: 1 = plain SoA, 2 = half 0 of the chain slots (chain initialisation)
    double *X; long ldX;
    double *ll;
    uint32_t k0, k1, iter, path0;
    // pCN chain state, same scheme as the path-per-lane kernel (bhip_path_kernel.h, KArgs): 16-byte slots
    // Wc[((i*D + row)*ldC + p)*2 + half], parity cur[p] selects the current half, accept flips it
    double *Wc; long ldC;
    unsigned char *cur;
    double *llcur;
    unsigned int *acc;
    double rho, srho;
    // user-defined target drift at large d (bhip_model_define_components, hipRTC): parameters by value, grid times
    double upar[16];
    const double *tt;
    const double *x0_dev;   // optional per-path starting points [d][ldx0] (segment chaining: src/euler.jl:267 returns the end point)
    long ldx0;
    uint32_t blk0;          // offset of the noise stream in pairs (multi-segment chains: segment << 24), as KArgs::blk0
    int defer_accept;       // pCN: do not decide -- only report llo in `ll` (joint accept over segments, bhip_segchains_*)
    int noise_spec;         // 4 (default) / 3 / 2: the noise specification (bhip_rng.h), as KArgs::noise_spec
};
// The target drift of the built-in instantiations is LinPro's B(x - mu), one of the five MFMA products.  A hipRTC user process
// supplies its drift COMPONENT-WISE instead: UD::bk(k, t, x, par) = b_k(t, x, P), where x points to the path's whole state
// vector (gathered per wave in LDS -- a lane holds only 8 of a path's 32 components).
struct NoUserDrift { static constexpr bool ON = false; };
template <bool B> struct TileTag { static constexpr bool value = B; };
typedef double tile_d2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) double *tile_cptr_t;   // constant address space: wave-uniform reads go through the scalar unit
// the ensembles and the chain state are streamed: written / read once per launch (same hint as the d <= 3 kernels)
#ifndef BHIP_TILE_NT
#define BHIP_TILE_NT 0
#endif
// pCN chains: the chain's current W[i+2] is fetched global -> LDS directly (global_load_lds_dwordx4, no staging registers)
// while step i is computed -- a full step of latency cover; 0: plain loads at the top of the step that consumes them
#ifndef BHIP_TILE_SPREAD
#define BHIP_TILE_SPREAD 2
#endif
#ifndef BHIP_TILE_DMA_LATE
#define BHIP_TILE_DMA_LATE 0
#endif
#ifndef BHIP_TILE_MDMA
#define BHIP_TILE_MDMA 1
#endif
#ifndef BHIP_TILE_SB_MASK
#define BHIP_TILE_SB_MASK 0x106   // what may still move across a store group: LDS reads (the next fragments), VALU, SALU
#endif
#ifndef BHIP_TILE_SPREAD_EVERY
#define BHIP_TILE_SPREAD_EVERY 2
#endif
#ifndef BHIP_TL_PAIR
#define BHIP_TL_PAIR 0      // tile lines: 1 = the parity halves of a chain's line as a 256-byte pair (what the d <= 3 lines gained 3-4 % from): measured at d = 32, 16.47-16.54 vs 16.40 ms -- not kept
#endif
#ifndef BHIP_TILE_LDSDMA
#define BHIP_TILE_LDSDMA 1
#endif
template <class V> __device__ __forceinline__ void tile_st(V *p, V v) { if constexpr (BHIP_TILE_NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <class V> __device__ __forceinline__ V tile_ld(const V *p) { if constexpr (BHIP_TILE_NT) return __builtin_nontemporal_load(p); else return *p; }

struct TileNoHook { __device__ __forceinline__ void operator()(int) const {} };
// PADK (zero-padded processes, template PAD of k_tile): K-slices nks .. 4T-1 hold nothing but the zero padding of the operand and of the
// matrix's columns -- their products add +0.0 and are skipped behind a wave-uniform branch (d = 9..12 on the 16-row tile: three k-steps
// instead of four)
// acc += M v (acc in the MFMA C/D layout of the head of the file; the caller initialises it -- zero, or a path-independent vector)
template <int T, class HOOK = TileNoHook, bool PADK = false>
__device__ __forceinline__ void tile_mv_acc(const double *__restrict__ Mf, const double (&v)[T][4], double4v (&acc)[T], int lane, HOOK hook = HOOK(), int nks = 4 * T)
{
    if constexpr ((BHIP_TILE_EXP & 4) != 0) {
#pragma unroll
        for (int tp = 0; tp < T; tp++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[tp][r] += v[tp][r] * Mf[lane];
        return;
    }
#pragma unroll
    for (int ks = 0; ks < 4 * T; ks++) {
        if (!PADK || ks < nks) {
#pragma unroll
            for (int tp = 0; tp < T; tp++)
                acc[tp] = __builtin_amdgcn_mfma_f64_16x16x4f64(Mf[(tp * 4 * T + ks) * 64 + lane], v[ks >> 2][ks & 3], acc[tp], 0, 0, 0);
        }
        hook(ks);   // (statically unrolled: ks is a constant in the hook)
    }
}

// TDA (round 6; user drift only): a TIME-DEPENDENT auxiliary beside a component-wise user drift -- B~(t), beta~(t) are functions of t for
// any dimension (src/partialbridge.jl:13-15, src/linpro.jl:181-204) -- : the step row carries a THIRD matrix, -B~_i in fragment order, and
// the vector c_i = B~_i mu~ - beta~_i behind its scalars, streamed to LDS with the row like A_i and P_i, in place of the two constants
// of the kernel (cst's second matrix and last vector).  Same four products, 8 KiB more per step and block at d = 32.
constexpr int tile_step_doubles(int D, bool tda) { return (tda ? 3 : 2) * D * D + (tda ? 3 : 2) * D + 4; }
template <int D, int NOISE, bool PAD = false, class UD = NoUserDrift, bool TDA = false>
__global__ __launch_bounds__(256, 2) void k_tile(const TArgs a)   // 2 waves per SIMD: <= 256 VGPR+AGPR
{
    static_assert(!TDA || UD::ON, "the per-step auxiliary matrix exists beside a user drift only (a LinPro target folds B~_i into A_i, b_i, c0_i)");
    constexpr int T = D / 16;
    constexpr int DD = D * D;
    constexpr int STEP = tile_step_doubles(D, TDA);   // A_i | -Hm_i, P_i (fragment order), b_i | hnu_i, q_i, dt_i, sqrt(dt_i), c0_i, 0 [, -B~_i (fragment order), c_i]
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *cm = lds;                  // 2*DD fragment matrices + 2*D vectors
    double *hb = lds + 2 * DD + 2 * D; // 2 * STEP
    double *rtab_lds = hb + 2 * STEP;  // the generator's table (TILE_RNG_DOUBLES), for the noise-drawing instantiations
    double *xs_lds = rtab_lds + TILE_RNG_DOUBLES;   // UD::ON: the state vectors of the block's 64 paths, [4 waves][16 paths][D]
    double *zb_lds = xs_lds + (UD::ON ? 64 * D : 0);   // noise exchange: per wave 16 paths x 16 normals (row stride 18: conflict-free)
    double *wb_lds = zb_lds + 4 * TILE_ZB;             // NOISE == 2: per wave 2T x 64 16-byte pieces of the chain's current W (LDS-DMA target)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, j = lane & 15;
    const long p_raw = (long)blockIdx.x * 64 + wave * 16 + j;
    const bool live = p_raw < a.P;
    const long p = live ? p_raw : a.P - 1;
    const int N = a.N, nsteps = N - 1, nll = N - 1 - a.skip;
    // PAD: the process has dtr < D components (even, so that a Philox block never straddles two grid points); rows
    // dtr..D-1 are zero padding of every matrix and vector, hold no data in the ensembles and are never loaded or stored
    const int dtr = PAD ? a.dtrue : D;
    auto ok = [&](int t, int r) { return !PAD || 16 * t + 4 * r + kq < dtr; };
    const int nks = PAD ? (dtr + 3) >> 2 : 4 * T;   // live K-slices of the matrix products (tile_mv)

    for (int c = tid; c < 2 * DD + 2 * D; c += 256) cm[c] = a.cst[c];
    for (int c = tid; c < STEP; c += 256) hb[c] = a.steps[c];
    // the generator's table of the launch's noise specification: v4 (the default) the 128 rows of the near octaves (IcdfLDSHot), v3 / v2 the
    // log + sincos tables
    const bool icdf = !(a.noise_spec == 2 || a.noise_spec == 3);
    if constexpr (NOISE == 1 || NOISE == 2) {
        if (icdf) IcdfLDSHot::load(rtab_lds, tid, 256);
        else TabLDS::load(rtab_lds, tid, 256);
    }
    __syncthreads();
    const TabLDS rtab(rtab_lds);
    bool cold = false;
    const double *Sf = cm, *Dmf_c = cm + DD;                     // sigma; B - B~ (user drift with a time-constant auxiliary only)
    const double *vend = cm + 2 * DD, *cvec_c = vend + D;        // V[N-1]; B~ mu~ - B mu - beta~ (the same)

    // Addressing: the lane's element (t, r) of grid row i lives at  base + i*D*ld + (4t + r)*(4*ld), base = array +
    // kq*ld + p.  One running pointer per array is advanced once per step and the 4T elements are reached by a
    // chain of additions of the wave-uniform stride 4*ld (one VALU each) instead of a 64-bit multiply per access.
    // Lanes beyond the ensemble (p_raw >= P) replicate path P-1 exactly (same path id, same loads), so their
    // stores write identical values to identical addresses and need no execution mask.
    const size_t rsWin = 4 * (size_t)a.ldWin, rsX = 4 * (size_t)a.ldX, rsWo = 4 * (size_t)a.ldWout * a.wstride;
    const double *winp = (NOISE == 0 || NOISE == 3) ? a.Win + (size_t)kq * a.ldWin + p : nullptr;
    double *xp = a.X ? a.X + (size_t)kq * a.ldX + p : nullptr;
    double *wop = (NOISE == 1 && a.Wout) ? a.Wout + ((size_t)kq * a.ldWout + p) * a.wstride : nullptr;
    // pCN chain state, TILE-LINE layout: a 128-byte line per (parity half h, grid point i, row group t, chain p) holding
    // the 16 components 16t .. 16t+15 of W[i]:   Wl[(((h*N + i)*T + t)*ld + p)*16 + tl_pos(4r + kq)],
    // tl_pos(4r + kq) = 8(r>>1) + 2kq + (r&1): a lane's values r = 2j, 2j+1 are neighbours (one 16-byte access) and the four
    // lanes kq = 0..3 of a chain cover a contiguous 64-byte half line per instruction, the line in two.
    // A lane reads its 8 values of the chain's CURRENT half and writes the proposal to the OTHER half (the accept flips the
    // chain's parity bit), so that whole lines travel: 8 m' B read + 8 m' B written per path-step -- the algorithmic bytes
    // (the 16-byte slots of round 1 moved the unchanged half too: 1280 instead of 768 B per path-step at d = 32).
    // BHIP_TL_PAIR (round 4): the two parity halves of a chain's line are NEIGHBOURS, Wl[((((i*T + t)*ld + p)*2 + h)*16 + ..] -- a 256-byte
    // pair, as the d <= 3 line layout has had since round 3: the read of the current half and the write of the other fall into the
    // same DRAM row (0: the halves 17 GB apart, Wl[(((h*N + i)*T + t)*ld + p)*16 + ..])
    constexpr size_t TL_LINE = BHIP_TL_PAIR ? 32 : 16;                                // doubles from a chain's line to the next chain's
    const size_t tl_grid = (size_t)T * a.ldC * TL_LINE, tl_half = BHIP_TL_PAIR ? (size_t)16 : (size_t)N * tl_grid;   // doubles per grid point / between the halves
    const double *wrd = nullptr;
    double *wwr = nullptr, *wst = nullptr;   // wst: where the step in flight stores its proposal line (BHIP_TILE_SPREAD)

    double x[T][4], wprev[T][4];
    {
        const double *q = winp;
        double *qo = wop;
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                x[t][r] = (a.x0_dev && ok(t, r)) ? a.x0_dev[(size_t)(16 * t + 4 * r + kq) * a.ldx0 + p] : a.x0[16 * t + 4 * r + kq];
                wprev[t][r] = (NOISE == 0 && ok(t, r)) ? *q : 0.0;
                if (NOISE == 0) q += rsWin;
                if (NOISE == 1 && a.Wout) { if (ok(t, r)) *qo = 0.0; qo += rsWo; }
            }
    }
    if (NOISE == 0) winp += (size_t)dtr * a.ldWin;                       // -> W[1]
    if (NOISE == 1 && a.Wout) wop += (size_t)dtr * a.ldWout * a.wstride;
    double ll = 0.0;
    const uint32_t path = a.path0 + (uint32_t)p;
    // pCN: W2 accumulates the fresh Wiener path, wprev the proposal Wo = rho*W + sqrt(1-rho^2)*W2
    double w2prev[NOISE == 2 ? T : 1][4];
    int cpar = 0;
    if constexpr (NOISE == 2) {
        cpar = a.cur[p];
        wrd = a.Wc + (size_t)cpar * tl_half + (size_t)p * TL_LINE + 2 * kq;
        wwr = a.Wc + (size_t)(cpar ^ 1) * tl_half + (size_t)p * TL_LINE + 2 * kq;
#pragma unroll
        for (int t = 0; t < T; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) w2prev[t][r] = 0.0;
#pragma unroll
            for (int jj = 0; jj < 2; jj++) *(tile_d2v *)(wwr + (size_t)t * a.ldC * TL_LINE + 8 * jj) = tile_d2v{0.0, 0.0};   // Wo[0] = 0 (W[0] = 0 is already in the current half)
        }
        wrd += tl_grid; wwr += tl_grid;      // -> grid point 1
    }
    double *wb = wb_lds + (size_t)wave * (2 * T * 128);
    auto dma_w = [&]() {   // the chain's current W at the grid point wrd stands on -> wb (lane L's piece q at wb + (q*64 + L)*2)
        if constexpr (NOISE == 2 && BHIP_TILE_LDSDMA && (BHIP_TILE_EXP & 32) == 0) {
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wrd + (size_t)t * a.ldC * TL_LINE + 8 * jj),
                                                     (__attribute__((address_space(3))) void *)(wb + (t * 2 + jj) * 128), 16, 0, 0);
        }
    };
    dma_w();              // grid point 1 ...
    if constexpr (NOISE == 2 && BHIP_TILE_LDSDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... landed before step 0 reads it

    // The time loop, specialised on whether the path (and, fresh noise, the Wiener path) is stored: the number of vector-memory
    // operations per step is then STATIC.  With the wave-uniform `if (a.X)` inside the loop the compiler merged the two paths'
    // counters conservatively and waited with vmcnt(0) for the staged matrix loads -- i.e. for the step's own stores to be
    // acknowledged by the L2, a write round trip per step (round 3: that, not the stores themselves, was what the knock-out
    // of the stores saved).
    auto time_loop = [&](auto hasx_tag, auto hasw_tag) {
    constexpr bool HASX = decltype(hasx_tag)::value, HASW = decltype(hasw_tag)::value;
    for (int i = 0; i < nsteps; i++) {
        const int cur = (BHIP_TILE_EXP & 1) ? 0 : i & 1;
        const double *hm = hb + cur * STEP, *pm = hm + DD, *hnu = hm + 2 * DD, *qv = hnu + D;   // A_i | -Hm_i, P_i, b_i | Hm_i nu_i, q_i
        // stage step i+1's matrix: global -> registers now, registers -> LDS after the compute
        // (unconditional loads from clamped indices: a conditional `idx < STEP ? load : 0.0` made the compiler clear the staging
        // registers at the top of the step, and a register that a load of the previous step may still be writing cannot be cleared
        // without s_waitcnt vmcnt(0) -- issued right after the first load of the step: its whole round trip, every step, in every
        // instantiation.  The last step re-reads its own row and lands it in the buffer nobody reads.)
        constexpr bool MDMA = NOISE == 2 && BHIP_TILE_LDSDMA && BHIP_TILE_MDMA;   // the matrix travels by LDS-DMA too (below)
        double stage[(STEP + 255) / 256];
        if constexpr ((BHIP_TILE_EXP & 1) == 0 && !MDMA) {
            // the previous step's stores (issued between its matrix products, a product or more ago) are waited for HERE, where
            // nothing else is pending: from now on the compiler sees loads only.  (The builtin, not inline assembly: its wait-count
            // pass reads s_waitcnt instructions, not asm text.)
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt and lgkmcnt untouched (gfx9 encoding)
            const double *src = a.steps + (size_t)min(i + 1, nsteps - 1) * STEP;
#pragma unroll
            for (int c = 0; c < (STEP + 255) / 256; c++) stage[c] = src[min(tid + 256 * c, STEP - 1)];
        }
        // The step's vector-memory schedule (round 3).  The compiler's wait-count pass treats the vmcnt counter as out of order as
        // soon as stores (or an LDS-DMA) are pending, so ANY wait for a load result is then vmcnt(0): a wait for everything in flight.
        //   * The step's stores are not issued in one burst in front of the matrix products but one by one between them (store_op /
        //     hook_at below): a store that has to wait for room in the memory pipeline waits behind a running MFMA chain.
        //   * Chains (NOISE == 2): the next matrix row AND the chain's W[i+2] travel global -> LDS by DMA, issued at the top of the step,
        //     no staging registers, nothing in the step waits for them; ONE vmcnt(0) -- DMAs and stores, all at least 34 MFMAs old --
        //     at the barrier that ends the step.
        //   * The other instantiations stage the matrix row through registers: loads at the top (after a vmcnt(0) for the previous
        //     step's stores, which are a product or more old), landed in LDS after the noise has been drawn.
        // d = 32, 65 536 paths x 1000 steps, same box: proposals 14.1 -> 13.55 ms (0.63 of the fp64 matrix peak), chains 17.4 -> 16.3 (0.52).
        auto land_stage = [&]() {
            if constexpr ((BHIP_TILE_EXP & 1) == 0 && !MDMA) {
#pragma unroll
                for (int c = 0; c < (STEP + 255) / 256; c++) {
                    const int idx = tid + 256 * c;
                    if (idx < STEP) {
                        hb[(cur ^ 1) * STEP + idx] = stage[c];
                    }
                }
            }
        };
        // (dt, sqrt(dt)) ride along with the staged matrix: a vector load here would be waited for with vmcnt(0) -- the
        // compiler's rule while an LDS-DMA is in flight -- i.e. for the chain-state DMA issued just above, a full memory
        // latency per step; a scalar load would turn every lgkmcnt(N) of the LDS -> MFMA pipeline into lgkmcnt(0)
        // (the instantiations without a DMA keep the plain load: measured 2 % faster there than the LDS read)
        const double dt = (NOISE == 2 && BHIP_TILE_LDSDMA) ? hm[2 * DD + 2 * D] : a.hdr[2 * i], rdt = (NOISE == 2 && BHIP_TILE_LDSDMA) ? hm[2 * DD + 2 * D + 1] : a.hdr[2 * i + 1];
        // (c0_i is read from the step's LDS row in every instantiation: it is needed after the products, when the row has long landed)

        auto issue_dmas = [&]() {
            if constexpr (NOISE == 2 && BHIP_TILE_LDSDMA) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the DMA may overwrite the pieces
                wrd += tl_grid;
                if (i + 1 < nsteps) {
                    if constexpr (MDMA) {
                        // step i+1's matrix row (Hm, nu, dt, sqrt(dt): STEP/2 16-byte pieces) global -> hb[cur ^ 1] without staging
                        // registers: nothing in the step has to wait for it (a register landing is waited for by the compiler with
                        // vmcnt(0) -- after an LDS-DMA it treats the counter as out of order -- which also drained the DMA of
                        // W[i+2] 1500 cycles after its issue: less than an HBM round trip under this kernel's load).  Both DMAs now
                        // have the whole step; they are waited for once, with the step's stores, at the barrier that ends it.
                        constexpr int NP = STEP / 2;
                        static_assert(STEP % 2 == 0, "16-byte pieces");
                        const double *src = a.steps + (size_t)(i + 1) * STEP;
                        double *dst = hb + (cur ^ 1) * STEP;
#pragma unroll
                        for (int c = 0; c < (NP + 255) / 256; c++) {
                            const int q0 = c * 256 + wave * 64;   // wave-uniform
                            if (q0 + lane < NP)
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 2 * (q0 + lane)),
                                                                 (__attribute__((address_space(3))) void *)(dst + 2 * q0), 16, 0, 0);
                        }
                    }
                    dma_w();
                }
            }
        };
        // ---- the Wiener increment tile
        double dw[T][4];
        if constexpr (NOISE == 0) {
            const double *q = winp;
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double wn = ok(t, r) ? *q : 0.0;
                    q += rsWin;
                    dw[t][r] = wn - wprev[t][r];
                    wprev[t][r] = wn;
                }
            winp += (size_t)dtr * a.ldWin;
            land_stage();
        } else if constexpr (NOISE == 3) {
            // stand-alone llikelihood(LeftRule(), X, Po): x_i comes from the stored path, nothing is propagated
            const double *q = winp;
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    x[t][r] = ok(t, r) ? *q : 0.0;
                    q += rsWin;
                    dw[t][r] = 0.0;
                }
            winp += (size_t)dtr * a.ldWin;
            land_stage();
        } else {
            double wcur[NOISE == 2 ? T : 1][4];
            if constexpr (NOISE == 2 && BHIP_TILE_LDSDMA) {
                // the chain's current W[i+1] landed in LDS during the previous step; read it, then start the DMA of W[i+2]
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) {
                        const tile_d2v v = *(const tile_d2v *)(wb + ((t * 2 + jj) * 64 + lane) * 2);
                        wcur[t][2 * jj] = v.x; wcur[t][2 * jj + 1] = v.y;
                    }
                if constexpr (!BHIP_TILE_DMA_LATE) issue_dmas();
            } else if constexpr (NOISE == 2) {   // the chain's current W[i+1]: issued first, consumed after the normals
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) {
                        const tile_d2v v = (BHIP_TILE_EXP & 8) ? tile_d2v{1e-3 * (double)lane, 2e-3} : tile_ld((const tile_d2v *)(wrd + (size_t)t * a.ldC * TL_LINE + 8 * jj));
                        wcur[t][2 * jj] = v.x; wcur[t][2 * jj + 1] = v.y;
                    }
            }
            // Row `row` of the path takes normal n = i*dtr + row = number n & 3 of Philox call n >> 2 (bhip_rng.h: four normals per
            // call).  Per pass t the path's rows 16t .. 16t+15 are the normals nb .. nb+15, nb = i*dtr + 16t: four calls, one per
            // lane kq of the path's column, whose normals land in the column's 16-entry exchange row in LDS -- lane kq writes
            // entries 4kq .. 4kq+3 and reads back entries 4r + kq, the rows it holds: a 4 x 4 transpose among the four lanes.
            // PAD (any dtr, odd ones too): nb need not be a multiple of 4; the window then straddles five calls and lane 0
            // draws the fifth.
            double mine[4 * T];   // statically indexed
            double *zb = zb_lds + (size_t)wave * TILE_ZB + j * 18;
            // (generic in the table accessor whose TYPE carries the noise specification, bhip_rng.h: one wave-uniform branch per step)
            auto draw_rows = [&](const auto &tb) {
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const uint32_t nb = (uint32_t)i * (uint32_t)dtr + 16u * (uint32_t)t;
                    double z[4];
                    if constexpr ((BHIP_TILE_EXP & 2) != 0) { z[0] = 1e-3 * (double)(lane + t); z[1] = -z[0]; z[2] = 0.5 * z[0]; z[3] = -z[2]; }
                    else normal_quad(tb, a.k0, a.k1, path, a.iter, (nb >> 2) + (uint32_t)kq + (a.blk0 >> 1), z[0], z[1], z[2], z[3]);
                    if constexpr (!PAD) {
                        *(tile_d2v *)(zb + 4 * kq) = tile_d2v{z[0], z[1]};
                        *(tile_d2v *)(zb + 4 * kq + 2) = tile_d2v{z[2], z[3]};
                    } else {
                        const int off = (int)(nb & 3u);
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int pos = 4 * kq + u - off;
                            if (pos >= 0) zb[pos] = z[u];          // (pos <= 15 always)
                        }
                        if (off != 0 && kq == 0) {                 // wave-uniform `off`: the window's last entries come from a fifth call
                            normal_quad(tb, a.k0, a.k1, path, a.iter, (nb >> 2) + 4u + (a.blk0 >> 1), z[0], z[1], z[2], z[3]);
#pragma unroll
                            for (int u = 0; u < 3; u++)
                                if (u < off) zb[16 + u - off] = z[u];
                        }
                    }
                    __builtin_amdgcn_wave_barrier();   // one wave's LDS operations execute in order: the writes precede the reads
#pragma unroll
                    for (int r = 0; r < 4; r++) mine[4 * t + r] = zb[4 * r + kq];
                    __builtin_amdgcn_wave_barrier();   // ... and the reads precede the next pass's writes
                }
            };
            if (a.noise_spec == 2) draw_rows(FullRes<TabLDS>(rtab));
            else if (a.noise_spec == 3) draw_rows(rtab);
            else draw_rows(IcdfLDSHot(rtab_lds, &cold));
            if constexpr (BHIP_TILE_DMA_LATE) issue_dmas();   // the reads of wb are long done; the DMAs still have the products' time
            land_stage();
            double *qo = wop;
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if constexpr (NOISE == 2) {
                        // sample!(W2); Wo = rho*W + sqrt(1-rho^2)*W2   partialbridge_fitzhugh.jl:145-147
                        const double wc = wcur[t][r];
                        const double w2 = w2prev[t][r] + rdt * mine[4 * t + r];
                        const double wo = a.rho * wc + a.srho * w2;
                        dw[t][r] = wo - wprev[t][r];
                        w2prev[t][r] = w2;
                        wprev[t][r] = wo;
                        if constexpr ((BHIP_TILE_EXP & 16) == 0 && !BHIP_TILE_SPREAD) {   // (zero-padded rows carry exact zeros)
                            if ((r & 1) == 1) tile_st((tile_d2v *)(wwr + (size_t)t * a.ldC * TL_LINE + 8 * (r >> 1)), tile_d2v{wprev[t][r - 1], wo});
                        }
                    } else {
                        const double wn = wprev[t][r] + rdt * mine[4 * t + r];   // sample!: W[i+1] = W[i] + sqrt(dt)*xi
                        dw[t][r] = wn - wprev[t][r];
                        wprev[t][r] = wn;
                    }
                }
            if constexpr (NOISE == 2) { wst = wwr; if (!BHIP_TILE_LDSDMA) wrd += tl_grid; wwr += tl_grid; }
            if constexpr (NOISE == 1 && HASW) {
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) { if (ok(t, r)) *qo = wprev[t][r]; qo += rsWo; }
                wop += (size_t)dtr * a.ldWout * a.wstride;
            }
        }

        // ---- X[i] = x (before the update, src/euler.jl:263)
        // BHIP_TILE_SPREAD: a step's stores (the proposal's W lines, X[i]) are not issued in one burst ahead of the matrix products but
        // in groups of 2-4 between them: a wave whose store has to wait for room in the memory pipeline then waits behind a running
        // MFMA chain, not in front of it (the values stay in their registers until the update at the end of the step).
        double *const xst = xp;
        auto store_w = [&](int t) {
            if constexpr (NOISE == 2 && BHIP_TILE_SPREAD == 1 && (BHIP_TILE_EXP & 16) == 0) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++) tile_st((tile_d2v *)(wst + (size_t)t * a.ldC * TL_LINE + 8 * jj), tile_d2v{wprev[t][2 * jj], wprev[t][2 * jj + 1]});
            }
        };
        auto store_x = [&](int t) {
            if constexpr (NOISE != 3 && HASX && BHIP_TILE_SPREAD == 1) {
                double *q = xst + (size_t)(4 * t) * rsX;
#pragma unroll
                for (int r = 0; r < 4; r++) { if (ok(t, r)) tile_st(q, x[t][r]); q += rsX; }
            }
        };
        auto spread = [&](int slot) {   // slot 0: before the first product, k: after the k-th
            if constexpr (BHIP_TILE_SPREAD == 1) {
                __builtin_amdgcn_sched_barrier(BHIP_TILE_SB_MASK);
                if (slot < T) store_w(slot);
                else if (slot - T < T) store_x(slot - T);
                __builtin_amdgcn_sched_barrier(BHIP_TILE_SB_MASK);
            }
        };
        // BHIP_TILE_SPREAD == 2: one store after every BHIP_TILE_SPREAD_EVERY-th group of T MFMAs (T * 64 cycles of matrix pipe)
        auto store_op = [&](int k) {   // op 0 .. 2T-1: the W pieces; 2T .. 6T-1: the X rows
            if (k < 2 * T) {
                if constexpr (NOISE == 2 && (BHIP_TILE_EXP & 16) == 0) {
                    const int t = k >> 1, jj = k & 1;
                    tile_st((tile_d2v *)(wst + (size_t)t * a.ldC * TL_LINE + 8 * jj), tile_d2v{wprev[t][2 * jj], wprev[t][2 * jj + 1]});
                }
            } else if (k < 6 * T) {
                if constexpr (NOISE != 3 && HASX) {
                    const int e = k - 2 * T, t = e >> 2, r = e & 3;
                    if (ok(t, r)) tile_st(xst + (size_t)e * rsX, x[t][r]);
                }
            }
        };
        static_assert(BHIP_TILE_SPREAD != 2 || 6 * T * BHIP_TILE_SPREAD_EVERY <= 3 * 4 * T, "every store of the step needs its slot among the three products that always run");
        auto hook_at = [&](int mv) {
            return [&, mv](int ks) {
                if constexpr (BHIP_TILE_SPREAD == 2) {
                    const int slot = mv * 4 * T + ks;
                    if (slot % BHIP_TILE_SPREAD_EVERY == 0) {
                        __builtin_amdgcn_sched_barrier(BHIP_TILE_SB_MASK);
                        store_op(slot / BHIP_TILE_SPREAD_EVERY);
                        __builtin_amdgcn_sched_barrier(BHIP_TILE_SB_MASK);
                    }
                }
            };
        };
        if constexpr (NOISE != 3 && HASX) {
            if constexpr (!BHIP_TILE_SPREAD) {
                double *q = xp;
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) { if (ok(t, r)) tile_st(q, x[t][r]); q += rsX; }
            }
            xp += (size_t)dtr * a.ldX;
        }

        // the four products (head of the file); mv = 0 .. 3 numbers them for the store slots
        auto init = [&](double4v (&acc)[T], const double *vec) {
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[t][r] = vec[16 * t + 4 * r + kq];
        };
        double4v rr[T], db[UD::ON ? T : 1], xn[T];
        double bu[UD::ON ? T : 1][4];
        double part = 0.0;
        if constexpr (!UD::ON) {
            init(rr, hnu);
            spread(0);
            tile_mv_acc<T, decltype(hook_at(0)), PAD>(hm, x, rr, lane, hook_at(0), nks);        // y = b_i + A_i x
            spread(1);
        } else {
            init(rr, hnu);
            spread(0);
            tile_mv_acc<T, decltype(hook_at(0)), PAD>(hm, x, rr, lane, hook_at(0), nks);        // r = hnu_i - Hm_i x
            spread(1);
            const double *Dmf = TDA ? hm + 2 * DD + 2 * D + 4 : Dmf_c, *cvec = TDA ? hm + 3 * DD + 2 * D + 4 : cvec_c;   // -B~_i, c_i of THIS step's row
            init(db, cvec);
            tile_mv_acc<T, TileNoHook, PAD>(Dmf, x, db, lane, TileNoHook(), nks);                 // bT - bA = c + (0 - B~) x  (+ b below)
            // gather the path's state (its components sit in 4 lanes x 8 registers) and evaluate b_k for this lane's rows
            double *xv = xs_lds + (size_t)(wave * 16 + j) * D;
            __builtin_amdgcn_wave_barrier();   // the previous step's reads of xv are done
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) xv[16 * t + 4 * r + kq] = x[t][r];
            __builtin_amdgcn_wave_barrier();   // one wave's LDS operations execute in order: the writes above precede the reads below
            const double ti = a.tt[i];
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * t + 4 * r + kq;
                    bu[t][r] = (!PAD || row < dtr) ? UD::bk(row, ti, xv, a.upar) : 0.0;
                    db[t][r] += bu[t][r];
                }
        }
        if constexpr (NOISE != 3) {
            init(xn, qv);
            spread(2);
            tile_mv_acc<T, decltype(hook_at(0)), PAD>(pm, x, xn, lane, hook_at(1), nks);    // q_i + P_i x
            spread(3);
            tile_mv_acc<T, decltype(hook_at(0)), PAD>(Sf, dw, xn, lane, hook_at(2), nks);   //   + sigma dW
        }

        // ---- llikelihood: som += dot(b - b~, r)*dt, reduced over the path's 4 row groups
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if constexpr (UD::ON) part += db[t][r] * rr[t][r];
                else part += x[t][r] * rr[t][r];                       // x . (b_i + A_i x)
            }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if constexpr (!UD::ON) part += hm[2 * DD + 2 * D + 2];        // + c0_i
        if (i < nll) ll += part * dt;

        if constexpr (NOISE != 3) {
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if constexpr (UD::ON) x[t][r] = xn[t][r] + bu[t][r] * dt;
                    else x[t][r] = xn[t][r];
                }
        }

        // LDS-only barrier: this wave's reads of hb[cur] and its writes of hb[cur ^ 1] are done (lgkmcnt), the block meets; no
        // wait on vector memory (a __syncthreads would drain the stores just issued)
        if constexpr ((BHIP_TILE_EXP & 1) == 0) {
            if constexpr (MDMA) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // + this wave's DMAs (the others read its pieces) and stores
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    };
    if (NOISE != 3 && a.X) {
        if (NOISE == 1 && a.Wout) time_loop(TileTag<true>(), TileTag<true>());
        else time_loop(TileTag<true>(), TileTag<false>());
    } else {
        if (NOISE == 1 && a.Wout) time_loop(TileTag<false>(), TileTag<true>());
        else time_loop(TileTag<false>(), TileTag<false>());
    }

    if (a.use_vend) {
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) x[t][r] = vend[16 * t + 4 * r + kq];
    }
    if (NOISE != 3 && a.X) {   // xp has advanced to row N-1
        double *q = xp;
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) { if (ok(t, r)) *q = x[t][r]; q += rsX; }
    }
    if constexpr (NOISE == 2) {
        // if log(rand()) <= llo - ll: W <- Wo (parity flip), ll <- llo, acc += 1     partialbridge_fitzhugh.jl:160-167
        if (live && kq == 0 && !a.defer_accept) {
            const double u = accept_uniform(a.k0, a.k1, path, a.iter);
            if (det_log(u) <= ll - a.llcur[p]) {   // (the log's table from constant memory: once per chain and launch)
                a.cur[p] = (unsigned char)(cpar ^ 1);
                a.llcur[p] = ll;
                a.acc[p] += 1u;
            }
        }
    }
    if (a.ll && live && kq == 0) a.ll[p] = ll;
}

// ---- BHIP_RTC_END  (above: device code, also embedded for hipRTC user drifts at large d; below: host launch)

// dynamic LDS of k_tile<D, ., ., UD>: constants, two step buffers, generator tables (+ the gathered states for a user drift)
constexpr size_t tile_lds_bytes(int D, bool user, bool chains = true, bool tda = false)
{
    return sizeof(double) * (2 * D * D + 2 * D + 2 * tile_step_doubles(D, tda) + TILE_RNG_DOUBLES + (user ? 64 * D : 0) + 4 * TILE_ZB + (chains ? 4 * 2 * (D / 16) * 128 : 0));
}

template <int D, int NOISE, bool PAD = false>
hipError_t launch_tile(const TArgs &a, hipStream_t st)
{
    const size_t lds = tile_lds_bytes(D, false, NOISE == 2);
    // per device and cheap: set on every launch (a process may drive several devices)
    hipError_t e = hipFuncSetAttribute((const void *)k_tile<D, NOISE, PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const long grid = (a.P + 63) / 64;
    hipLaunchKernelGGL((k_tile<D, NOISE, PAD>), dim3((unsigned)grid), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int D, bool PAD>
hipError_t launch_tile_noise(const TArgs &a, int noise, hipStream_t st)
{
    return noise == 3 ? launch_tile<D, 3, PAD>(a, st) : noise == 2 ? launch_tile<D, 2, PAD>(a, st) : noise ? launch_tile<D, 1, PAD>(a, st) : launch_tile<D, 0, PAD>(a, st);
}

}  // namespace bhip
