// bhip_pc_kernel.h -- the fused path kernel with wave specialisation (d <= 3, noise dimension m' <= 3).
//
// Same arithmetic as k_paths<.., NOISE_FRESH, ..> / k_chain_lines (it calls the same path_step), different division of
// labour.  In those kernels one lane does everything for its path: the Philox / Box-Muller normal (state-independent,
// about half of the instructions) and the Euler recurrence (sequential in time).  With one path per lane the named
// small ensembles starve the chip -- 65 536 paths are one wave per SIMD, 32 768 chains leave half of the SIMDs idle -- and
// a lone wave cannot hide the latency of its own dependent chain.  Here a 128-thread workgroup owns 64 paths and splits
// the work BY WAVE:
//
//   wave 0, the PRODUCER: everything that does not depend on the state -- sample!(W2, Wiener())  (src/wiener.jl:24-58),
//           the pCN mix Wo = rho*W + sqrt(1-rho^2)*W2 (partialbridge_fitzhugh.jl:147) and the whole chain-state traffic
//           (the 128-byte lines of bhip_chain_kernel.h, read, mixed in an LDS tile, written to the other parity half).
//           Eight independent Philox blocks per 16-value chunk: instruction-level parallelism instead of a dependent chain.
//           The generator's table lives in LDS (IcdfLDS under the default noise specification v4, TabLDS under v3 / v2).
//   wave 1, the CONSUMER: solve!(Euler(), Xo, x0, Wo, Po) + llikelihood (src/euler.jl:247-268, src/partialbridge.jl:67-77
//           ...) exactly as path_step<.., NOISE_EXT, ..> does them, with the driving Wiener values read from the
//           producer's LDS tile instead of HBM; stores Xo; Metropolis-Hastings accept at the end.
//
// Hand-over: two 64 x 17 double tiles (a chunk = 16 values = 16/m' grid points per chain); the producer fills tile t&1
// with chunk t while the consumer works on chunk t-1 in the other tile; one workgroup barrier per chunk.  Twice the
// waves for the same paths (every SIMD busy at 32 768 chains, two waves per SIMD at 65 536), each with half the
// instructions, its own register budget (the generator's constants in the producer, the coefficient rows in the
// consumer: no scalar-register spills) -- and bit-identical results (tests/test_gpu_pc.py: == the monolithic kernels).
//
// RLDS (small ensembles, NPAIR > 1: LDS and registers to spare): the coefficient rows of a chunk travel through LDS
// as well.  A consumer wave that is alone on its SIMD otherwise exposes one scalar-load round trip to L2 per time step
// (the 128 KB of rows do not fit the scalar cache, and nothing else is there to issue meanwhile): measured ~640 cycles
// per step for ~50 VALU instructions.  The producer fetches the next chunk's rows with two vector loads per lane a
// whole chunk ahead and the consumer reads each row with broadcast ds_reads (~100 cycles, no scalar traffic).
#pragma once
#include "bhip_chain_kernel.h"

namespace bhip {

enum { NOISE_FRESH_PC = 6, NOISE_PCN_LINES_PC = 7 };

#ifndef PC_QUAD_UNROLL
#define PC_QUAD_UNROLL 1    // Philox calls (two Box-Muller pairs each) in flight per producer lane: instruction-level parallelism vs registers
#endif
#ifndef PC_CONS_PRIO
#define PC_CONS_PRIO 3
#endif
#ifndef PC_CONS_UNROLL_SMALL
#define PC_CONS_UNROLL_SMALL 8   // steps of the consumer's interior loop per iteration in the small-ensemble workgroups (a consumer alone on
#endif                           // its SIMD: the LDS reads of eight steps batched ahead of their use; 2 at the 128-register cap of the large ones).
                                 // Same-box A/B 2 -> 8: FHN pCN 32 768 0.225 -> 0.209 ms, FHN fresh 65 536 0.319 -> 0.298, C2 0.219 -> 0.212
#ifndef PC_WPE
#define PC_WPE 4            // minimum waves per SIMD the register allocation must allow (8 workgroups per CU by LDS)
#endif
#if defined(PC_STAMP) && PC_STAMP >= 2
#define PC_CONS_UNROLL 1
#else
#define PC_CONS_UNROLL (RLDS ? PC_CONS_UNROLL_SMALL : 2)
#endif
// DRAWER waves per pair (round 6).  Fresh proposals of a small ensemble (RLDS workgroups: one wave per role and SIMD) are bound by the
// PRODUCER wave -- its stream of ~20 vector instructions per normal is mostly serial inside one wave, and one wave per SIMD cannot fill
// the issue port (profiles/r4_c2_budget.txt: producer 383 cycles of work per step, consumer 217, the launch 397).  The draws of a chunk
// are independent of each other once the Wiener cumulation leaves the producer: the tile carries the INCREMENTS rdt*z, the consumer
// adds them up (W[j] = W[j-1] + rdt*z: the very operation, the very bits, src/wiener.jl:55) and stores W where it is kept.  So the
// Philox calls of a chunk are dealt to PC_NDRAW producer waves of the pair (calls [dr*NQ/NDRAW, (dr+1)*NQ/NDRAW) to drawer dr; the
// drawer of the last call carries its last m' normals into the next chunk), every one on its own SIMD slot.  pCN chains keep one
// producer (its cumulation of W2 and the line moves are sequential in the chunk).
#ifndef PC_NDRAW
#define PC_NDRAW 2
#endif
#ifndef PC_NDRAW_D1
#define PC_NDRAW_D1 PC_NDRAW   // drawers per pair for state dimension 1 (measurement hook: 4 = one Philox call per drawer and chunk, five waves per SIMD at <= 96 registers)
#endif
constexpr int pc_ndraw(int mode, bool rlds, int d) { return (mode == NOISE_FRESH_PC && rlds) ? (d == 1 ? PC_NDRAW_D1 : PC_NDRAW) : 1; }
constexpr int pc_threads(int mode, bool rlds, int npair, int d) { return 64 * npair * (pc_ndraw(mode, rlds, d) + 1); }
template <int V> struct PcInt { static constexpr int value = V; };
constexpr int PC_TILE = 64 * LINE_ROW;                          // doubles per hand-over tile
// LDS of a workgroup: the generator's table of the launch's noise specification (v4: 10 240 bytes, v3 / v2: 2 576), then per pair the two
// hand-over tiles (17 408 bytes) and, with RLDS, two chunks of coefficient rows.  Large ensembles want 8 pairs per CU: one pair per
// workgroup under v3 / v2 (19 984 bytes), FOUR pairs sharing one table under v4 (79 872 bytes, two workgroups per CU).
constexpr int pc_tab_doubles(int noise_spec) { return noise_spec == 2 || noise_spec == 3 ? RNG_TAB_DOUBLES : ICDF_TAB_DOUBLES; }
constexpr size_t pc_lds_bytes(int noise_spec, int npair, int crow /* doubles of coefficient rows per chunk, 0 without RLDS */)
{
    return sizeof(double) * (pc_tab_doubles(noise_spec) + npair * (2 * PC_TILE + 2 * crow));
}
// time-blocked path stores (KArgs::Xtb): a consumer lane collects sixteen grid points of its chain in LDS -- [k][16] + 1 double of padding
// per lane: conflict-free 8-byte accesses -- and the wave writes them out as whole 128-byte lines, eight lanes per line
constexpr int pc_xs_row(int d) { return 16 * d + 1; }
#ifndef PC_TBX_UNROLL_PPR
#define PC_TBX_UNROLL_PPR 2   // steps per iteration of the interior loop in the copy that stores time-blocked paths, per-chain guide rows
#endif
struct TabTrue { static constexpr bool value = true; };
struct TabFalse { static constexpr bool value = false; };
constexpr size_t pc_xs_bytes(int d, int npair) { return sizeof(double) * 64 * pc_xs_row(d) * npair; }
typedef const __attribute__((address_space(3))) double *ldsrow_t;

// The hand-over barrier of a chunk: the wave's LDS operations are done, the workgroup meets.  NOT __syncthreads: its
// workgroup fence is `s_waitcnt vmcnt(0)` on gfx950 -- a wait for every store the wave has issued (the consumer's path stores, the
// producer's line stores) to be acknowledged by the L2, once per chunk, which the waves of a small ensemble (one or two per
// SIMD) cannot hide.  Nothing but LDS is handed over between the waves of a pair.
__device__ __forceinline__ void pc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef PC_STAMP
// level 1: two stamps per chunk and wave -- cycles working, cycles waiting at the hand-over barrier
struct PcStamp {
    unsigned long long t0, work = 0, bar = 0, tot0;
    __device__ __forceinline__ static unsigned long long now() { __builtin_amdgcn_sched_barrier(0); const unsigned long long t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); return t; }
    __device__ __forceinline__ void begin() { tot0 = t0 = now(); }
    __device__ __forceinline__ void before_barrier() { const unsigned long long t = now(); work += t - t0; t0 = t; }
    __device__ __forceinline__ void after_barrier() { const unsigned long long t = now(); bar += t - t0; t0 = t; }
    __device__ __forceinline__ void write(unsigned long long *buf, int slot, int role, const unsigned long long *ph) const
    {
        if ((threadIdx.x & 63) != 0 || !buf) return;
        unsigned long long *o = buf + (size_t)slot * 12;
        o[0] = 1 + role; o[1] = now() - tot0; o[2] = work; o[3] = bar;
        for (int k = 0; k < 6; k++) o[4 + k] = ph ? ph[k] : 0ull;
    }
};
#define PC_ST(x) x
#else
#define PC_ST(x)
#endif

// KArgs::rdtp (device): rdtp[j] = sqrt(tt[j] - tt[j-1]) for 1 <= j <= N-1, rdtp[0] = 0, zero padded to a multiple of 16:
// the Wiener increment INTO grid point j, so that a chunk reads 16/m' consecutive, aligned values and grid point 0
// (W[0] = 0 + 0*z = 0) needs no special case.

template <class M, int GK, int MO, int MODE, int FL, int NPAIR, bool PPR = false /* per-chain guide rows (bhip_guide_kernel.h) */,
          bool RLDS = (NPAIR > 1) /* coefficient rows through LDS (small ensembles); false at NPAIR = 4: the large-ensemble workgroup of v4 */>
__global__ __launch_bounds__(pc_threads(MODE, RLDS, NPAIR, M::D) <= 1024 ? pc_threads(MODE, RLDS, NPAIR, M::D) : 1024, (RLDS || PPR) ? (pc_ndraw(MODE, RLDS, M::D) + 1) : PC_WPE) void k_pc(const KArgs a)
{
    constexpr int D = M::D, MP = M::MP;
    constexpr int NDRAW = pc_ndraw(MODE, RLDS, M::D), NTHR = pc_threads(MODE, RLDS, NPAIR, M::D);
    static_assert(MP >= 1 && MP <= 3, "a chunk holds 16/m' grid points (m' = 3: lines padded to 4 components)");
    static_assert(MODE == NOISE_FRESH_PC || MODE == NOISE_PCN_LINES_PC, "producer/consumer kernel: fresh proposals or pCN on the line layout");
    constexpr bool PCN = MODE == NOISE_PCN_LINES_PC;
    constexpr int MPP = line_mpp(MP);        // components per grid point inside a line / tile row
    constexpr int SPC = LINE_DOUBLES / MPP;  // grid points per chunk
    constexpr int NV = SPC * MP;             // values (normals) per chunk: 16, 16, 12
    constexpr int NQ = NV / 4;               // Philox calls per chunk (four normals each, bhip_rng.h): 4, 4, 3
    using RL = RowLayout<GK, D, MO, is_constdiff<M>::value>;
    // Workgroup = NPAIR producer/consumer pairs.  Small ensembles run 2 or 4 pairs per workgroup (and then RLDS): a
    // workgroup's waves are dealt to the four SIMDs of its CU in turn, so 4 waves sit on 4 different SIMDs and the 8 waves
    // of a 4-pair workgroup put one producer and one consumer on every SIMD, whereas the waves of independent 128-thread
    // workgroups may share a SIMD while another one idles (measured at 32 768 chains: producer-only 0.21 ms,
    // consumer-only 0.21 ms, both 0.31 ms in 128-thread workgroups, 0.25 ms in 256-thread ones).
    static_assert(RLDS ? NPAIR > 1 : true, "RLDS workgroups hold 2 or 4 pairs");
    extern __shared__ __attribute__((aligned(16))) double pc_lds_all[];
    double *tab = pc_lds_all;                                           // [pc_tab_doubles(noise_spec)], shared by the pairs
    const int tabd = pc_tab_doubles(a.noise_spec);
    constexpr int CROW = SPC * RL::RS;                                  // doubles of coefficient rows per chunk
    const int wave = threadIdx.x >> 6, pair = wave % NPAIR, role = wave / NPAIR;   // waves 0..NDRAW*NPAIR-1 produce (role = drawer index), the last NPAIR consume
    double *pc_lds = pc_lds_all + tabd + pair * (2 * PC_TILE + (RLDS ? 2 * CROW : 0));   // [2][PC_TILE] then (RLDS) [2][CROW]
    double *crow = pc_lds + 2 * PC_TILE;
    // (time-blocked path stores: the consumers' staging rows lie behind the pairs' tiles -- allocated by the launch only when a.Xtb is set)
    double *xs_all = pc_lds_all + tabd + NPAIR * (2 * PC_TILE + (RLDS ? 2 * CROW : 0));
    if (a.noise_spec == 2 || a.noise_spec == 3) TabLDS::load(tab, threadIdx.x, NTHR);
    else IcdfLDS::load(tab, threadIdx.x, NTHR);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    long c0 = ((long)blockIdx.x * NPAIR + pair) * 64;
    // a second pair without chains (odd number of 64-chain groups) re-runs the last group -- identical values to
    // identical addresses -- so that it takes part in every barrier; it commits nothing (no lane is live)
    const bool dup = c0 >= a.P;
    if (dup) c0 = (a.P - 1) / 64 * 64;
    const bool live = !dup && c0 + lane < a.P;
    // lanes beyond the ensemble replicate path P-1 exactly (same tile row, same stream): their stores write identical
    // values to identical addresses and need no execution mask
    const int row = c0 + lane < a.P ? lane : (int)(a.P - 1 - c0);
    const long p = c0 + row;
    const int N = a.N, nsteps = N - 1;
    const int nch = (N + SPC - 1) / SPC;
    const uint32_t path = a.path0 + (uint32_t)p;

    auto producer = [&](auto drc) {
        // ------------------------------------------------------------------ producer (drawer DR of NDRAW)
        constexpr int DR = decltype(drc)::value;
        constexpr int CLO = DR * NQ / NDRAW, CHI = (DR + 1) * NQ / NDRAW;   // this drawer's Philox calls of a chunk
        constexpr bool LASTDR = CHI == NQ;                                   // ... the last call's tail is carried into the next chunk
        const TabLDS rtab(tab);
        const cptr_t rdtp = (cptr_t)(uintptr_t)a.rdtp;
        double w2prev[MP];
#pragma unroll
        for (int c = 0; c < MP; c++) w2prev[c] = 0.0;
        // grid point j, component c takes normal (j-1)*m' + c: a chunk's first m' values are the LAST m' normals of the Philox
        // call drawn at the end of the previous chunk (chunk 0: grid point 0, multiplied by rdtp[0] = 0)
        double carry[MP];
#pragma unroll
        for (int c = 0; c < MP; c++) carry[c] = 0.0;
        // cooperative line moves (pCN): instruction q moves the lines of chains c0 + 8q + lane/8; lane%8 selects 16 bytes
        const int sub = lane >> 3, part = 2 * (lane & 7);
        // addresses: a wave-uniform base per chunk (the 64 chains' 256-byte pairs of chunk k are contiguous) + a small per-lane
        // offset that never changes -- line (8q + sub) of the group, parity half, 16-byte piece -- whose half bit is flipped for
        // the store: 8 registers instead of 2 x 8 64-bit addresses (which spilled at the 128-register cap)
        // (the eight offsets differ by q * 256 doubles and by the parity bit of chain 8q + sub: ONE register of offset and ONE of
        // parity bits, the rest is two integer operations per access -- eight registers of offsets were what spilled)
        const int vbase = (int)line_index(0, 0, sub, nch, a.ldC) + part;
        uint32_t hbits = 0u;
        auto voff = [&](int q) { return vbase + q * (8 * 2 * LINE_DOUBLES) + (int)((hbits >> q) & 1u) * LINE_DOUBLES; };
        d2v stage[8];
        auto chunk_base = [&](int k) { return a.Wc + line_index(0, k, c0, nch, a.ldC); };
        auto fetch = [&](int k) {
            const double *kb = chunk_base(k);
#pragma unroll
            for (int q = 0; q < 8; q++) stage[q] = ld_stream((const d2v *)(kb + voff(q)));
        };
        if constexpr (PCN) {
#pragma unroll
            for (int q = 0; q < 8; q++)   // cur[] is allocated (and zeroed) up to ld
                hbits |= (uint32_t)(a.cur[c0 + 8 * q + sub] & 1) << q;
            fetch(0);
        }
        // RLDS: chunk k needs the rows of steps i = SPC*k - 1 .. SPC*k + SPC - 2, contiguous in memory; every lane moves
        // 16-byte pieces (RS is even and the rows are 256-byte aligned), clamped into the array at both ends
        constexpr int NRV = (CROW + 127) / 128;
        d2v rstage[NRV];
        const long rlast = (long)nsteps * RL::RS - 2;
        auto fetch_rows = [&](int k) {
#pragma unroll
            for (int q = 0; q < NRV; q++) {
                long e = ((long)SPC * k - 1) * RL::RS + (q * 64 + lane) * 2;
                e = e < 0 ? 0 : (e > rlast ? rlast : e);
                rstage[q] = *(const d2v *)(a.rows + e);
            }
        };
        if constexpr (RLDS && DR == 0) fetch_rows(0);   // (the coefficient rows travel with drawer 0)
        PC_ST(PcStamp ps; ps.begin();)

        for (int k = 0; k <= nch; k++) {
            if (k < nch) {
                double *tile = pc_lds + (k & 1) * PC_TILE;
                double *mine = tile + row * LINE_ROW;
                if constexpr (RLDS && DR == 0) {
#pragma unroll
                    for (int q = 0; q < NRV; q++) {
                        const int e = (q * 64 + lane) * 2;
                        if (e < CROW) *(d2v *)(crow + (k & 1) * CROW + e) = rstage[q];
                    }
                    if (k + 1 < nch) fetch_rows(k + 1);
                }
                if constexpr (PCN) {
#pragma unroll
                    for (int q = 0; q < 8; q++) {   // staged lines -> tile
                        double *d = tile + (8 * q + sub) * LINE_ROW + part;
                        d[0] = stage[q].x; d[1] = stage[q].y;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (k + 1 < nch) fetch(k + 1);   // in flight while this chunk is mixed
                }
                // value v = s*MP + c of the chunk (grid point j = SPC*k + s, component c; tile position s*MPP + c) takes normal
                // n = (j-1)*MP + c = NV*k - MP + v.  The chunk draws the Philox calls NQ*k .. NQ*k + NQ-1 = normals NV*k .. NV*k + NV-1:
                // values 0 .. MP-1 come from the carry, normal t of the chunk's own draws is value t + MP, and the last MP of them
                // are carried into the next chunk.
                // RLDS (a producer that is alone on its SIMD): the chunk's 16/m' scales in one scalar load up front and the
                // calls fully unrolled -- otherwise every value waits for its own scalar load
                double rd[SPC];
                if constexpr (RLDS) {
#pragma unroll
                    for (int s = 0; s < SPC; s++) rd[s] = rdtp[(size_t)k * SPC + s];
                }
                auto value = [&](int v, double zz) {
                    const int s = v / MP, c = v % MP, j = k * SPC + s, pos = s * MPP + c;
                    double rdt;
                    if constexpr (RLDS) rdt = rd[s];
                    else rdt = rdtp[j];                                           // wave-uniform: scalar load
                    if constexpr (!PCN) {
                        (void)j;
                        mine[pos] = rdt * zz;                                   // the INCREMENT rootdt*randn; the consumer adds it up   src/wiener.jl:55
                    } else {
                        const double wc = mine[pos];
                        const double w2 = w2prev[c] + rdt * zz;
                        w2prev[c] = w2;
                        mine[pos] = a.rho * wc + a.srho * w2;                    // Wo = rho*W + sqrt(1-rho^2)*W2
                    }
                };
                const uint32_t q0 = (uint32_t)(NQ * k) + (a.blk0 >> 1);   // blk0 counts pairs (two per call) and is even
                // the chunk's draws, generic in the table accessor whose TYPE carries the noise specification (bhip_rng.h): both copies
                // are straight-line code, one wave-uniform branch per chunk picks one
                auto draws = [&](const auto &tb) {
                    auto draw = [&](int i, double (&z)[4]) {
#ifdef PC_KNOCKOUT_NOISE   /* measurement only: what the consumer alone costs */
                        z[0] = 0.25; z[1] = -0.5; z[2] = 0.125; z[3] = -0.75;
#else
                        normal_quad(tb, a.k0, a.k1, path, a.iter, q0 + (uint32_t)i, z[0], z[1], z[2], z[3]);
#endif
                    };
                    if constexpr (LASTDR) {
#pragma unroll
                        for (int c = 0; c < MP; c++) value(c, carry[c]);
                    }
                    // every call but the chunk's last: all four normals are values of this chunk (component index static for m' = 1, 2 at
                    // any unrolling; m' = 3 -- three calls -- is unrolled fully)
                    constexpr int CEND = LASTDR ? NQ - 1 : CHI;
                    constexpr int QU = (RLDS || MP == 3) ? NQ : PC_QUAD_UNROLL;
#pragma unroll QU
                    for (int i = CLO; i < CEND; i++) {
                        double z[4];
                        draw(i, z);
#pragma unroll
                        for (int u = 0; u < 4; u++) value(4 * i + u + MP, z[u]);
                    }
                    if constexpr (LASTDR) {
                        double z[4];
                        draw(NQ - 1, z);
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (u < 4 - MP) value(NV - 4 + u + MP, z[u]);
                            else carry[u - (4 - MP)] = z[u];
                        }
                    }
                };
                if (a.noise_spec == 2) draws(FullRes<TabLDS>(rtab));
                else if (a.noise_spec == 3) draws(rtab);
                else draws(IcdfLDS(tab));
                if constexpr (PCN) {
                    __builtin_amdgcn_wave_barrier();
                    double *kb = chunk_base(k);
#pragma unroll
                    for (int q = 0; q < 8; q++) {   // tile -> the lines of the other halves
                        const double *d = tile + (8 * q + sub) * LINE_ROW + part;
                        st_stream((d2v *)(kb + (voff(q) ^ LINE_DOUBLES)), d2v{d[0], d[1]});
                    }
                }
            }
            PC_ST(ps.before_barrier();)
            pc_barrier();   // chunk k is complete; the consumer has finished chunk k-1 (the tile written next)
            PC_ST(ps.after_barrier();)
        }
        PC_ST(ps.write(a.stamp, blockIdx.x * ((NDRAW + 1) * NPAIR) + wave, 0, nullptr);)
    };
    if (role < NDRAW) {   // (wave-uniform)
        if constexpr (NDRAW == 1) producer(PcInt<0>());
        else if constexpr (NDRAW == 2) { if (role == 0) producer(PcInt<0>()); else producer(PcInt<1>()); }
        else {
            static_assert(NDRAW == 4, "PC_NDRAW: 1, 2 or 4 drawer waves per pair");
            if (role == 0) producer(PcInt<0>()); else if (role == 1) producer(PcInt<1>()); else if (role == 2) producer(PcInt<2>()); else producer(PcInt<3>());
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer
    // The consumer's step is a chain of dependent fp64 operations; the producer's instructions are independent filler.  With the
    // consumer at the higher issue priority its next instruction goes as soon as its operands are there and the producer takes
    // the cycles in between (C2, one pair per SIMD: 0.239 -> 0.222 ms per launch in a same-box A/B; nothing where memory binds).
#ifdef PC_CONS_PRIO_ALL
    __builtin_amdgcn_s_setprio(PC_CONS_PRIO);
#else
    if constexpr (RLDS) __builtin_amdgcn_s_setprio(PC_CONS_PRIO);
#endif
    const M model(a.mpar);
    const int nll = N - 1 - a.skip;
    const cptr_t rows = (cptr_t)(uintptr_t)a.rows;
    LaneState<D, MP> st;
#pragma unroll
    for (int k = 0; k < D; k++) st.y[k] = a.x0_dev ? a.x0_dev[k * a.ldx0 + p] : a.x0[k];
    // time-blocked path stores (multi-segment chains): yy[i] = y is collected per lane in LDS; when a block of sixteen grid points is
    // complete the wave moves it out like the producer moves the W lines -- instruction q writes the lines of chains c0 + 8q + lane/8,
    // lane%8 selects 16 bytes: whole 128-byte lines per instruction -- into the half of each chain's pair that is NOT its current path
    // (or the buffer a.xsel names: four bits per chain)
    constexpr bool TBX = PCN && (FL & 1) == 0;
    constexpr int XSR = pc_xs_row(D);
    double *xs = nullptr, *xsw = nullptr;
    char *xlane = nullptr;
    uint32_t xhb = 0u;
    bool xtb = false;
    if constexpr (TBX) {
        xtb = a.Xtb != nullptr;
        if (xtb) {
            xsw = xs_all + pair * (64 * XSR);
            xs = xsw + lane * XSR;
            const int sub = lane >> 3, part = 2 * (lane & 7);
            xlane = (char *)(a.Xtb + ((size_t)c0 + sub) * 16 + part);
#pragma unroll
            for (int q = 0; q < 8; q++) {   // cur[] / xsel[] are allocated (and zeroed) up to ld; without xsel the proposal goes to the other half
                const long c = c0 + 8 * q + sub;
                xhb |= (uint32_t)(a.xsel ? (a.xsel[c] & 15) : ((a.cur[c] & 1) ^ 1)) << (4 * q);
            }
        }
    }
    auto flush_x = [&](int blk) {
        __builtin_amdgcn_wave_barrier();
        const int sub = lane >> 3, part = 2 * (lane & 7);
#pragma unroll
        for (int k = 0; k < D; k++) {
            char *kb = xlane + ((size_t)blk * D + k) * (size_t)a.ldC * 128;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const double *src = xsw + (8 * q + sub) * XSR + k * 16 + part;
                st_stream((d2v *)(kb + (size_t)q * (8 * 128) + (size_t)((xhb >> (4 * q)) & 15u) * ((size_t)a.xtb_half * 8)), d2v{src[0], src[1]});
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    // (a block can only be complete at the FIRST step of a chunk -- i = SPC*kc + s - 1 with SPC a divisor of 16 --: the interior loop's
    // unrolled body holds the line moves once, not once per step; with them in every step the small-ensemble instantiations went from
    // ~145 to 256 registers and spilled)
    auto stage_x = [&](auto on, int i, bool first_of_chunk) {   // yy[i] = y, BEFORE the update (src/euler.jl:263)
        if constexpr (decltype(on)::value) {
#pragma unroll
            for (int k = 0; k < D; k++) xs[k * 16 + (i & 15)] = st.y[k];
            if (first_of_chunk && (i & 15) == 15) flush_x(i >> 4);
        }
    };
    // The per-chain start is the consumer's only vector load.  Left to itself the compiler defers the wait for it to the first use
    // of the state it can find on every path -- the join block at the end of a chunk, INSIDE the chunk loop: an s_waitcnt vmcnt(0)
    // per chunk, i.e. a wait for the acknowledgement of every path store the wave has issued, once per chunk.  Waiting here,
    // once, (the builtin: the wait-count pass reads instructions, not asm text) leaves the loop without one.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), gfx9 encoding; expcnt / lgkmcnt untouched
    st.ll = 0.0; st.zq[0] = st.zq[1] = st.zq[2] = 0.0;
#pragma unroll
    for (int k = 0; k < MP; k++) { st.wprev[k] = 0.0; st.w2prev[k] = 0.0; }
    double *xout = nullptr;
    long ldx = 0;
    if constexpr ((FL & 1) != 0) {
        if constexpr (PCN) { xout = a.Xo; ldx = a.ldC; }   // wave-uniform base; the lane's chain index rides as a 32-bit offset
        else {   // (a pair's 64 paths share a part -- xpart is a multiple of 64 --: the base stays wave-uniform)
            const int j = a.xpart ? __builtin_amdgcn_readfirstlane((int)(c0 / a.xpart)) : 0;
            xout = (j == 0 ? a.X : j == 1 ? a.Xp1 : a.Xp2) - (long)j * a.xpart; ldx = a.ldX;
        }
    }
    constexpr int CFL = FL & ~2;   // (path_step stores no W under NOISE_EXT: the consumer's loop below does)
    // fresh proposals: the tile holds the increments; W[j] = W[j-1] + rdt*z is added up here and, where W is kept, stored from here
    double *wout = nullptr;
    long ldwo = 0;
    if constexpr (!PCN && (FL & 2) != 0) { wout = a.Wout + (size_t)p * a.wstride; ldwo = a.ldWout * a.wstride; }
    auto tile_w = [&](const double *mine, int s, int j, double (&wn)[MP]) {
#pragma unroll
        for (int c = 0; c < MP; c++) {
            const double v = mine[s * MPP + c];
            if constexpr (PCN) wn[c] = v;
            else {
                wn[c] = st.wprev[c] + v;                                    // yy[i] = yy[i-1] + rootdt*randn   src/wiener.jl:55
                if constexpr ((FL & 2) != 0) st_stream(&wout[((size_t)j * MP + c) * ldwo], wn[c]);
            }
        }
    };
    // per-chain guides: the chain's compact row (Hd, V, linearisation datum) of step i+1 is fetched while step i is computed
    // and expanded into the row entries right before its step (expand_pp_row); the three time entries stay shared
    constexpr int NPP = PPR ? pp_row_len<D>() : 1, NE = PPR ? RL::LEN - 3 : 1;
    RegRow<NPP> rcur, rnxt;
    auto fetch_row = [&](int i, RegRow<NPP> &r) {
        if constexpr (PPR) {
            const double *src = a.prows + (size_t)min(i, N - 2) * NPP * a.ldr + p;
#pragma unroll
            for (int q = 0; q < NPP; q++) r.v[q] = src[(size_t)q * a.ldr];
        }
    };
    auto ppr_step = [&](int i, const double *wn) {
        if constexpr (PPR) {
            ExpRow<NE> x;
            x.sh = rows + (size_t)i * RL::RS;
            expand_pp_row<M>(model, a.lna, rcur.v, x.e);
            fetch_row(i + 1, rnxt);
            path_step<M, GK, MO, NOISE_EXT, CFL, ExpRow<NE>>(model, a, x, i, nll, path, wn, nullptr, 0, xout, ldx, st, TabConst(), (uint32_t)p * 8u);
            rcur = rnxt;
        }
    };
    fetch_row(0, rcur);
    PC_ST(PcStamp ps; ps.begin(); for (int q = 0; q < 6; q++) st.tacc[q] = 0ull; st.tlast = ps.t0;)
    // the chunk loop, held twice by the instantiations that can store time-blocked paths: with the wave-uniform `if (xtb)` inside the
    // unrolled step the small-ensemble kernels WITHOUT such stores went from ~145 registers to 256 and spilled up to 1 KB
    auto chunk_loop = [&](auto tbx_on) {
    for (int k = 0; k <= nch; k++) {
        if (k > 0) {
            const int kc = k - 1;
            const double *mine = pc_lds + (kc & 1) * PC_TILE + row * LINE_ROW;
            const ldsrow_t lrows = (ldsrow_t)(__attribute__((address_space(3))) double *)(crow + (kc & 1) * CROW);   // row of step j0 + s - 1 at s*RS
            const int j0 = kc * SPC;
            if (kc > 0 && j0 + SPC <= N) {
                // interior chunk: SPC steps  i = j - 1
                constexpr int UNR = (decltype(tbx_on)::value && PPR) ? PC_TBX_UNROLL_PPR : PC_CONS_UNROLL;
#pragma unroll UNR
                for (int s = 0; s < SPC; s++) {
                    double wn[MP];
                    tile_w(mine, s, j0 + s, wn);
                    const int i = j0 + s - 1;
                    stage_x(tbx_on, i, s == 0);
#ifdef PC_KNOCKOUT_STEP   /* measurement only: what the producer alone costs */
                    st.ll += wn[0];
#else
                    if constexpr (PPR) ppr_step(i, wn);
                    else if constexpr (RLDS) path_step<M, GK, MO, NOISE_EXT, CFL>(model, a, lrows + s * RL::RS, i, nll, path, wn, nullptr, 0, xout, ldx, st, TabConst(), (uint32_t)p * 8u);
                    else path_step<M, GK, MO, NOISE_EXT, CFL>(model, a, rows + (size_t)i * RL::RS, i, nll, path, wn, nullptr, 0, xout, ldx, st, TabConst(), (uint32_t)p * 8u);
#endif
                }
            } else {
                // first chunk (grid point 0 has no step) and the ragged last chunk
#pragma unroll 1
                for (int s = 0; s < SPC; s++) {
                    const int i = j0 + s - 1;
                    if (i >= nsteps) continue;
                    double wn[MP];
                    tile_w(mine, s, j0 + s, wn);                            // (grid point 0: W[0] = 0 + 0*z, stored where W is kept; no step)
                    if (i < 0) continue;
                    stage_x(tbx_on, i, true);
                    if constexpr (PPR) ppr_step(i, wn);
                    else if constexpr (RLDS) path_step<M, GK, MO, NOISE_EXT, CFL>(model, a, lrows + s * RL::RS, i, nll, path, wn, nullptr, 0, xout, ldx, st, TabConst(), (uint32_t)p * 8u);
                    else path_step<M, GK, MO, NOISE_EXT, CFL>(model, a, rows + (size_t)i * RL::RS, i, nll, path, wn, nullptr, 0, xout, ldx, st, TabConst(), (uint32_t)p * 8u);
                }
            }
        }
        PC_ST(ps.before_barrier();)
        pc_barrier();
        PC_ST(ps.after_barrier(); st.tlast = ps.t0;)
    }
    };
    if constexpr (TBX) {
        if (xtb) chunk_loop(TabTrue());
        else chunk_loop(TabFalse());
    } else chunk_loop(TabFalse());
    PC_ST(ps.write(a.stamp, blockIdx.x * ((NDRAW + 1) * NPAIR) + wave, 1, st.tacc);)

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
        for (int k = 0; k < D; k++) st_stream(&xout[((size_t)(N - 1) * D + k) * ldx + p], st.y[k]);
    }
    if constexpr (TBX) {
        if (xtb) {   // the end point closes the last block (its tail beyond N - 1 is padding)
#pragma unroll
            for (int k = 0; k < D; k++) xs[k * 16 + ((N - 1) & 15)] = st.y[k];
            flush_x((N - 1) >> 4);
            if (a.xend) {
#pragma unroll
                for (int k = 0; k < D; k++) a.xend[(size_t)k * a.ldC + p] = st.y[k];
            }
        }
    }
    if constexpr (PCN) {
        // if log(rand()) <= llo - ll: W <- Wo (parity flip), ll <- llo, acc += 1      partialbridge_fitzhugh.jl:160-167
        if (live) {
            if (!a.defer_accept) {
                const double u = accept_uniform(a.k0, a.k1, path, a.iter);
                if (det_log(u) <= st.ll - a.llcur[p]) {   // (the log's table from constant memory: once per chain and launch)
                    a.cur[p] = (unsigned char)(a.cur[p] ^ 1);
                    a.llcur[p] = st.ll;
                    a.acc[p] += 1u;
                }
            }
            if (a.ll) a.ll[p] = st.ll;
        }
    } else {
        if (live && a.ll) a.ll[p] = st.ll;
    }
}

// ---- BHIP_RTC_END  (above: device code, also embedded for hipRTC user models; below: host launch)

// Small ensembles: up to 512 groups of 64 chains (32 768 chains) as 2-pair workgroups -- one wave per SIMD on up to 256
// CUs --, up to 1024 groups (65 536) as 4-pair workgroups -- one producer and one consumer per SIMD; both RLDS (LDS and
// 256 registers per lane are free at two waves per SIMD).  Larger ensembles: 8 pairs per CU at the 128-register cap -- under the noise
// specification v4 as two 4-pair workgroups (the 10-KB table shared by four pairs; one pair per workgroup would fit five per CU), under
// v3 / v2 as eight 128-thread workgroups.
#ifndef PC_MAX_GROUPS_2PAIR
#define PC_MAX_GROUPS_2PAIR 512
#endif
#ifndef PC_MAX_GROUPS_4PAIR
#define PC_MAX_GROUPS_4PAIR 1024
#endif
// pairs per workgroup of the LARGE ensembles under v4 (measurement hook: BHIP_PC_LARGE_NPAIR = 1 | 4 in the environment, read once)
inline int pc_large_npair()
{
    static const int v = []() { const char *e = getenv("BHIP_PC_LARGE_NPAIR"); return e && e[0] == '1' ? 1 : 4; }();
    return v;
}
// the LDS a time-blocked launch (a.Xtb) needs on top must still fit: 160 KB per workgroup
constexpr size_t PC_LDS_MAX = 160 * 1024;
template <class M, int GK, int MO, int MODE, int FL, int NPAIR, bool PPR = false, bool RLDS = (NPAIR > 1)>
hipError_t launch_pc_n(const KArgs &a, hipStream_t st, long groups)
{
    using RL = RowLayout<GK, M::D, MO, is_constdiff<M>::value>;
    const size_t lds = pc_lds_bytes(a.noise_spec, NPAIR, RLDS ? (LINE_DOUBLES / line_mpp(M::MP)) * RL::RS : 0) + (a.Xtb ? pc_xs_bytes(M::D, NPAIR) : 0);
    if (lds > PC_LDS_MAX) return hipErrorInvalidValue;
    // more than the default 64 KB of dynamic LDS: opt in -- per device and cheap, so on every launch (a process may drive several devices)
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute((const void *)k_pc<M, GK, MO, MODE, FL, NPAIR, PPR, RLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_pc<M, GK, MO, MODE, FL, NPAIR, PPR, RLDS>), dim3((unsigned)((groups + NPAIR - 1) / NPAIR)), dim3(pc_threads(MODE, RLDS, NPAIR, M::D)), lds, st, a);
    return hipGetLastError();
}
// pairs per workgroup of a launch (also what the hipRTC route of do_launch follows): 2 / 4 (RLDS) for small ensembles -- fewer when
// the time-blocked staging rows of a d = 3 guide would not fit next to four pairs' tiles --, then 1 or 4 without RLDS
inline int pc_choose_npair_rt(const KArgs &a, long groups, int d, int rs, int mp, bool *rlds, int mode)
{
    const int crow = (LINE_DOUBLES / line_mpp(mp)) * rs;
    // (a workgroup holds 1024 threads at most: with four drawers per pair two pairs are the largest RLDS workgroup)
    auto fits = [&](int np, bool rl) { return pc_lds_bytes(a.noise_spec, np, rl ? crow : 0) + (a.Xtb ? pc_xs_bytes(d, np) : 0) <= PC_LDS_MAX && pc_threads(mode, rl, np, d) <= 1024; };
    *rlds = true;
    if (groups <= PC_MAX_GROUPS_2PAIR && fits(2, true)) return 2;
    if (groups <= PC_MAX_GROUPS_4PAIR && fits(4, true)) return 4;
    if (groups <= PC_MAX_GROUPS_4PAIR && fits(2, true)) return 2;
    *rlds = false;
    const bool v4 = !(a.noise_spec == 2 || a.noise_spec == 3);
    if (v4 && pc_large_npair() == 4 && fits(4, false)) return 4;
    return 1;
}
template <class M, int GK, int MO, int MODE, int FL, bool PPR>
hipError_t launch_pc(const KArgs &a, hipStream_t st)
{
    using RL = RowLayout<GK, M::D, MO, is_constdiff<M>::value>;
    const long groups = (a.P + 63) / 64;
    bool rlds;
    const int np = pc_choose_npair_rt(a, groups, M::D, RL::RS, M::MP, &rlds, MODE);
    if (np == 2) return launch_pc_n<M, GK, MO, MODE, FL, 2, PPR>(a, st, groups);
    if (np == 4 && rlds) return launch_pc_n<M, GK, MO, MODE, FL, 4, PPR>(a, st, groups);
    if (np == 4) return launch_pc_n<M, GK, MO, MODE, FL, 4, PPR, false>(a, st, groups);
    return launch_pc_n<M, GK, MO, MODE, FL, 1, PPR>(a, st, groups);
}

}  // namespace bhip
