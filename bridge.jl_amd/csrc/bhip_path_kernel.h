// bhip_path_kernel.h -- the fused path-per-lane kernel (d <= 3).
//
// One lane owns one path / chain and walks the whole time grid sequentially with its state in
// registers.  Per step it fuses what the reference does in four separate passes over N-length arrays:
//
//   LOOP A  sample!(W2, Wiener())                      src/wiener.jl:24-58
//   LOOP P  Wo = rho*W + sqrt(1-rho^2)*W2              partialbridge_fitzhugh.jl:147
//   LOOP B  solve!(Euler(), Xo, x0, Wo, Po)            src/euler.jl:247-268 (guided), :135-152 (plain)
//   LOOP C  llikelihood(LeftRule(), Xo, Po; skip)      src/partialbridge.jl:67-77, src/guip.jl:429-438,
//                                                      src/partialbridgenuH.jl:171-181, src/partialbridgen!.jl:81-97
//   + MH accept                                        partialbridge_fitzhugh.jl:160-167
//
// Fusing is exact: llikelihood re-evaluates r and both drifts at the stored X[i], which is the
// register state the Euler step uses.
//
// Memory: the path-independent per-step coefficients ("rows": t, dt, sqrt(dt), B~, beta~, guide)
// are read through the scalar unit (wave-uniform address, constant address space -> s_load into
// SGPRs, served by the scalar cache / L2).  The only per-lane HBM traffic is the SoA ensemble:
// 8 B per component per step, 64 consecutive lanes = 512 contiguous bytes (coalesced).
#pragma once
#include "bhip_models.h"
#include "bhip_rng.h"
#include "bhip_smallmat.h"

namespace bhip {

enum { NOISE_EXT = 0, NOISE_FRESH = 1, NOISE_PCN = 2, NOISE_LLONLY = 3, NOISE_INNOV = 4,
       NOISE_PCN_LINES = 5 /* pCN step on the line layout, k_chain_lines in bhip_chain_kernel.h (m' <= 3) */ };

template <class T> struct bhip_unref { typedef T type; };
template <class T> struct bhip_unref<T &> { typedef T type; };
template <class T> struct bhip_unref<const T &> { typedef T type; };
template <class T> struct bhip_unref<const T> { typedef T type; };
template <bool C, class A, class B> struct bhip_cond { typedef A type; };   // (no <type_traits> under hipRTC)
template <class A, class B> struct bhip_cond<false, A, B> { typedef B type; };

// does the model functor provide inv(sigma)*v (square, invertible diffusion coefficient)?
template <class...> using bhip_void_t = void;
template <class M, class = void> struct has_sinv { static constexpr bool value = false; };
template <class M> struct has_sinv<M, bhip_void_t<decltype(&M::sinv_mul)>> { static constexpr bool value = true; };

// constdiff(P) (src/types.jl:36): every built-in process has a constant sigma; a hipRTC user process with a
// state-dependent sigma(t,x,P) declares `static constexpr bool STATE_SIGMA = true` and provides amat().
template <class M, class = void> struct is_constdiff { static constexpr bool value = true; };
template <class M> struct is_constdiff<M, bhip_void_t<decltype(M::STATE_SIGMA)>> { static constexpr bool value = !M::STATE_SIGMA; };

// GUIDE_QF (internal, not part of the ABI; round 5; d <= 3 under BHIP_OPT_FUSED_ARITHMETIC since round 6): the REGROUPED step of the LinPro family at 4 <= d <= 12.  Target and auxiliary are
// affine and everything but x and dW is path-independent, so the host (finish_guide, regroup_step) turns the five d x d products of a
// step into three whose accumulators start from path-independent vectors, exactly as for the tile kernel (bhip_tile_kernel.h):
//     dot(b - b~, r) = c0_i + x . (bv_i + A_i x)        x_{i+1} = q_i + P_i x + sigma dW
// row: t, dt, sqrt(dt), A_i (d*d), bv_i (d), P_i (d*d), q_i (d), c0_i.  Tolerance parity, like every path at d > 3.
#define BHIP_GUIDE_QF 5
constexpr int BHIP_MAXD_LANE = 12;   // one path per lane up to here (4..8 since round 3, 9..12 since round 4); beyond: the MFMA tile kernel

// targets with an affine drift and a constant sigma, for which the regrouped rows exist at d <= 3 (MLinPro<1..3>: `AFFINE_TARGET`)
template <class M, class = void> struct is_affine_target { static constexpr bool value = false; };
template <class M> struct is_affine_target<M, bhip_void_t<decltype(M::AFFINE_TARGET)>> { static constexpr bool value = M::AFFINE_TARGET; };

// processes whose parameter block is re-opened from device memory at every step (MLinPro<4..8, bhip_cptr_t>)
template <class M, class = void> struct is_streamed { static constexpr bool value = false; };
template <class M> struct is_streamed<M, bhip_void_t<decltype(M::STREAMED)>> { static constexpr bool value = M::STREAMED; };

struct KArgs {
    const double *rows;   // [N-1][rs] packed per-step coefficients (device)
    const double *rdtp;   // rdtp[j] = sqrt(tt[j] - tt[j-1]) (0 for j = 0), zero padded to a multiple of 16 (bhip_pc_kernel.h)
    int rs, N, skip;
    int use_vend;         // GuidedBridge endpoint rule: X[N-1] = V[N-1]
    long P;               // paths
    const double *x0_dev; // optional per-path starts [D][ldx0]
    long ldx0;
    const double *Win;    // EXT: driving W [N][MP][ldWin];  LLONLY: X to evaluate
    long ldWin;
    double *Wout;         // FRESH: optional W store
    long ldWout;
    double *X;            // optional X store [N][D][ldX]
    long ldX;
    // X in PARTS (bhip_sample_solve_parts): paths [j*xpart, (j+1)*xpart) go to part j -- X, Xp1, Xp2, each [N][D][ldX] in an allocation of
    // its own (different pieces of the device memory: one write stream per piece instead of one in all); xpart = 0: one buffer
    double *Xp1, *Xp2;
    long xpart;           // paths per part, a multiple of 64
    // the INPUT ensemble (Win: the driving W of NOISE_EXT, the stored X of NOISE_LLONLY / NOISE_INNOV) in parts, likewise (bhip_solve_parts,
    // bhip_llikelihood_parts): paths [j*wpart, (j+1)*wpart) are read from part j -- Win, Winp1, Winp2, each [N][.][ldWin]; 0: one buffer
    const double *Winp1, *Winp2;
    long wpart;
    double *ll;           // optional per-path log-likelihood
    // pCN chain state (NOISE_PCN).  W lives in 16-byte SLOTS: Wc[((i*MP + k)*ldC + p)*2 + h], h = 0/1;
    // half cur[p] holds the chain's current W, the other half receives the proposal Wo, and an accept
    // flips cur[p] (the reference's W <-> Wo swap, test/partialbridgenuH.jl:186-187, without copies).
    // A lane always loads and stores its whole slot: 16 B per lane, 1 KiB contiguous per wave, no
    // partially written lines although neighbouring chains have different parities.
    // Xo is the proposal path buffer (plain SoA, every chain writes it every iteration).
    double *Wc;
    double *Xo;
    long ldC;
    int wstride;          // FRESH W store: 1 = plain SoA, 2 = half 0 of the chain slots (chain initialisation)
    unsigned char *cur;
    double *llcur;
    unsigned int *acc;
    double rho, srho;
    uint32_t k0, k1, iter, path0;
    uint32_t blk0;        // offset of the Philox block index (multi-segment chains: segment << 24; 0 otherwise)
    int defer_accept;     // pCN modes: do not decide -- only report llo in `ll` (joint accept over segments, bhip_segchains_*)
    int noise_spec;       // BHIP_OPT_NOISE_SPEC: 4 (and anything else) the default stream bhip-philox-v4, 3: bhip-philox-v3, 2: bhip-philox-v2 (bhip_rng.h)
    double x0[BHIP_MAXD_LANE];       // d <= 3 for every process; LinPro targets of dimension 4..12 run one path per lane too
    double vend[BHIP_MAXD_LANE];
    double mu_aux[BHIP_MAXD_LANE];
    double mpar[40];
    const double *mpar_dev;   // 4 <= d <= 8: the parameter block in device memory (MLinPro<D, bhip_cptr_t>)
    // per-chain coefficient rows (device-built guides, bhip_guide_kernel.h): prows[(i*PRL + q)*ldr + p] holds entry 3 + q of
    // chain p's row of step i (the three time entries stay in the shared rows); per-chain endpoint rule
    const double *prows;
    long ldr;
    int lna;                      // the per-chain rows carry LinearNoiseAppr data (slope) instead of the linearisation point
    const double *vend_pc;        // [D][ldr]
    const unsigned char *uv_pc;   // [ldr]
    // multi-segment chains on the line layout (bhip_segchains.inc; wave-specialised kernel, instantiation without the plain X store): the
    // paths in the TIME-BLOCKED layout -- sixteen grid points of ONE chain side by side, one 128-byte line, so that whoever reads the
    // current paths of chains with different parities fetches nothing else: Xtb[h*xtb_half + (((i >> 4)*D + k)*ldC + p)*16 + (i & 15)];
    // half cur[p] holds the chain's current path, half cur[p] ^ 1 receives the proposal (like W: accept = parity flip, no copy);
    // xsel (optional): the buffer h = xsel[p] that receives chain p's proposal, out of a ring of up to 16 (deferred mcnext!: the current
    // paths of the last few iterations stay where they are until the statistics pass has read them); xend [D][ldC]: the proposal's end
    // point, the start of the next segment
    double *Xtb;
    long xtb_half;
    double *xend;
    const unsigned char *xsel;
#ifdef PC_STAMP   /* measurement builds only (scripts/gpu_stamp_probe.py): per-wave cycle budget of the producer / consumer waves */
    unsigned long long *stamp;
#endif
};

// the X store's base for path p (KArgs::xpart): shifted so that the GLOBAL path index addresses the part the path belongs to
__device__ __forceinline__ double *x_store_base(const KArgs &a, long p)
{
    if (!a.xpart) return a.X;
    const int j = (int)(p / a.xpart);
    return (j == 0 ? a.X : j == 1 ? a.Xp1 : a.Xp2) - (long)j * a.xpart;
}

// ... and the input ensemble's base for path p (KArgs::wpart)
__device__ __forceinline__ const double *in_base(const KArgs &a, long p)
{
    if (!a.wpart) return a.Win;
    const int j = (int)(p / a.wpart);
    return (j == 0 ? a.Win : j == 1 ? a.Winp1 : a.Winp2) - (long)j * a.wpart;
}

typedef const __attribute__((address_space(4))) double *cptr_t;

// a coefficient row whose time entries are shared (scalar loads) and whose other entries belong to the lane's chain
struct PerPathRow {
    cptr_t sh;
    const double *pp;
    long ld;
    BHIP_DEV double operator[](int q) const { return q < 3 ? sh[q] : pp[(size_t)(q - 3) * ld]; }
};
// the chain's COMPACT entries (Hd, V, linearisation datum) in registers, fetched one step ahead by k_paths ...
template <int NPP>
struct RegRow {
    double v[NPP];
};
// ... and the row the step reads, expanded from them (expand_pp_row)
template <int NE>
struct ExpRow {
    cptr_t sh;
    double e[NE];
    BHIP_DEV double operator[](int q) const { return q < 3 ? sh[q] : e[q - 3]; }
};
typedef double d2v __attribute__((ext_vector_type(2)));   // one 16-byte chain slot

// Streaming accesses of the ensemble: every byte is written once / read once per launch and never
// re-used by this kernel, so they carry the non-temporal hint (measured on the bench workload:
// 2.52 -> 2.31 ms per MCMC iteration, 1.58 -> 1.50 ms per proposal sweep).
#ifndef BHIP_NT_LOADS
#define BHIP_NT_LOADS 1
#endif
#ifndef BHIP_NT_STORES
#define BHIP_NT_STORES 1
#endif
template <class T> BHIP_DEV void st_stream(T *p, T v)
{
    if constexpr (BHIP_NT_STORES) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <class T> BHIP_DEV T ld_stream(const T *p)
{
    if constexpr (BHIP_NT_LOADS) return __builtin_nontemporal_load(p);
    else return *p;
}

// CD = constant diffusivity.  With a state-dependent a(t,x) the PartialBridge row carries M (in place of
// the pre-multiplied (aL')M) and, for the extra log-likelihood terms of src/partialbridge.jl:79-84,
// H = L'ML and the auxiliary a~.
template <int GK, int D, int MO, bool CD = true>
struct RowLayout {
    static constexpr int T = 0, DT = 1, RDT = 2;
    static constexpr int B = 3;              // D*D col-major
    static constexpr int BETA = 3 + D * D;   // D
    static constexpr int G = 3 + D * D + D;  // guide part
    static constexpr int GLEN = GK == BHIP_GUIDE_HV ? (D == 1 ? 3 : D == 2 ? 8 : 14)   // Hd / cofactors, det, V, 1/divisor
                              : GK == BHIP_GUIDE_LMMU ? (MO * D + MO + 2 * D * MO + (CD ? 0 : 2 * D * D))
                              : GK == BHIP_GUIDE_NUH ? (D * D + D) : GK == BHIP_GUIDE_QF ? (D * D + D + 1) : 0;
    static constexpr int LM_H = G + MO * D + MO + 2 * D * MO;   // LMMU, !CD: H (D*D), then a~ (D*D)
    static constexpr int LEN = GK == BHIP_GUIDE_NONE ? 3 : G + GLEN;
    static constexpr int RS = (LEN + 1) & ~1;
};

// r((i,t),x,Po) and g = a*L'*M*q | a*r, from the packed row
template <class M, int GK, int MO, class GP = const double *>
BHIP_DEV void guide_terms(const M &model, double t, GP g, const double *x, double *r, double *gd)
{
    constexpr int D = M::D;
    if constexpr (GK == BHIP_GUIDE_HV) {
        // Hd[i] \ (V[i] - x)                                   src/guip.jl:192-193
        // the divisions by the row's Hd (d = 1) / det(Hd) (d = 2, 3) use the row's pre-computed reciprocal: three operations
        // each, the bits of `/` (sm_div_by, bhip_smallmat.h)
        if constexpr (D == 1) {
            r[0] = sm_div_by(g[1] - x[0], g[0], g[2]);
        } else if constexpr (D == 2) {
            const double w0 = g[5] - x[0], w1 = g[6] - x[1];
            r[0] = sm_div_by(g[3] * w0 - g[2] * w1, g[4], g[7]);
            r[1] = sm_div_by(g[0] * w1 - g[1] * w0, g[4], g[7]);
        } else {
            const double w0 = g[10] - x[0], w1 = g[11] - x[1], w2 = g[12] - x[2];
            r[0] = sm_div_by(g[0] * w0 + g[1] * w1 + g[2] * w2, g[9], g[13]);
            r[1] = sm_div_by(g[3] * w0 + g[4] * w1 + g[5] * w2, g[9], g[13]);
            r[2] = sm_div_by(g[6] * w0 + g[7] * w1 + g[8] * w2, g[9], g[13]);
        }
        model.amul(t, x, r, gd);
    } else if constexpr (GK == BHIP_GUIDE_LMMU) {
        // q = (v - mu[i]) - L[i]*x ; r = (L'M)q ; g = ((aL')M)q    src/partialbridge.jl:53-57
        double q[MO];
#pragma unroll
        for (int j = 0; j < MO; j++) {
            double s = g[j] * x[0];
#pragma unroll
            for (int k = 1; k < D; k++) s += g[j + MO * k] * x[k];
            q[j] = g[MO * D + j] - s;
        }
        const GP R = g + MO * D + MO, G = g + MO * D + MO + D * MO;
        if constexpr (is_constdiff<M>::value) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                double sr = R[i] * q[0];
#pragma unroll
                for (int j = 1; j < MO; j++) sr += R[i + D * j] * q[j];
                r[i] = sr;
                // a structurally zero row of sigma makes row i of a*L'*M an exact zero (+-0): not computed, not added
                if (M::noisy(i)) {   // folds after unrolling
                    double sg = G[i] * q[0];
#pragma unroll
                    for (int j = 1; j < MO; j++) sg += G[i + D * j] * q[j];
                    gd[i] = sg;
                } else gd[i] = 0.0;
            }
        } else {
            // a(t,x) depends on the state: ((a*L')*M)*q evaluated per step; the G slot holds M (MO x MO)
            double A[D * D], AL[D * MO], ALM[D * MO];
            model.amat(t, x, A);
#pragma unroll
            for (int i = 0; i < D; i++)
#pragma unroll
                for (int j = 0; j < MO; j++) {
                    double sa = A[i] * g[j];                                   // (a*L')[i][j] = sum_k a[i][k]*L[j][k]
#pragma unroll
                    for (int k = 1; k < D; k++) sa += A[i + D * k] * g[j + MO * k];
                    AL[i + D * j] = sa;
                }
#pragma unroll
            for (int i = 0; i < D; i++)
#pragma unroll
                for (int j = 0; j < MO; j++) {
                    double sa = AL[i] * G[MO * j];
#pragma unroll
                    for (int l = 1; l < MO; l++) sa += AL[i + D * l] * G[l + MO * j];
                    ALM[i + D * j] = sa;
                }
#pragma unroll
            for (int i = 0; i < D; i++) {
                double sr = R[i] * q[0], sg = ALM[i] * q[0];
#pragma unroll
                for (int j = 1; j < MO; j++) { sr += R[i + D * j] * q[j]; sg += ALM[i + D * j] * q[j]; }
                r[i] = sr; gd[i] = sg;
            }
        }
    } else if constexpr (GK == BHIP_GUIDE_NUH) {
        // r = H[i]*(nu[i] - x) ; g = a*r                            src/partialbridgenuH.jl:157-161
        double w[D];
#pragma unroll
        for (int k = 0; k < D; k++) w[k] = g[D * D + k] - x[k];
        if constexpr (is_streamed<M>::value) matvec_streamed<D, GP>(g, w, r);
        else {
#pragma unroll
            for (int i = 0; i < D; i++) {
                double s = g[i] * w[0];
#pragma unroll
                for (int j = 1; j < D; j++) s += g[i + D * j] * w[j];
                r[i] = s;
            }
        }
        model.amul(t, x, r, gd);
    }
}

#ifndef BHIP_KCH
#define BHIP_KCH 4
#endif
// depth of the prefetch window of the pCN kernel (a 16-byte slot per step in flight).  Measured on the bench
// workload: 1, 2, 3 and 4 steps ahead run within 2 % of each other; 2 keeps the kernel at 118 VGPRs, 3 and 4 sit
// on the 128-register cap of 4 waves per SIMD and spill a few loop-invariant values to scratch.
#ifndef BHIP_PATHS_BLOCK
#define BHIP_PATHS_BLOCK 256   // threads per workgroup of k_paths (<= 256: its launch bound)
#endif
#ifndef BHIP_PATHS_UNR_MULT
#define BHIP_PATHS_UNR_MULT 1   // (2: twice the steps per loop iteration in the noise-drawing d <= 3 kernels -- measured, no gain)
#endif
#ifndef BHIP_KCH_PCN
#define BHIP_KCH_PCN 2
#endif

// per-lane state carried through the time loop
template <int D, int MP>
struct LaneState {
    double y[D];
    double ll;
    double wprev[MP], w2prev[MP];
    double zq[3];   // normals 1..3 of the current Philox call (four normals per call, bhip_rng.h)
#ifdef PC_STAMP
    unsigned long long tacc[6], tlast;
#endif
};

#if defined(PC_STAMP) && PC_STAMP >= 2
// s_memtime stamps INSIDE the step (level 2): every mark closes a phase -- waits for the wave's LDS / scalar loads, reads the
// clock, adds the cycles since the previous mark to phase k.  Scheduling barriers on both sides keep the phases apart: this is a
// budget of the raw, un-overlapped costs (the production schedule batches the LDS reads of eight steps), not a timing of it.
template <class ST> BHIP_DEV void pc_mark(ST &st, int k)
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    st.tacc[k] += t - st.tlast;
    st.tlast = t;
    __builtin_amdgcn_sched_barrier(0);
}
#define PSTAMP(k) do { if constexpr (NOISE == NOISE_EXT) pc_mark(st, k); } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif

// One Euler step of one path, branch-free (a single basic block so that the scheduler can interleave
// the state-independent work -- Philox, Box-Muller, address arithmetic -- with the dependent chain).
//   win_k : EXT: W[i+1] ; PCN: current chain W[i+1] ; LLONLY: X[i]       (already in registers)
//   FL    : bit0 store X, bit1 store W, bit2 PartialBridge!-style log-likelihood (two dots)
//   RowPtr: where the step's coefficient row is read from -- cptr_t (constant address space: scalar loads into SGPRs)
//           or an LDS pointer (the consumer waves of bhip_pc_kernel.h at small ensembles: broadcast ds_reads)
//   Tab   : where the generator's table is read from and -- its TYPE -- under which noise specification (bhip_rng.h: IcdfLDS / IcdfConst
//           the default v4, TabLDS / TabConst v3, FullRes<..> v2); the modes that draw nothing never touch it
template <class M, int GK, int MO, int NOISE, int FL, class RowPtr = cptr_t, class Tab = TabConst>
BHIP_DEV void path_step(const M &model, const KArgs &a, RowPtr row, int i, int nll, uint32_t path, const double *win_k,
                        double *wout, long ldwo, double *xout, long ldx, LaneState<M::D, M::MP> &st, const Tab &tab = Tab(),
                        uint32_t xl = 0u /* the lane's BYTE offset when xout is the wave-uniform base: address = scalar base + 32-bit lane offset */,
                        int i4 = -1 /* i & 3 where the caller knows it statically (unrolled time loops); -1: taken from i */)
{
    constexpr int D = M::D, MP = M::MP;
    using RL = RowLayout<GK, D, MO, is_constdiff<M>::value>;
    // the whole coefficient row through the scalar unit, one wait
    // (STREAMED models, 4 <= d <= 8: only the time entries now; the guide and auxiliary parts are read phase by phase below,
    // each pinned behind the phase before it -- the whole row and the model's matrices do not fit the scalar registers together)
    constexpr bool STR = is_streamed<M>::value;
    double rw[RL::RS];
#pragma unroll
    for (int q = 0; q < (STR ? 3 : RL::LEN); q++) rw[q] = row[q];
    const double t = rw[RL::T], dt = rw[RL::DT];
    PSTAMP(0);   // phase 0: everything since the previous step's last mark + the row in registers (LDS / scalar-load latency)

    if constexpr (NOISE == NOISE_LLONLY) {
#pragma unroll
        for (int k = 0; k < D; k++) st.y[k] = win_k[k];
    }

    if constexpr (NOISE == NOISE_INNOV) {
        // innovations!(::EulerMaruyama, W, Y, P)  src/euler.jl:358-376 (the inverse map X -> W, square sigma):
        //   ww[i] = w;  w = w + inv(sigma)*(yy[i+1] - yy[i] - _b((i,t),yy[i],P)*dt)      win_k = yy[i+1]
        static_assert(D == MP, "innovations need a square, invertible sigma");
#pragma unroll
        for (int k = 0; k < D; k++) st_stream(&wout[((size_t)i * D + k) * ldwo], st.wprev[k]);
        double bI[D];
        model.b(t, st.y, bI);
        if constexpr (GK != BHIP_GUIDE_NONE) {
            double r[D], g[D];
            if constexpr (STR) {
                RowPtr rg = row;
                bhip_after(rg, bI[D - 1]);
                guide_terms<M, GK, MO, RowPtr>(model, t, rg + RL::G, st.y, r, g);
            } else guide_terms<M, GK, MO>(model, t, rw + RL::G, st.y, r, g);
#pragma unroll
            for (int k = 0; k < D; k++)
                if (M::noisy(k)) bI[k] = bI[k] + g[k];
        }
        double df[D], inc[D];
#pragma unroll
        for (int k = 0; k < D; k++) df[k] = win_k[k] - st.y[k] - bI[k] * dt;
        model.sinv_mul(df, inc);
#pragma unroll
        for (int k = 0; k < D; k++) { st.wprev[k] = st.wprev[k] + inc[k]; st.y[k] = win_k[k]; }
        return;
    }

    // ---- LOOP A / P: the Wiener increment of this step
    double dw[MP];
    if constexpr (NOISE == NOISE_EXT) {
#pragma unroll
        for (int k = 0; k < MP; k++) {
            const double wn = win_k[k];
            dw[k] = wn - st.wprev[k];
            st.wprev[k] = wn;
        }
    } else if constexpr (NOISE == NOISE_FRESH || NOISE == NOISE_PCN) {
        const double rdt = rw[RL::RDT];
#pragma unroll
        for (int k = 0; k < MP; k++) {
            // normal n = i*MP + k is number n & 3 of Philox call n >> 2: drawn with the call's first normal, then taken from the
            // lane state.  The callers unroll the time loop so that n & 3 is static (one basic block); in their ragged tails it is a
            // wave-uniform branch.
            const int n = i * MP + k, ph = ((i4 >= 0 ? i4 : (i & 3)) * MP + k) & 3;
            double z;
            if (ph == 0) normal_quad(tab, a.k0, a.k1, path, a.iter, (uint32_t)(n >> 2) + (a.blk0 >> 1), z, st.zq[0], st.zq[1], st.zq[2]);
            else z = ph == 1 ? st.zq[0] : ph == 2 ? st.zq[1] : st.zq[2];
            if constexpr (NOISE == NOISE_FRESH) {
                const double wn = st.wprev[k] + rdt * z;          // yy[i] = yy[i-1] + rootdt*randn
                dw[k] = wn - st.wprev[k];                          // ww[i+1] - ww[i]
                st.wprev[k] = wn;
                if constexpr ((FL & 2) != 0) st_stream(&wout[((size_t)(i + 1) * MP + k) * ldwo], wn);
            } else {
                const double wc = win_k[k];
                const double w2 = st.w2prev[k] + rdt * z;
                const double wo = a.rho * wc + a.srho * w2;        // Wo = rho*W + sqrt(1-rho^2)*W2
                dw[k] = wo - st.wprev[k];
                st.w2prev[k] = w2;
                st.wprev[k] = wo;                                  // the caller stores it into the chain slot
            }
        }
    }

    // ---- LOOP B: yy[i] = y (stored BEFORE the update, src/euler.jl:263)
    if constexpr (NOISE != NOISE_LLONLY && (FL & 1) != 0) {
#pragma unroll
        for (int k = 0; k < D; k++)   // (uniform row base) + (the lane's 32-bit BYTE offset): the store takes its base from scalar registers
            st_stream((double *)((char *)(xout + ((size_t)i * D + k) * ldx) + xl), st.y[k]);
    }
    PSTAMP(1);   // phase 1: dw + the issue of the X stores (back-pressure shows here)

    if constexpr (GK == BHIP_GUIDE_QF && !STR) {
        // the regrouped step at d <= 3 (round 6; BHIP_OPT_FUSED_ARITHMETIC only -- tolerance parity like every regrouped step): the whole
        // row is in registers, A_i, bv_i and c0_i carry dt already (finish_guide), every product is a fused multiply-add:
        //     ll += c0'_i + x . (bv'_i + A'_i x)          x_{i+1} = (q_i + sigma dW) + P_i x
        // What the bit-exact step makes the NEXT step wait for -- b, the guide solve, a r, b~, the dot product, the update: a chain of
        // ~15 dependent fp64 operations at d = 1 -- is here ONE dependent fused multiply-add per component (q_i + sigma dW does not
        // depend on the state, the log-likelihood hangs off the chain): a consumer wave that is alone on its SIMD stops being bound
        // by the latency of its own recurrence (C2: ~310 cycles per step for ~50 instructions, profiles/r5_c2_sq.txt).
        double part = rw[RL::G + D * D + D];                                             // c0'_i
#pragma unroll
        for (int q = 0; q < D; q++) {
            double s = rw[RL::BETA + q];                                                 // bv'_i
#pragma unroll
            for (int j = 0; j < D; j++) s = __builtin_fma(rw[RL::B + q + D * j], st.y[j], s);
            part = __builtin_fma(st.y[q], s, part);
        }
        const double lln = st.ll + part;
        st.ll = (i < nll) ? lln : st.ll;
        if constexpr (NOISE != NOISE_LLONLY) {
            double sw[D], xn[D];
            model.sdw(t, st.y, dw, sw);                                                  // sigma dW (constant sigma: off the state's chain)
#pragma unroll
            for (int q = 0; q < D; q++) {
                double s = rw[RL::G + D * D + q] + sw[q];                                // q_i + sigma dW
#pragma unroll
                for (int j = 0; j < D; j++) s = __builtin_fma(rw[RL::G + q + D * j], st.y[j], s);
                xn[q] = s;
            }
#pragma unroll
            for (int q = 0; q < D; q++) st.y[q] = xn[q];
        }
        PSTAMP(2); PSTAMP(3); PSTAMP(4);
        return;
    } else if constexpr (GK == BHIP_GUIDE_QF) {
        // the regrouped step (see BHIP_GUIDE_QF above): three products, accumulators started from the row's vectors
        static_assert(STR, "the streamed form of the regrouped rows (4 <= d <= 12)");
        double yv[D], xn[D];
        RowPtr ra = row;
        bhip_after(ra, st.y[D - 1]);
        matvec_streamed_init<D, RowPtr>(ra + RL::B, ra + RL::BETA, st.y, yv);          // bv_i + A_i x
        double part = st.y[0] * yv[0];
#pragma unroll
        for (int k = 1; k < D; k++) part = __builtin_fma(st.y[k], yv[k], part);
        RowPtr rc = row;
        bhip_after(rc, part);
        const double lln = st.ll + (part + rc[RL::G + D * D + D]) * dt;                 // + c0_i
        st.ll = (i < nll) ? lln : st.ll;
        if constexpr (NOISE != NOISE_LLONLY) {
            matvec_streamed_init<D, RowPtr>(rc + RL::G, rc + RL::G + D * D, st.y, xn);   // q_i + P_i x
            auto S = model.p + D * D + D;                                                // sigma of the LinPro block [B, mu, sigma, a, ...]
            matvec_streamed_acc<D>(S, dw, xn);                                           //   + sigma dW
#pragma unroll
            for (int k = 0; k < D; k++) st.y[k] = xn[k];
        }
        PSTAMP(2); PSTAMP(3); PSTAMP(4);
        return;
    }
    double bT[D];
    model.b(t, st.y, bT);
    if constexpr (GK != BHIP_GUIDE_NONE) {
        double r[D], g[D];
        if constexpr (STR) {
            RowPtr rg = row;
            bhip_after(rg, bT[D - 1]);
            guide_terms<M, GK, MO, RowPtr>(model, t, rg + RL::G, st.y, r, g);
        } else guide_terms<M, GK, MO>(model, t, rw + RL::G, st.y, r, g);
        // ---- LOOP C: som += dot(b - b~, r)*dt ;  b~ = B~(x - mu~) + beta~  (mu~ = 0 for the affine form
        // B~x + beta~, beta~ = 0 for the LinPro form B~(x - mu~): adding/subtracting 0.0 is exact)
        double xm[D], bA[D];
#pragma unroll
        for (int k = 0; k < D; k++) xm[k] = st.y[k] - a.mu_aux[k];
        if constexpr (STR) {
            RowPtr rb = row;
            bhip_after(rb, g[D - 1]);
            matvec_streamed<D, RowPtr>(rb + RL::B, xm, bA);
            RowPtr rbe = row;
            bhip_after(rbe, bA[D - 1]);
#pragma unroll
            for (int q = 0; q < D; q++) bA[q] = bA[q] + rbe[RL::BETA + q];
        } else {
#pragma unroll
            for (int q = 0; q < D; q++) {
                double s = rw[RL::B + q] * xm[0];
#pragma unroll
                for (int j = 1; j < D; j++) s += rw[RL::B + q + D * j] * xm[j];
                bA[q] = s + rw[RL::BETA + q];
            }
        }
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if constexpr ((FL & 4) != 0) {
            s1 = bT[0] * r[0]; s2 = bA[0] * r[0];
#pragma unroll
            for (int k = 1; k < D; k++) { s1 += bT[k] * r[k]; s2 += bA[k] * r[k]; }
        } else {
            s0 = (bT[0] - bA[0]) * r[0];
#pragma unroll
            for (int k = 1; k < D; k++) s0 += (bT[k] - bA[k]) * r[k];
        }
        double lln;
        if constexpr ((FL & 4) != 0) lln = (st.ll + s1 * dt) - s2 * dt;   // PartialBridge!: src/partialbridgen!.jl:96-97
        else lln = st.ll + s0 * dt;                                        // src/partialbridge.jl:77
        if constexpr (!is_constdiff<M>::value && GK == BHIP_GUIDE_LMMU) {
            // A = a(t,x,P) - a~_i;  som -= 0.5*tr(A*H)*dt;  som += 0.5*(r'*A*r)*dt      src/partialbridge.jl:79-84
            double A[D * D];
            model.amat(t, st.y, A);
#pragma unroll
            for (int q = 0; q < D * D; q++) A[q] = A[q] - rw[RL::LM_H + D * D + q];
            double trAH = 0.0, quad = 0.0;
#pragma unroll
            for (int ii = 0; ii < D; ii++) {
                double e = A[ii] * rw[RL::LM_H + D * ii];                 // (A*H)[ii][ii] = sum_k A[ii][k]*H[k][ii]
#pragma unroll
                for (int k = 1; k < D; k++) e += A[ii + D * k] * rw[RL::LM_H + k + D * ii];
                trAH = ii == 0 ? e : trAH + e;
            }
#pragma unroll
            for (int jj = 0; jj < D; jj++) {
                double e = r[0] * A[D * jj];                              // (r'*A)[jj] = sum_i r[i]*A[i][jj]
#pragma unroll
                for (int ii = 1; ii < D; ii++) e += r[ii] * A[ii + D * jj];
                quad = jj == 0 ? e * r[0] : quad + e * r[jj];
            }
            lln = lln - (0.5 * trAH) * dt;
            lln = lln + (0.5 * quad) * dt;
        }
        st.ll = (i < nll) ? lln : st.ll;                                   // skip: only i < N-1-skip contribute
#pragma unroll
        for (int k = 0; k < D; k++)                                // _b = b + a*(...); exact zeros of a are not added
            if (M::noisy(k)) bT[k] = bT[k] + g[k];
    }
    if constexpr (NOISE != NOISE_LLONLY) {
        double s[D];
        model.sdw(t, st.y, dw, s);
#pragma unroll
        for (int k = 0; k < D; k++)   // src/euler.jl:264; a structurally zero sigma row contributes an exact "+ 0.0"
            st.y[k] = M::noisy(k) ? st.y[k] + bT[k] * dt + s[k] : st.y[k] + bT[k] * dt;
    }
    PSTAMP(2);   // phase 2: the step's arithmetic (issue; the last result is still in the pipeline)
#if defined(PC_STAMP) && PC_STAMP >= 2
    if constexpr (NOISE == NOISE_EXT) {   // phase 3: wait for the state itself (a dependent move), phase 4: an empty phase = the cost of a mark
        asm volatile("v_mov_b64 %0, %0" : "+v"(st.y[0]));
        asm volatile("s_nop 0" ::: "memory");
    }
#endif
    PSTAMP(3);
    PSTAMP(4);
}

// Minimum waves per SIMD the register allocator must allow.  The BASELINE workloads launch
// 262 144 lanes = 4096 waves = exactly 4 waves per SIMD on 256 CUs: above 128 VGPRs only 3 waves fit,
// the last quarter of the grid runs as a second round and the kernel takes 4/3 as long (measured).
#ifndef BHIP_WPE
#define BHIP_WPE 4
#endif

template <class M, int GK, int MO, int NOISE, int FL, bool PPR = false /* per-chain coefficient rows */>
__global__ __launch_bounds__(256, (PPR || M::D > 4 || (M::D > 3 && NOISE == NOISE_PCN)) ? 2 : BHIP_WPE) void k_paths(const KArgs a)
{
    constexpr int D = M::D, MP = M::MP;
    using RL = RowLayout<GK, D, MO, is_constdiff<M>::value>;
    // the instantiations that draw normals keep the generator's table in LDS (10 KB per block under the default specification v4;
    // fresh proposals can also draw v3 / v2 and load that table, 2.5 KB, instead)
    constexpr bool DRAWS = NOISE == NOISE_FRESH || NOISE == NOISE_PCN;
    constexpr bool TWO_COPIES = DRAWS && NOISE == NOISE_FRESH;   // (three since round 5: one time loop per specification)
    __shared__ __attribute__((aligned(16))) double rng_tab[DRAWS ? (TWO_COPIES ? RNG_LDS_DOUBLES : ICDF_TAB_DOUBLES) : 2];
    if constexpr (DRAWS) {
        if (TWO_COPIES && (a.noise_spec == 2 || a.noise_spec == 3)) TabLDS::load(rng_tab, threadIdx.x, blockDim.x);
        else IcdfLDS::load(rng_tab, threadIdx.x, blockDim.x);
        __syncthreads();
    }
    using TabT = typename bhip_cond<DRAWS, IcdfLDS, TabConst>::type;
    const TabT tab = [&]() { if constexpr (DRAWS) return IcdfLDS(rng_tab); else return TabConst(); }();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.P) return;
    // the model functor: built once from the kernel arguments, or (STREAMED) re-opened from the device copy at every step
    auto make_model = [&]() {
        if constexpr (is_streamed<M>::value) {
            bhip_cptr_t mp = (bhip_cptr_t)(uintptr_t)a.mpar_dev;
            asm volatile("" : "+s"(mp));   // opaque: the loads of this step's operands cannot be hoisted out of the time loop
            return M(mp);
        } else return M(a.mpar);
    };
    const M model = make_model();
    const int N = a.N;
    const int nll = N - 1 - a.skip;
    const cptr_t rows = (cptr_t)(uintptr_t)a.rows;
    // per-chain rows: the chain's RL::LEN - 3 entries of step i+1 are fetched while step i is computed (two register
    // rows in rotation; the loop is unrolled by two) -- per-lane loads, unlike the shared rows' scalar loads
    constexpr int NPP = PPR ? pp_row_len<D>() : 1, NE = PPR ? RL::LEN - 3 : 1;
    using RowT = typename bhip_cond<PPR, ExpRow<NE>, cptr_t>::type;
    RegRow<NPP> rr[2];
    auto fetch_row = [&](int i, RegRow<NPP> &r) {
        if constexpr (PPR) {
            const double *src = a.prows + (size_t)min(i, N - 2) * NPP * a.ldr + p;
#pragma unroll
            for (int q = 0; q < NPP; q++) r.v[q] = src[(size_t)q * a.ldr];
        }
    };
    auto rowat = [&](int i, const RegRow<NPP> &r) {
        if constexpr (PPR) {
            ExpRow<NE> x;
            x.sh = rows + (size_t)i * RL::RS;
            expand_pp_row<M>(model, a.lna, r.v, x.e);
            return x;
        } else return rows + (size_t)i * RL::RS;
    };
    fetch_row(0, rr[0]);

    LaneState<D, MP> st;
#pragma unroll
    for (int k = 0; k < D; k++) st.y[k] = a.x0_dev ? a.x0_dev[k * a.ldx0 + p] : a.x0[k];
    st.ll = 0.0;
    st.zq[0] = st.zq[1] = st.zq[2] = 0.0;
#pragma unroll
    for (int k = 0; k < MP; k++) { st.wprev[k] = 0.0; st.w2prev[k] = 0.0; }

    // per-lane stream pointers
    const double *win = nullptr;  // EXT: driving W, LLONLY: X
    double *wout = nullptr;       // FRESH: W store
    double *xout = nullptr;
    d2v *wslot = nullptr;         // PCN: this chain's W slots
    long ldwi = 0, ldwo = 0, ldx = 0;
    int c = 0;
    if constexpr (NOISE == NOISE_PCN) {
        c = a.cur[p];
        wslot = reinterpret_cast<d2v *>(a.Wc) + p;
        if constexpr ((FL & 1) != 0) { xout = a.Xo + p; ldx = a.ldC; }
    } else {
        if constexpr (NOISE != NOISE_FRESH) { win = in_base(a, p) + p; ldwi = a.ldWin; }
        if constexpr ((FL & 2) != 0) { wout = a.Wout + (size_t)p * a.wstride; ldwo = a.ldWout * a.wstride; }
        if constexpr ((FL & 1) != 0) { xout = x_store_base(a, p) + p; ldx = a.ldX; }
    }
    if constexpr (NOISE == NOISE_EXT) {
#pragma unroll
        for (int k = 0; k < MP; k++) st.wprev[k] = win[k * ldwi];
    }
    if constexpr (NOISE == NOISE_INNOV) {
#pragma unroll
        for (int k = 0; k < D; k++) st.y[k] = win[k * ldwi];   // yy[1]
    }
    if constexpr (NOISE == NOISE_FRESH && (FL & 2) != 0) {
#pragma unroll
        for (int k = 0; k < MP; k++) wout[k * ldwo] = 0.0;   // W[1] = 0
    }
    if constexpr (NOISE == NOISE_PCN) {
#pragma unroll
        for (int k = 0; k < MP; k++) wslot[(size_t)k * a.ldC] = d2v{0.0, 0.0};   // W[1] = Wo[1] = 0
    }
    const uint32_t path = a.path0 + (uint32_t)p;

    // Rolling prefetch window: the per-lane HBM read of step i+PF (driving Wiener value / chain slot, or X
    // for the stand-alone llikelihood) is issued while step i is computed, so PF loads per lane are in
    // flight and their latency overlaps arithmetic instead of being exposed once per step.  Stores are
    // fire-and-forget.  The loop is unrolled by two so that the Philox block parity is static.
    constexpr int PF = NOISE == NOISE_PCN ? (MP > 6 ? 1 : BHIP_KCH_PCN) : BHIP_KCH;   // (m' = 7, 8: two slot rows ahead do not fit the register file)
    constexpr int NIN = (NOISE == NOISE_LLONLY || NOISE == NOISE_INNOV) ? D : MP;
    constexpr bool READS = NOISE == NOISE_EXT || NOISE == NOISE_LLONLY || NOISE == NOISE_INNOV;
    constexpr int OFF = NOISE == NOISE_LLONLY ? 0 : 1;   // INNOV reads X[i+1] (X[0] is loaded up front)
    const int nsteps = N - 1;
    double pf[PF][NIN];
    d2v pfs[PF][MP];
    if constexpr (READS) {
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const int ii = min(j, nsteps - 1) + OFF;   // clamped: short grids re-read a valid row
#pragma unroll
            for (int k = 0; k < NIN; k++) pf[j][k] = ld_stream(&win[((size_t)ii * NIN + k) * ldwi]);
        }
    }
    if constexpr (NOISE == NOISE_PCN) {
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const int ii = min(j, nsteps - 1) + 1;
#pragma unroll
            for (int k = 0; k < MP; k++) pfs[j][k] = ld_stream(&wslot[((size_t)ii * MP + k) * a.ldC]);
        }
    }
    d2v slot[MP];
    auto advance = [&](int i, double (&cur)[NIN]) {
        if constexpr (READS) {
#pragma unroll
            for (int k = 0; k < NIN; k++) cur[k] = pf[0][k];
#pragma unroll
            for (int j = 0; j + 1 < PF; j++)
#pragma unroll
                for (int k = 0; k < NIN; k++) pf[j][k] = pf[j + 1][k];
            const int ii = min(i + PF, nsteps - 1) + OFF;
#pragma unroll
            for (int k = 0; k < NIN; k++) pf[PF - 1][k] = ld_stream(&win[((size_t)ii * NIN + k) * ldwi]);
        }
        if constexpr (NOISE == NOISE_PCN) {
#pragma unroll
            for (int k = 0; k < MP; k++) { slot[k] = pfs[0][k]; cur[k] = c ? slot[k].y : slot[k].x; }
#pragma unroll
            for (int j = 0; j + 1 < PF; j++)
#pragma unroll
                for (int k = 0; k < MP; k++) pfs[j][k] = pfs[j + 1][k];
            const int ii = min(i + PF, nsteps - 1) + 1;
#pragma unroll
            for (int k = 0; k < MP; k++) pfs[PF - 1][k] = ld_stream(&wslot[((size_t)ii * MP + k) * a.ldC]);
        }
    };
    auto commit = [&](int i) {   // PCN: the proposal Wo[i+1] goes into the other half of the slot
        if constexpr (NOISE == NOISE_PCN) {
#pragma unroll
            for (int k = 0; k < MP; k++)
                st_stream(&wslot[((size_t)(i + 1) * MP + k) * a.ldC], c ? d2v{st.wprev[k], slot[k].y} : d2v{slot[k].x, st.wprev[k]});
        }
    };
    // unrolled so that the position of a step's normals inside their quad (four normals per quad) and the register row
    // of the per-chain coefficients are static: four steps per iteration where normals are drawn (two for m' = 2), else two
    constexpr int UNR = (DRAWS ? ((MP == 2 || MP % 4 == 0) ? 2 : 4) : 2) * ((DRAWS && !PPR && M::D <= 3) ? BHIP_PATHS_UNR_MULT : 1);
    // the time loop, generic in the table accessor: its TYPE carries the noise specification (bhip_rng.h), so the kernels that draw
    // hold the loop twice and ONE wave-uniform branch per launch picks the copy -- the default copy is the loop it always was
    auto time_loop = [&](const auto &tb) {
        using TB = typename bhip_unref<decltype(tb)>::type;
        int i = 0;
        for (; i + UNR - 1 < nsteps; i += UNR) {
#pragma unroll
            for (int u = 0; u < UNR; u++) {   // i is a multiple of UNR: u is the step's static phase
                double cur[NIN];
                advance(i + u, cur);
                fetch_row(i + u + 1, rr[(u + 1) & 1]);
                path_step<M, GK, MO, NOISE, FL, RowT, TB>(is_streamed<M>::value ? make_model() : model, a, rowat(i + u, rr[u & 1]), i + u, nll, path, cur, wout, ldwo, xout, ldx, st, tb, 0u, u);
                commit(i + u);
            }
        }
#pragma unroll 1
        for (; i < nsteps; i++) {   // ragged tail (< UNR steps): dynamic phase, the current per-chain row always in rr[0]
            double cur[NIN];
            advance(i, cur);
            fetch_row(i + 1, rr[1]);
            path_step<M, GK, MO, NOISE, FL, RowT, TB>(is_streamed<M>::value ? make_model() : model, a, rowat(i, rr[0]), i, nll, path, cur, wout, ldwo, xout, ldx, st, tb);
            commit(i);
            rr[0] = rr[1];
        }
    };
    // Fresh proposals hold the loop once per specification (every copy one basic block per step: the scheduler interleaves the
    // generator with the recurrence; a branch at the draw instead cost the 4..8-dimensional kernels 40 %).  The pCN step on the slots
    // has no registers for several copies (they spilled) and stays on the default stream: under BHIP_OPT_NOISE_SPEC = 2 / 3 the host sends
    // chains to the wave-specialised kernel (d <= 3) or the tile kernel (d > 3) and refuses what only this kernel could run (do_launch).
    if constexpr (TWO_COPIES) {
        if (a.noise_spec == 2) time_loop(FullRes<TabLDS>(TabLDS(rng_tab)));
        else if (a.noise_spec == 3) time_loop(TabLDS(rng_tab));
        else time_loop(tab);
    } else time_loop(tab);

    if constexpr (NOISE == NOISE_INNOV) {
#pragma unroll
        for (int k = 0; k < D; k++) st_stream(&wout[((size_t)(N - 1) * D + k) * ldwo], st.wprev[k]);   // ww[N] = w
    }
    if constexpr (NOISE != NOISE_LLONLY && NOISE != NOISE_INNOV) {
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
    }

    if constexpr (NOISE == NOISE_PCN) {
        // if log(rand()) <= llo - ll: X<-Xo, W<-Wo (parity flip), ll<-llo, acc+=1
        if (!a.defer_accept) {
            const double u = accept_uniform(a.k0, a.k1, path, a.iter);
            const double llc = a.llcur[p];
            if (det_log(u) <= st.ll - llc) {
                a.cur[p] = (unsigned char)(c ^ 1);
                a.llcur[p] = st.ll;
                a.acc[p] += 1u;
            }
        }
        if (a.ll) a.ll[p] = st.ll;   // llo trace
    } else {
        if (a.ll) a.ll[p] = st.ll;
    }
}

// ---- BHIP_RTC_END  (everything above is device code and is also embedded, flattened, into the
// library for hipRTC-compiled user models -- gen_rtc_src.py; below: host-side launch / dispatch)

typedef hipError_t (*launch_fn)(const KArgs &, hipStream_t);

template <class M, int GK, int MO, int FL, bool PPR = false>
hipError_t launch_chain_lines(const KArgs &a, hipStream_t st);   // bhip_chain_kernel.h
template <class M, int GK, int MO, int MODE, int FL, bool PPR = false>
hipError_t launch_pc(const KArgs &a, hipStream_t st);            // bhip_pc_kernel.h

template <class M, int GK, int MO, int NOISE, int FL, bool PPR = false>
hipError_t launch_paths(const KArgs &a, hipStream_t st)
{
    const int block = BHIP_PATHS_BLOCK;
    const long grid = (a.P + block - 1) / block;
    hipLaunchKernelGGL((k_paths<M, GK, MO, NOISE, FL, PPR>), dim3((unsigned)grid), dim3(block), 0, st, a);
    return hipGetLastError();
}

// per-chain coefficient rows (GuidedBridge's with LinearAppr auxiliaries re-linearised per chain, bhip_guide_kernel.h):
// the pCN proposal on either chain layout and the stand-alone log-likelihood
template <class M>
launch_fn get_launch_ppr(int noise, int fl)
{
    switch (noise) {
    case NOISE_PCN: return (fl & 1) ? launch_paths<M, BHIP_GUIDE_HV, 1, NOISE_PCN, 1, true> : launch_paths<M, BHIP_GUIDE_HV, 1, NOISE_PCN, 0, true>;
    case NOISE_PCN_LINES:
        // wave-specialised kernel with per-chain rows in the consumer (rows fetched one step ahead); k_chain_lines<.., PPR> is its
        // one-lane-does-everything twin (A/B, BHIP_OPT_WAVE_SPECIALISED = 0)
        if constexpr (M::MP <= 3)
            return (fl & 2) ? ((fl & 1) ? launch_chain_lines<M, BHIP_GUIDE_HV, 1, 1, true> : launch_chain_lines<M, BHIP_GUIDE_HV, 1, 0, true>)
                            : ((fl & 1) ? launch_pc<M, BHIP_GUIDE_HV, 1, 7, 1, true> : launch_pc<M, BHIP_GUIDE_HV, 1, 7, 0, true>);
        return nullptr;
    case NOISE_LLONLY: return launch_paths<M, BHIP_GUIDE_HV, 1, NOISE_LLONLY, 0, true>;
    }
    return nullptr;
}

// all (guide, obs-dim, noise, flags) instantiations of one model.
// fl: bit0 store X, bit1 store W (FRESH only), bit2 two-dot log-likelihood (PartialBridge!, NUH guide only).
template <class M, int GK, int MO, int TWO>
launch_fn get_launch_gk(int noise, int fl)
{
    constexpr int T = TWO ? 4 : 0;
    switch (noise) {
    case NOISE_EXT: return (fl & 1) ? launch_paths<M, GK, MO, NOISE_EXT, 1 | T> : launch_paths<M, GK, MO, NOISE_EXT, 0 | T>;
    case NOISE_FRESH:
        switch (fl & 3) {
        case 0: return launch_paths<M, GK, MO, NOISE_FRESH, 0 | T>;
        case 1: return launch_paths<M, GK, MO, NOISE_FRESH, 1 | T>;
        case 2: return launch_paths<M, GK, MO, NOISE_FRESH, 2 | T>;
        default: return launch_paths<M, GK, MO, NOISE_FRESH, 3 | T>;
        }
    case NOISE_PCN:
        if constexpr (GK != BHIP_GUIDE_NONE) return (fl & 1) ? launch_paths<M, GK, MO, NOISE_PCN, 1 | T> : launch_paths<M, GK, MO, NOISE_PCN, 0 | T>;
        return nullptr;
    case NOISE_PCN_LINES:
        if constexpr (GK != BHIP_GUIDE_NONE && M::MP <= 3) return (fl & 1) ? launch_chain_lines<M, GK, MO, 1 | T> : launch_chain_lines<M, GK, MO, 0 | T>;
        return nullptr;
    case 6 /* NOISE_FRESH_PC */:
        if constexpr (M::MP <= 3) {
            switch (fl & 3) {
            case 0: return launch_pc<M, GK, MO, 6, 0 | T>;
            case 1: return launch_pc<M, GK, MO, 6, 1 | T>;
            case 2: return launch_pc<M, GK, MO, 6, 2 | T>;
            default: return launch_pc<M, GK, MO, 6, 3 | T>;
            }
        }
        return nullptr;
    case 7 /* NOISE_PCN_LINES_PC */:
        if constexpr (GK != BHIP_GUIDE_NONE && M::MP <= 3) return (fl & 1) ? launch_pc<M, GK, MO, 7, 1 | T> : launch_pc<M, GK, MO, 7, 0 | T>;
        return nullptr;
    case NOISE_LLONLY:
        if constexpr (GK != BHIP_GUIDE_NONE) return launch_paths<M, GK, MO, NOISE_LLONLY, 0 | T>;
        return nullptr;
    case NOISE_INNOV:
        if constexpr (has_sinv<M>::value && !TWO && GK != BHIP_GUIDE_QF) return launch_paths<M, GK, MO, NOISE_INNOV, 2>;   // (innovations! needs _b itself: never on regrouped rows)
        return nullptr;
    }
    return nullptr;
}

// LinPro targets of dimension 4..8 (one path per lane, matrices through the scalar unit): the guide always in the form
// r = H_i (nu_i - x) -- GuidedBridge pre-inverted on the host like on the MFMA tile kernel, (L,M,mu) mapped likewise --,
// external or fresh noise, stand-alone llikelihood, plain Euler-Maruyama, pCN chains (16-byte slots: current and proposal value of a
// component side by side, the layout of the d <= 3 fall-back).
template <class M>
launch_fn get_launch_mid(int gk, int noise, int fl)
{
    if (gk == BHIP_GUIDE_NONE) {
        switch (noise) {
        case NOISE_EXT: return (fl & 1) ? launch_paths<M, BHIP_GUIDE_NONE, 1, NOISE_EXT, 1> : launch_paths<M, BHIP_GUIDE_NONE, 1, NOISE_EXT, 0>;
        case NOISE_INNOV: return launch_paths<M, BHIP_GUIDE_NONE, 1, NOISE_INNOV, 2>;
        case NOISE_FRESH:
            switch (fl & 3) {
            case 0: return launch_paths<M, BHIP_GUIDE_NONE, 1, NOISE_FRESH, 0>;
            case 1: return launch_paths<M, BHIP_GUIDE_NONE, 1, NOISE_FRESH, 1>;
            case 2: return launch_paths<M, BHIP_GUIDE_NONE, 1, NOISE_FRESH, 2>;
            default: return launch_paths<M, BHIP_GUIDE_NONE, 1, NOISE_FRESH, 3>;
            }
        }
        return nullptr;
    }
    // guided: the regrouped rows (GUIDE_QF) for everything but innovations!, which needs _b = b + a r itself and keeps the (nu, H) rows
    if (gk == BHIP_GUIDE_NUH) return noise == NOISE_INNOV ? launch_paths<M, BHIP_GUIDE_NUH, 1, NOISE_INNOV, 2> : nullptr;   // inv(sigma) by LU on the host, streamed like the other matrices
    switch (noise) {
    case NOISE_EXT: return (fl & 1) ? launch_paths<M, BHIP_GUIDE_QF, 1, NOISE_EXT, 1> : launch_paths<M, BHIP_GUIDE_QF, 1, NOISE_EXT, 0>;
    case NOISE_FRESH:
        switch (fl & 3) {
        case 0: return launch_paths<M, BHIP_GUIDE_QF, 1, NOISE_FRESH, 0>;
        case 1: return launch_paths<M, BHIP_GUIDE_QF, 1, NOISE_FRESH, 1>;
        case 2: return launch_paths<M, BHIP_GUIDE_QF, 1, NOISE_FRESH, 2>;
        default: return launch_paths<M, BHIP_GUIDE_QF, 1, NOISE_FRESH, 3>;
        }
    case NOISE_LLONLY: return launch_paths<M, BHIP_GUIDE_QF, 1, NOISE_LLONLY, 0>;
    case NOISE_PCN: return (fl & 1) ? launch_paths<M, BHIP_GUIDE_QF, 1, NOISE_PCN, 1> : launch_paths<M, BHIP_GUIDE_QF, 1, NOISE_PCN, 0>;   // pCN chains on the 16-byte slots
    }
    return nullptr;
}

template <class M>
launch_fn get_launch(int gk, int mo, int noise, int fl)
{
    constexpr int D = M::D;
    switch (gk) {
    case BHIP_GUIDE_NONE: return get_launch_gk<M, BHIP_GUIDE_NONE, 1, 0>(noise, fl);
    case BHIP_GUIDE_HV: return get_launch_gk<M, BHIP_GUIDE_HV, 1, 0>(noise, fl);
    case BHIP_GUIDE_NUH: return get_launch_gk<M, BHIP_GUIDE_NUH, 1, 0>(noise, fl);
    case BHIP_GUIDE_NUH_INPLACE: return get_launch_gk<M, BHIP_GUIDE_NUH, 1, 1>(noise, fl);
#ifdef BHIP_FUSED
    case BHIP_GUIDE_QF:   // the regrouped step of an affine target at d <= 3 (host: finish_guide; do_launch selects it under BHIP_OPT_FUSED_ARITHMETIC)
        if constexpr (is_affine_target<M>::value) return get_launch_gk<M, BHIP_GUIDE_QF, 1, 0>(noise, fl);
        return nullptr;
#endif
    case BHIP_GUIDE_LMMU:
        if (mo == 1) return get_launch_gk<M, BHIP_GUIDE_LMMU, 1, 0>(noise, fl);
        if constexpr (D >= 2) { if (mo == 2) return get_launch_gk<M, BHIP_GUIDE_LMMU, 2, 0>(noise, fl); }
        if constexpr (D >= 3) { if (mo == 3) return get_launch_gk<M, BHIP_GUIDE_LMMU, 3, 0>(noise, fl); }
        return nullptr;
    }
    return nullptr;
}

}  // namespace bhip
