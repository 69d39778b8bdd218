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

namespace bhip {

enum { NOISE_EXT = 0, NOISE_FRESH = 1, NOISE_PCN = 2, NOISE_LLONLY = 3 };

struct KArgs {
    const double *rows;   // [N-1][rs] packed per-step coefficients (device)
    int rs, N, skip;
    int aux_linpro;       // 1: b~ = B(x - mu~), 0: b~ = B x + beta~
    int ll_two_dots;      // PartialBridge! accumulates dot(b,r)dt - dot(b~,r)dt
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
    double *ll;           // optional per-path log-likelihood
    // pCN chain state (NOISE_PCN): double buffers + per-chain parity
    double *Wb[2];
    double *Xb[2];
    long ldC;
    unsigned char *cur;
    double *llcur;
    unsigned int *acc;
    double rho, srho;
    uint32_t k0, k1, iter, path0;
    double x0[3];
    double vend[3];
    double mu_aux[3];
    double mpar[32];
};

typedef const __attribute__((address_space(4))) double *cptr_t;

template <int GK, int D, int MO>
struct RowLayout {
    static constexpr int T = 0, DT = 1, RDT = 2;
    static constexpr int B = 3;              // D*D col-major
    static constexpr int BETA = 3 + D * D;   // D
    static constexpr int G = 3 + D * D + D;  // guide part
    static constexpr int GLEN = GK == BHIP_GUIDE_HV ? (D == 1 ? 2 : D == 2 ? 7 : 13)
                              : GK == BHIP_GUIDE_LMMU ? (MO * D + MO + 2 * D * MO)
                              : GK == BHIP_GUIDE_NUH ? (D * D + D) : 0;
    static constexpr int LEN = GK == BHIP_GUIDE_NONE ? 3 : G + GLEN;
    static constexpr int RS = (LEN + 1) & ~1;
};

// r((i,t),x,Po) and g = a*L'*M*q | a*r, from the packed row
template <class M, int GK, int MO>
BHIP_DEV void guide_terms(const M &model, cptr_t g, const double *x, const double *vmu_unused, double *r, double *gd)
{
    constexpr int D = M::D;
    if constexpr (GK == BHIP_GUIDE_HV) {
        // Hd[i] \ (V[i] - x)                                   src/guip.jl:192-193
        if constexpr (D == 1) {
            r[0] = (g[1] - x[0]) / g[0];
        } else if constexpr (D == 2) {
            const double w0 = g[5] - x[0], w1 = g[6] - x[1];
            r[0] = (g[3] * w0 - g[2] * w1) / g[4];
            r[1] = (g[0] * w1 - g[1] * w0) / g[4];
        } else {
            const double w0 = g[10] - x[0], w1 = g[11] - x[1], w2 = g[12] - x[2];
            r[0] = (g[0] * w0 + g[1] * w1 + g[2] * w2) / g[9];
            r[1] = (g[3] * w0 + g[4] * w1 + g[5] * w2) / g[9];
            r[2] = (g[6] * w0 + g[7] * w1 + g[8] * w2) / g[9];
        }
        model.amul(r, gd);
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
        cptr_t R = g + MO * D + MO, G = g + MO * D + MO + D * MO;
#pragma unroll
        for (int i = 0; i < D; i++) {
            double sr = R[i] * q[0], sg = G[i] * q[0];
#pragma unroll
            for (int j = 1; j < MO; j++) { sr += R[i + D * j] * q[j]; sg += G[i + D * j] * q[j]; }
            r[i] = sr; gd[i] = sg;
        }
    } else if constexpr (GK == BHIP_GUIDE_NUH) {
        // r = H[i]*(nu[i] - x) ; g = a*r                            src/partialbridgenuH.jl:157-161
        double w[D];
#pragma unroll
        for (int k = 0; k < D; k++) w[k] = g[D * D + k] - x[k];
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = g[i] * w[0];
#pragma unroll
            for (int j = 1; j < D; j++) s += g[i + D * j] * w[j];
            r[i] = s;
        }
        model.amul(r, gd);
    }
}

template <class M, int GK, int MO, int NOISE>
__global__ __launch_bounds__(256) void k_paths(const KArgs a)
{
    constexpr int D = M::D, MP = M::MP;
    using RL = RowLayout<GK, D, MO>;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.P) return;
    const M model(a.mpar);
    const int N = a.N;
    const cptr_t rows = (cptr_t)(uintptr_t)a.rows;

    double y[D];
#pragma unroll
    for (int k = 0; k < D; k++) y[k] = a.x0_dev ? a.x0_dev[k * a.ldx0 + p] : a.x0[k];

    // per-lane stream pointers
    const double *win = nullptr;  // W read (EXT: driving path, PCN: current state, LLONLY: X)
    double *wout = nullptr;       // W write (FRESH: optional store, PCN: proposal)
    double *xout = nullptr;
    long ldwi = 0, ldwo = 0, ldx = 0;
    int c = 0;
    if constexpr (NOISE == NOISE_PCN) {
        c = a.cur[p];
        win = a.Wb[c] + p; wout = a.Wb[c ^ 1] + p; ldwi = ldwo = a.ldC;
        if (a.Xb[0]) { xout = a.Xb[c ^ 1] + p; ldx = a.ldC; }
    } else {
        if (a.Win) { win = a.Win + p; ldwi = a.ldWin; }
        if (a.Wout) { wout = a.Wout + p; ldwo = a.ldWout; }
        if (a.X) { xout = a.X + p; ldx = a.ldX; }
    }

    double ll = 0.0;
    double wprev[MP], w2prev[MP];
#pragma unroll
    for (int k = 0; k < MP; k++) { wprev[k] = 0.0; w2prev[k] = 0.0; }
    if constexpr (NOISE == NOISE_EXT) {
#pragma unroll
        for (int k = 0; k < MP; k++) wprev[k] = win[k * ldwi];
    }
    if constexpr (NOISE == NOISE_FRESH || NOISE == NOISE_PCN) {
        if (wout) {
#pragma unroll
            for (int k = 0; k < MP; k++) wout[k * ldwo] = 0.0;  // W[1] = 0 ; rho*0 + srho*0 = 0
        }
    }
    const uint32_t path = a.path0 + (uint32_t)p;
    double zc = 0.0;  // second normal of the current Philox block

    for (int i = 0; i < N - 1; i++) {
        const cptr_t row = rows + (size_t)i * RL::RS;
        const double t = row[RL::T], dt = row[RL::DT];

        if constexpr (NOISE == NOISE_LLONLY) {
#pragma unroll
            for (int k = 0; k < D; k++) y[k] = win[((size_t)i * D + k) * ldwi];
        }

        // ---- LOOP A / P: the Wiener increment of this step
        double dw[MP];
        if constexpr (NOISE == NOISE_EXT) {
#pragma unroll
            for (int k = 0; k < MP; k++) {
                const double wn = win[((size_t)(i + 1) * MP + k) * ldwi];
                dw[k] = wn - wprev[k];
                wprev[k] = wn;
            }
        } else if constexpr (NOISE == NOISE_FRESH || NOISE == NOISE_PCN) {
            const double rdt = row[RL::RDT];
#pragma unroll
            for (int k = 0; k < MP; k++) {
                const int n = i * MP + k;
                double z;
                if ((n & 1) == 0) normal_pair(a.k0, a.k1, path, a.iter, (uint32_t)(n >> 1), z, zc);
                else z = zc;
                if constexpr (NOISE == NOISE_FRESH) {
                    const double wn = wprev[k] + rdt * z;          // yy[i] = yy[i-1] + rootdt*randn
                    dw[k] = wn - wprev[k];                          // ww[i+1] - ww[i]
                    wprev[k] = wn;
                    if (wout) wout[((size_t)(i + 1) * MP + k) * ldwo] = wn;
                } else {
                    const double wc = win[((size_t)(i + 1) * MP + k) * ldwi];
                    const double w2 = w2prev[k] + rdt * z;
                    const double wo = a.rho * wc + a.srho * w2;     // Wo = rho*W + sqrt(1-rho^2)*W2
                    dw[k] = wo - wprev[k];
                    w2prev[k] = w2;
                    wprev[k] = wo;
                    wout[((size_t)(i + 1) * MP + k) * ldwo] = wo;
                }
            }
        }

        // ---- LOOP B: yy[i] = y (stored BEFORE the update, src/euler.jl:263)
        if constexpr (NOISE != NOISE_LLONLY) {
            if (xout) {
#pragma unroll
                for (int k = 0; k < D; k++) xout[((size_t)i * D + k) * ldx] = y[k];
            }
        }

        double bT[D];
        model.b(t, y, bT);
        if constexpr (GK != BHIP_GUIDE_NONE) {
            double r[D], g[D];
            guide_terms<M, GK, MO>(model, row + RL::G, y, nullptr, r, g);
            // ---- LOOP C: som += dot(b - b~, r)*dt
            if (i < N - 1 - a.skip) {
                double bA[D];
                if (a.aux_linpro) {
                    double xm[D];
#pragma unroll
                    for (int k = 0; k < D; k++) xm[k] = y[k] - a.mu_aux[k];
#pragma unroll
                    for (int q = 0; q < D; q++) {
                        double s = row[RL::B + q] * xm[0];
#pragma unroll
                        for (int j = 1; j < D; j++) s += row[RL::B + q + D * j] * xm[j];
                        bA[q] = s;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < D; q++) {
                        double s = row[RL::B + q] * y[0];
#pragma unroll
                        for (int j = 1; j < D; j++) s += row[RL::B + q + D * j] * y[j];
                        bA[q] = s + row[RL::BETA + q];
                    }
                }
                if (a.ll_two_dots) {
                    double s1 = bT[0] * r[0], s2 = bA[0] * r[0];
#pragma unroll
                    for (int k = 1; k < D; k++) { s1 += bT[k] * r[k]; s2 += bA[k] * r[k]; }
                    ll += s1 * dt;
                    ll -= s2 * dt;
                } else {
                    double s = (bT[0] - bA[0]) * r[0];
#pragma unroll
                    for (int k = 1; k < D; k++) s += (bT[k] - bA[k]) * r[k];
                    ll += s * dt;
                }
            }
#pragma unroll
            for (int k = 0; k < D; k++) bT[k] = bT[k] + g[k];   // _b = b + a*(...)
        }
        if constexpr (NOISE != NOISE_LLONLY) {
            double s[D];
            model.sdw(dw, s);
#pragma unroll
            for (int k = 0; k < D; k++) y[k] = y[k] + bT[k] * dt + s[k];   // src/euler.jl:264
        }
    }

    if constexpr (NOISE != NOISE_LLONLY) {
        if (a.use_vend) {   // endpoint(y, P::GuidedBridge) src/euler.jl:241-242
#pragma unroll
            for (int k = 0; k < D; k++) y[k] = a.vend[k];
        }
        if (xout) {
#pragma unroll
            for (int k = 0; k < D; k++) xout[((size_t)(N - 1) * D + k) * ldx] = y[k];
        }
    }

    if constexpr (NOISE == NOISE_PCN) {
        // if log(rand()) <= llo - ll: X<-Xo, W<-Wo (parity flip), ll<-llo, acc+=1
        const double u = accept_uniform(a.k0, a.k1, path, a.iter);
        const double llc = a.llcur[p];
        if (det_log(u) <= ll - llc) {
            a.cur[p] = (unsigned char)(c ^ 1);
            a.llcur[p] = ll;
            a.acc[p] += 1u;
        }
        if (a.ll) a.ll[p] = ll;   // llo trace
    } else {
        if (a.ll) a.ll[p] = ll;
    }
}

typedef hipError_t (*launch_fn)(const KArgs &, hipStream_t);

template <class M, int GK, int MO, int NOISE>
hipError_t launch_paths(const KArgs &a, hipStream_t st)
{
    const int block = 256;
    const long grid = (a.P + block - 1) / block;
    hipLaunchKernelGGL((k_paths<M, GK, MO, NOISE>), dim3((unsigned)grid), dim3(block), 0, st, a);
    return hipGetLastError();
}

// all (guide, obs-dim, noise) instantiations of one model
template <class M>
launch_fn get_launch(int gk, int mo, int noise)
{
    constexpr int D = M::D;
#define BHIP_N4(GK_, MO_)                                                              \
    switch (noise) {                                                                   \
    case NOISE_EXT: return launch_paths<M, GK_, MO_, NOISE_EXT>;                       \
    case NOISE_FRESH: return launch_paths<M, GK_, MO_, NOISE_FRESH>;                   \
    case NOISE_PCN: return launch_paths<M, GK_, MO_, NOISE_PCN>;                       \
    case NOISE_LLONLY: return launch_paths<M, GK_, MO_, NOISE_LLONLY>;                 \
    }                                                                                  \
    return nullptr;
    switch (gk) {
    case BHIP_GUIDE_NONE:
        if (noise == NOISE_EXT) return launch_paths<M, BHIP_GUIDE_NONE, 1, NOISE_EXT>;
        if (noise == NOISE_FRESH) return launch_paths<M, BHIP_GUIDE_NONE, 1, NOISE_FRESH>;
        return nullptr;
    case BHIP_GUIDE_HV: BHIP_N4(BHIP_GUIDE_HV, 1)
    case BHIP_GUIDE_NUH:
    case BHIP_GUIDE_NUH_INPLACE: BHIP_N4(BHIP_GUIDE_NUH, 1)
    case BHIP_GUIDE_LMMU:
        if (mo == 1) { BHIP_N4(BHIP_GUIDE_LMMU, 1) }
        if constexpr (D >= 2) { if (mo == 2) { BHIP_N4(BHIP_GUIDE_LMMU, 2) } }
        if constexpr (D >= 3) { if (mo == 3) { BHIP_N4(BHIP_GUIDE_LMMU, 3) } }
        return nullptr;
    }
#undef BHIP_N4
    return nullptr;
}

}  // namespace bhip
