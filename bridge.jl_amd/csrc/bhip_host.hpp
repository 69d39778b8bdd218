// bhip_host.hpp -- host-side (C++) part of the product: the path-independent guide
// pre-computation, evaluated exactly as the reference constructors do, and
// the packing of per-step coefficient rows for the device kernel.
//
//   kernelr3                        src/ode.jl:44-49
//   gpHinv! / gpV!                  src/gode.jl:2-3,13,21 via _solvebackward!  src/ode.jl:88-97
//   partialbridgeode!               src/partialbridge.jl:1-22
//   updatenuH+C, partialbridgeodenuH!   src/partialbridgenuH.jl:1-55
//   PartialBridge! (R3!)            src/partialbridgen!.jl:7-56, src/ode!.jl:21-29
//   lptilde / logpdfnormal / traceB src/guip.jl:202-206, src/gaussian.jl:66-75
//
// Small matrices use the StaticArrays closed forms (inv/det/solve for n <= 3), larger ones LU with
// partial pivoting; everything is column-major, products accumulate left to right.
#pragma once
#include "../../include/bridgehip.h"
#include "bhip_trig.h"
#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace bhip {

struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    Mat(int r_, int c_, const double *src) : r(r_), c(c_), a(src, src + (size_t)r_ * c_) {}
    double &operator()(int i, int j) { return a[i + (size_t)r * j]; }
    double operator()(int i, int j) const { return a[i + (size_t)r * j]; }
    int n() const { return r * c; }
};

inline Mat operator*(const Mat &A, const Mat &B)
{
    Mat C(A.r, B.c);
    for (int j = 0; j < B.c; j++)
        for (int i = 0; i < A.r; i++) {
            double s = A(i, 0) * B(0, j);
            for (int l = 1; l < A.c; l++) s += A(i, l) * B(l, j);
            C(i, j) = s;
        }
    return C;
}
inline Mat tr(const Mat &A)
{
    Mat T(A.c, A.r);
    for (int j = 0; j < A.c; j++)
        for (int i = 0; i < A.r; i++) T(j, i) = A(i, j);
    return T;
}
inline Mat operator+(const Mat &A, const Mat &B) { Mat C = A; for (int k = 0; k < C.n(); k++) C.a[k] = A.a[k] + B.a[k]; return C; }
inline Mat operator-(const Mat &A, const Mat &B) { Mat C = A; for (int k = 0; k < C.n(); k++) C.a[k] = A.a[k] - B.a[k]; return C; }
inline Mat operator-(const Mat &A) { Mat C = A; for (int k = 0; k < C.n(); k++) C.a[k] = -A.a[k]; return C; }
inline Mat operator*(double s, const Mat &A) { Mat C = A; for (int k = 0; k < C.n(); k++) C.a[k] = s * A.a[k]; return C; }
inline Mat outer(const Mat &X) { return X * tr(X); }   // src/misc.jl:63
inline double dot(const Mat &x, const Mat &y) { double s = x.a[0] * y.a[0]; for (int k = 1; k < x.n(); k++) s += x.a[k] * y.a[k]; return s; }
inline double trace(const Mat &A) { double s = A(0, 0); for (int k = 1; k < A.r; k++) s += A(k, k); return s; }
inline Mat eye(int n) { Mat I(n, n); for (int k = 0; k < n; k++) I(k, k) = 1.0; return I; }

struct LU {
    int n; std::vector<double> f; std::vector<int> piv; bool ok = true;
    explicit LU(const Mat &A) : n(A.r), f(A.a), piv(A.r)
    {
        for (int k = 0; k < n; k++) {
            int p = k; double best = std::fabs(f[k + (size_t)n * k]);
            for (int i = k + 1; i < n; i++) if (std::fabs(f[i + (size_t)n * k]) > best) { best = std::fabs(f[i + (size_t)n * k]); p = i; }
            piv[k] = p;
            if (best == 0.0) { ok = false; return; }
            if (p != k) for (int j = 0; j < n; j++) std::swap(f[k + (size_t)n * j], f[p + (size_t)n * j]);
            const double inv = 1.0 / f[k + (size_t)n * k];
            for (int i = k + 1; i < n; i++) f[i + (size_t)n * k] *= inv;
            for (int j = k + 1; j < n; j++) {
                const double akj = f[k + (size_t)n * j];
                for (int i = k + 1; i < n; i++) f[i + (size_t)n * j] -= f[i + (size_t)n * k] * akj;
            }
        }
    }
    void solve(double *b) const
    {
        for (int k = 0; k < n; k++) if (piv[k] != k) std::swap(b[k], b[piv[k]]);
        for (int k = 0; k < n; k++) for (int i = k + 1; i < n; i++) b[i] -= f[i + (size_t)n * k] * b[k];
        for (int k = n - 1; k >= 0; k--) { b[k] /= f[k + (size_t)n * k]; for (int i = 0; i < k; i++) b[i] -= f[i + (size_t)n * k] * b[k]; }
    }
};

// StaticArrays det.jl
inline double det(const Mat &A)
{
    const int n = A.r; const double *a = A.a.data();
    if (n == 1) return a[0];
    if (n == 2) return a[0] * a[3] - a[2] * a[1];
    if (n == 3) {
        const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
        return a[0] * c0 + a[1] * c1 + a[2] * c2;
    }
    LU lu(A);
    if (!lu.ok) return 0.0;
    double d = 1.0;
    for (int k = 0; k < n; k++) { d *= lu.f[k + (size_t)n * k]; if (lu.piv[k] != k) d = -d; }
    return d;
}
// StaticArrays inv.jl (1x1, 2x2 adjugate/det, 3x3 cross products), LU otherwise
inline Mat inv(const Mat &A)
{
    const int n = A.r; const double *a = A.a.data();
    Mat R(n, n);
    if (n == 1) { R.a[0] = 1.0 / a[0]; return R; }
    if (n == 2) {
        const double d = det(A);
        R.a[0] = a[3] / d; R.a[1] = -(a[1] / d); R.a[2] = -(a[2] / d); R.a[3] = a[0] / d;
        return R;
    }
    if (n == 3) {
        double x0[3] = {a[0], a[1], a[2]}; const double x1[3] = {a[3], a[4], a[5]}, x2[3] = {a[6], a[7], a[8]};
        double y0[3] = {x1[1] * x2[2] - x1[2] * x2[1], x1[2] * x2[0] - x1[0] * x2[2], x1[0] * x2[1] - x1[1] * x2[0]};
        const double d = x0[0] * y0[0] + x0[1] * y0[1] + x0[2] * y0[2];
        for (int k = 0; k < 3; k++) { x0[k] = x0[k] / d; y0[k] = y0[k] / d; }
        const double y1[3] = {x2[1] * x0[2] - x2[2] * x0[1], x2[2] * x0[0] - x2[0] * x0[2], x2[0] * x0[1] - x2[1] * x0[0]};
        const double y2[3] = {x0[1] * x1[2] - x0[2] * x1[1], x0[2] * x1[0] - x0[0] * x1[2], x0[0] * x1[1] - x0[1] * x1[0]};
        R.a[0] = y0[0]; R.a[1] = y1[0]; R.a[2] = y2[0]; R.a[3] = y0[1]; R.a[4] = y1[1]; R.a[5] = y2[1];
        R.a[6] = y0[2]; R.a[7] = y1[2]; R.a[8] = y2[2];
        return R;
    }
    LU lu(A);
    for (int j = 0; j < n; j++) {
        double *col = &R.a[(size_t)n * j];
        for (int i = 0; i < n; i++) col[i] = (i == j) ? 1.0 : 0.0;
        if (lu.ok) lu.solve(col);
        else for (int i = 0; i < n; i++) col[i] = INFINITY;
    }
    return R;
}
// StaticArrays solve.jl  A \ b
inline Mat solve(const Mat &A, const Mat &b)
{
    const int n = A.r;
    Mat x(n, 1);
    auto a = [&](int i, int j) { return A(i - 1, j - 1); };
    if (n == 1) { x.a[0] = b.a[0] / A.a[0]; return x; }
    if (n == 2) {
        const double d = det(A);
        x.a[0] = (a(2, 2) * b.a[0] - a(1, 2) * b.a[1]) / d;
        x.a[1] = (a(1, 1) * b.a[1] - a(2, 1) * b.a[0]) / d;
        return x;
    }
    if (n == 3) {
        const double d = det(A);
        x.a[0] = ((a(2, 2) * a(3, 3) - a(2, 3) * a(3, 2)) * b.a[0] + (a(1, 3) * a(3, 2) - a(1, 2) * a(3, 3)) * b.a[1] + (a(1, 2) * a(2, 3) - a(1, 3) * a(2, 2)) * b.a[2]) / d;
        x.a[1] = ((a(2, 3) * a(3, 1) - a(2, 1) * a(3, 3)) * b.a[0] + (a(1, 1) * a(3, 3) - a(1, 3) * a(3, 1)) * b.a[1] + (a(1, 3) * a(2, 1) - a(1, 1) * a(2, 3)) * b.a[2]) / d;
        x.a[2] = ((a(2, 1) * a(3, 2) - a(2, 2) * a(3, 1)) * b.a[0] + (a(1, 2) * a(3, 1) - a(1, 1) * a(3, 2)) * b.a[1] + (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) * b.a[2]) / d;
        return x;
    }
    LU lu(A);
    x = b;
    lu.solve(x.a.data());
    return x;
}

// src/gaussian.jl:66-75
inline double logpdfnormal(const Mat &x, const Mat &Sigma)
{
    const int d = x.n();
    const double log2pi = std::log(2 * M_PI);
    if (d == 1) return -(x.a[0] * x.a[0] / Sigma.a[0] + std::log(Sigma.a[0]) + log2pi) / 2;
    Mat S(d, d);
    std::vector<double> y(d);
    for (int j = 0; j < d; j++) {
        double s = Sigma(j, j);
        for (int k = 0; k < j; k++) s -= S(j, k) * S(j, k);
        S(j, j) = std::sqrt(s);
        for (int i = j + 1; i < d; i++) {
            double t = Sigma(i, j);
            for (int k = 0; k < j; k++) t -= S(i, k) * S(j, k);
            S(i, j) = t / S(j, j);
        }
    }
    for (int i = 0; i < d; i++) {
        double t = x.a[i];
        for (int k = 0; k < i; k++) t -= S(i, k) * y[k];
        y[i] = t / S(i, i);
    }
    double n2 = 0, sl = 0;
    for (int i = 0; i < d; i++) { n2 += y[i] * y[i]; sl += std::log(S(i, i)); }
    const double nrm = std::sqrt(n2);
    return -(nrm * nrm + 2 * sl + d * log2pi) / 2;
}

// ---- Ralston-3 step, src/ode.jl:44-49
template <class F>
inline Mat kernelr3(F &&f, double t, const Mat &y, double dt)
{
    const Mat k1 = f(t, y);
    const Mat k2 = f(t + 1.0 / 2 * dt, y + (1.0 / 2 * dt) * k1);
    const Mat k3 = f(t + 3.0 / 4 * dt, y + (3.0 / 4 * dt) * k2);
    return y + dt * ((2.0 / 9) * k1 + (1.0 / 3) * k2 + (4.0 / 9) * k3);
}

// ---- auxiliary linear process Pt
struct Aux {
    int kind = -1, d = 0, mp = 0;
    std::vector<double> par;
    bhip_aux_fn fn = nullptr;
    void *user = nullptr;
    int cb_linpro = 0;
    std::vector<double> cb_mu;
    // BHIP_AUX_LINEARAPPR (src/linpro.jl:181-192): coefficients per grid INDEX -- xx_i (d), B_i (d*d), b_i (d), Sigma_i (d*mp).
    // The time-based accessors below are only ever called with grid times; the index is recovered by exact match.
    std::vector<double> la_tt, la_xx, la_B, la_b, la_S;
    bool la_noise = false;   // the coefficients carry a LinearNoiseAppr (src/guip.jl:114-146): B_i = 0, b_i = slope of its deterministic path
    int la_index(double t) const
    {
        const size_t i = (size_t)(std::lower_bound(la_tt.begin(), la_tt.end(), t) - la_tt.begin());
        return (int)std::min(i, la_tt.size() - 1);
    }
    Mat Bi(int i) const { return Mat(d, d, la_B.data() + (size_t)i * d * d); }
    Mat xxi(int i) const { return Mat(d, 1, la_xx.data() + (size_t)i * d); }
    Mat bi(int i) const { return Mat(d, 1, la_b.data() + (size_t)i * d); }
    Mat Si(int i) const { return Mat(d, mp, la_S.data() + (size_t)i * d * mp); }

    bool linpro_form() const { return kind == BHIP_AUX_LINPRO || (kind == BHIP_AUX_CALLBACK && cb_linpro); }
    const double *mu() const { return kind == BHIP_AUX_LINPRO ? par.data() + d * d : cb_mu.data(); }
    double uv(double t) const
    {   // partialbridge_fitzhugh.jl:70-73
        const double lam = (t - par[5]) / (par[7] - par[5]);
        return par[8] * lam + par[6] * (1 - lam);
    }
    void callback(double t, Mat *B, Mat *beta, Mat *a) const
    {
        Mat b_(d, d), be_(d, 1), a_(d, d);
        fn(t, b_.a.data(), be_.a.data(), a_.a.data(), user);
        if (B) *B = b_;
        if (beta) *beta = be_;
        if (a) *a = a_;
    }
    Mat B(double t) const
    {
        if (kind == BHIP_AUX_LINEARAPPR) return Bi(la_index(t));                                  // B((i,s), P) = P.B[i]   :188
        if (kind == BHIP_AUX_CALLBACK) { Mat b_; callback(t, &b_, nullptr, nullptr); return b_; }
        if (kind == BHIP_AUX_FHN_STARTEND) {   // :103
            const double u = uv(t);
            Mat b_(2, 2);
            b_(0, 0) = 1 / par[0] - 3 * (u * u) / par[0]; b_(0, 1) = -1 / par[0];
            b_(1, 0) = par[2]; b_(1, 1) = -1.0;
            return b_;
        }
        return Mat(d, d, par.data());
    }
    Mat beta(double t) const
    {
        if (kind == BHIP_AUX_LINEARAPPR) { const int i = la_index(t); return bi(i) - Bi(i) * xxi(i); }   // P.b[i] - P.B[i]*P.xx[i]   :189
        if (kind == BHIP_AUX_CALLBACK) { Mat b_; callback(t, nullptr, &b_, nullptr); return b_; }
        if (kind == BHIP_AUX_FHN_STARTEND) {   // :104
            const double u = uv(t);
            Mat b_(2, 1);
            b_.a[0] = par[1] / par[0] + 2 * (u * u * u) / par[0];
            b_.a[1] = par[3];
            return b_;
        }
        if (kind == BHIP_AUX_LINPRO) return (-Mat(d, d, par.data())) * Mat(d, 1, par.data() + d * d);   // -P.B*P.mu
        return Mat(d, 1, par.data() + d * d);
    }
    bool has_sigma() const { return kind != BHIP_AUX_CALLBACK; }
    Mat sigma(double t) const
    {
        if (kind == BHIP_AUX_LINEARAPPR) return Si(la_index(t));                                  // a = outer(P.Sigma[i])  :190-191
        if (kind == BHIP_AUX_FHN_STARTEND) { Mat s(2, 1); s.a[0] = 0.0; s.a[1] = par[4]; return s; }
        return Mat(d, mp, par.data() + d * d + d);
    }
    Mat a(double t) const
    {
        if (kind == BHIP_AUX_CALLBACK) { Mat a_; callback(t, nullptr, nullptr, &a_); return a_; }
        return outer(sigma(t));
    }
};

// ---- guide arrays (all grid indices, reference layouts)
struct Guide {
    int kind = BHIP_GUIDE_NONE, m = 0;
    std::vector<Mat> Hd, V;          // HV
    std::vector<Mat> L, M, mu; Mat v;  // LMMU
    std::vector<Mat> nu, H; double C = 0.0;  // NUH
    double trB = 0.0; bool have_trB = false;
};

// GuidedBridge(tt, P, Pt, v, hT)  src/guip.jl:172-180
inline void guide_hv(const std::vector<double> &tt, const Aux &Pt, const Mat &v, const Mat &hT, Guide &g)
{
    const int N = (int)tt.size();
    g.kind = BHIP_GUIDE_HV; g.m = Pt.d;
    g.Hd.assign(N, Mat()); g.V.assign(N, Mat());
    auto dHinv = [&](double t, const Mat &K) { const Mat B = Pt.B(t); return B * K + K * tr(B) - Pt.a(t); };   // gode.jl:3
    auto F = [&](double t, const Mat &x) { return Pt.B(t) * x + Pt.beta(t); };                                  // gode.jl:2
    Mat y = hT;
    g.Hd[N - 1] = y;
    for (int i = N - 2; i >= 0; i--) { y = kernelr3(dHinv, tt[i + 1], y, tt[i] - tt[i + 1]); g.Hd[i] = y; }
    Mat w = v;
    g.V[N - 1] = w;
    for (int i = N - 2; i >= 0; i--) { w = kernelr3(F, tt[i + 1], w, tt[i] - tt[i + 1]); g.V[i] = w; }
    // traceB(tt, Pt) = solve(R3(), _traceB, tt, 0.0, Pt)   src/guip.jl:202-203
    Mat s(1, 1);
    auto trB = [&](double t, const Mat &) { Mat o(1, 1); o.a[0] = trace(Pt.B(t)); return o; };
    for (int i = 1; i < N; i++) s = kernelr3(trB, tt[i - 1], s, tt[i] - tt[i - 1]);
    g.trB = s.a[0]; g.have_trB = true;
}

// GuidedBridge(tt, P, Pt::LinearAppr, v, hT)  src/guip.jl:181-189 with solvebackwardi! / kerneli(::Heun)  src/ode.jl:98-113.
// As committed, kerneli reads an `i` that solvebackwardi! never passes and the V equation hands `b` where only `_b` exists for
// a LinearAppr: the reference constructor cannot run.  Restated with the evident intention (i = the loop index, b = _b):
//     k1 = f_i(y);  k2 = f_{i+1}(y + dt*k1);  y <- y + dt/2*(k1 + k2),  dt = tt[i] - tt[i+1]     (1-based i = N-1 .. 1)
//     f_j(K) = B_j K + K B_j' - outer(Sigma_j),   f_j(x) = B_j (x - xx_j) + b_j                   src/linpro.jl:187-191
inline void guide_hv_heuni(const std::vector<double> &tt, const Aux &Pt, const Mat &v, const Mat &hT, Guide &g)
{
    const int N = (int)tt.size();
    g.kind = BHIP_GUIDE_HV; g.m = Pt.d;
    g.Hd.assign(N, Mat()); g.V.assign(N, Mat());
    auto fH = [&](int j, const Mat &K) { const Mat B = Pt.Bi(j); return B * K + K * tr(B) - outer(Pt.Si(j)); };
    auto fV = [&](int j, const Mat &x) { return Pt.Bi(j) * (x - Pt.xxi(j)) + Pt.bi(j); };
    Mat y = hT;
    g.Hd[N - 1] = y;
    for (int i = N - 2; i >= 0; i--) {
        const double dt = tt[i] - tt[i + 1];
        const Mat k1 = fH(i, y), k2 = fH(i + 1, y + dt * k1);
        y = y + (dt / 2) * (k1 + k2);
        g.Hd[i] = y;
    }
    Mat w = v;
    g.V[N - 1] = w;
    for (int i = N - 2; i >= 0; i--) {
        const double dt = tt[i] - tt[i + 1];
        const Mat k1 = fV(i, w), k2 = fV(i + 1, w + dt * k1);
        w = w + (dt / 2) * (k1 + k2);
        g.V[i] = w;
    }
    g.have_trB = false;   // traceB needs B(t, P) at off-grid times: lptilde is not defined for a LinearAppr auxiliary
}

// Bridge.bderiv(t, x, P) for the processes the reference defines it for (Lorenz src/Models.jl:49-53, Pendulum :81-84,
// LinPro src/linpro.jl:82, Wiener src/wiener.jl:147) and linearappr(Y, P)  src/linpro.jl:196-204
// (host_bderiv / host_b are defined after ModelHost, below)

// partialbridgeode!(::R3, ...)  src/partialbridge.jl:1-22
inline void guide_lmmu(const std::vector<double> &tt, const Aux &Pt, const Mat &L0, const Mat &v, const Mat &Sigma, Guide &g)
{
    const int N = (int)tt.size(), m = L0.r;
    g.kind = BHIP_GUIDE_LMMU; g.m = m; g.v = v;
    g.L.assign(N, Mat()); g.M.assign(N, Mat()); g.mu.assign(N, Mat());
    Mat L = L0, Mp = Sigma, mu(m, 1);
    for (int k = 0; k < m; k++) mu.a[k] = 0 * L0(k, 0);
    g.L[N - 1] = L; g.M[N - 1] = inv(Sigma); g.mu[N - 1] = mu;
    for (int i = N - 2; i >= 0; i--) {
        const double dt = tt[i] - tt[i + 1];
        L = kernelr3([&](double t, const Mat &y) { return (-y) * Pt.B(t); }, tt[i + 1], L, dt);
        Mp = kernelr3([&](double t, const Mat &) {
                 if (Pt.has_sigma()) return -outer(L * Pt.sigma(t));
                 return -((L * Pt.a(t)) * tr(L));
             }, tt[i + 1], Mp, dt);
        mu = kernelr3([&](double t, const Mat &) { return (-L) * Pt.beta(t); }, tt[i + 1], mu, dt);
        g.L[i] = L; g.M[i] = inv(Mp); g.mu[i] = mu;
    }
}

// updatenuH+C + partialbridgeodenuH!(::R3, ...)  src/partialbridgenuH.jl:1-55
inline void guide_nuh(const std::vector<double> &tt, const Aux &Pt, const Mat &L, const Mat &v, double eps, const Mat &Sigma, Guide &g)
{
    const int N = (int)tt.size(), d = Pt.d, m = L.r;
    g.kind = BHIP_GUIDE_NUH; g.m = m;
    g.nu.assign(N, Mat()); g.H.assign(N, Mat());
    const Mat Si = inv(Sigma);
    Mat H0 = (tr(L) * Si) * L;
    for (int k = 0; k < d; k++) H0(k, k) = H0(k, k) + eps;
    Mat Hp = inv(H0);
    Mat nu = ((Hp * tr(L)) * Si) * v;
    double C = 0.0;
    C += 0.5 * dot(v, solve(Sigma, v));
    C += m / 2.0 * std::log(2 * M_PI) + 0.5 * std::log(det(Sigma));
    Mat H = inv(Hp);
    g.H[N - 1] = H; g.nu[N - 1] = nu;
    auto dHp = [&](double t, const Mat &y) { const Mat By = Pt.B(t) * y; return By + tr(By) - Pt.a(t); };
    auto bt = [&](double t, const Mat &y) { return Pt.B(t) * y + Pt.beta(t); };
    for (int i = N - 2; i >= 0; i--) {
        const double dt = tt[i] - tt[i + 1];
        Hp = kernelr3(dHp, tt[i + 1], Hp, dt);
        const Mat F = H * nu;
        const Mat a = Pt.a(tt[i + 1]);
        C += (dot(Pt.beta(tt[i + 1]), F) + 0.5 * dot(F, a * F) - 0.5 * trace(H * a)) * dt;
        nu = kernelr3(bt, tt[i + 1], nu, dt);
        g.nu[i] = nu;
        H = inv(Hp);
        g.H[i] = H;
    }
    g.C = C;
}

// PartialBridge!(tt, P, Pt, L, v, eps, Sigmanoise)  src/partialbridgen!.jl:13-55
inline void guide_nuh_inplace(const std::vector<double> &tt, const Aux &Pt, const Mat &L, const Mat &v, double eps, const Mat &Sn, Guide &g)
{
    const int N = (int)tt.size(), d = Pt.d, m = L.r;
    g.kind = BHIP_GUIDE_NUH_INPLACE; g.m = m;
    g.nu.assign(N, Mat()); g.H.assign(N, Mat());
    bool zero = true;
    for (double x : Sn.a) if (x != 0.0) zero = false;
    Mat G(m, m);
    if (zero) for (int k = 0; k < m; k++) G(k, k) = 1.0 / 2.220446049250313e-16;
    else G = inv(Sn);
    Mat H0 = (tr(L) * G) * L;
    for (int k = 0; k < d; k++) H0(k, k) = H0(k, k) + eps;
    Mat S = inv(H0);
    Mat nu = ((S * tr(L)) * G) * v;
    std::vector<Mat> St(N);
    St[N - 1] = S; g.nu[N - 1] = nu;
    auto dP = [&](double t, const Mat &y) { const Mat By = Pt.B(t) * y; return By + tr(By) - Pt.a(t); };
    auto bt = [&](double t, const Mat &y) { return Pt.B(t) * y + Pt.beta(t); };
    for (int i = N - 2; i >= 0; i--) {
        const double dt = tt[i] - tt[i + 1];
        nu = kernelr3(bt, tt[i + 1], nu, dt);
        S = kernelr3(dP, tt[i + 1], S, dt);
        g.nu[i] = nu; St[i] = S;
    }
    for (int i = 0; i < N; i++) g.H[i] = inv(St[i]);
}

// gpupdate(Hd, V, L, Sigma, v)  src/guip.jl:221-231 -- the backward link between chained GuidedBridge
// segments (test/smoothing.jl:73-83): fold the observation v = L x + N(0, Sigma) into (Hdiamond, V).
inline void gpupdate(const Mat &Hd, const Mat &V, const Mat &L, const Mat &Sigma, const Mat &v, Mat &Hd_out, Mat &V_out)
{
    const int d = Hd.r;
    bool allinf = true;
    for (int k = 0; k < d; k++) if (!(std::isinf(Hd(k, k)) && Hd(k, k) > 0)) allinf = false;
    const Mat Si = inv(Sigma);
    if (allinf) {
        const Mat LtSi = tr(L) * Si;
        const Mat A = LtSi * L;
        Hd_out = inv(A);
        V_out = solve(A, LtSi * v);
        return;
    }
    const Mat S = Sigma + (L * Hd) * tr(L);
    const Mat Z = eye(d) - ((Hd * tr(L)) * inv(S)) * L;
    const Mat ZH = Z * Hd;
    Hd_out = ZH;
    V_out = ((ZH * tr(L)) * Si) * v + Z * V;
}

// ---- target model, host view
struct ModelHost {
    int id = -1, d = 0, mp = 0;
    std::vector<double> par;   // user parameters
    std::vector<double> dpar;  // device parameter block (par + derived constants)
    Mat a;                     // a = sigma*sigma' (constant diffusivity)
    bool constdiff = true;     // false: hipRTC user process with a state-dependent sigma(t,x,P)
};

inline int model_setup(int id, int d_hint, const double *par, int npar, ModelHost &mh, std::string &err)
{
    int need = 0, d = 0, mp = 0;
    switch (id) {
    case BHIP_MODEL_WIENER: d = mp = d_hint; need = 0; break;
    case BHIP_MODEL_OU: d = mp = 1; need = 2; break;
    case BHIP_MODEL_LINPRO: d = mp = d_hint; need = 2 * d * d + d; break;
    case BHIP_MODEL_FHN: d = 2; mp = 1; need = 5; break;
    case BHIP_MODEL_NCLAR: d = 3; mp = 1; need = 3; break;
    case BHIP_MODEL_INTDIFF: d = 2; mp = 1; need = 1; break;
    case BHIP_MODEL_LORENZ: d = 3; mp = 3; need = 6; break;
    case BHIP_MODEL_FHN2: d = 2; mp = 2; need = 6; break;
    case BHIP_MODEL_PENDULUM: d = 2; mp = 1; need = 2; break;
    default: err = "unknown model id"; return BHIP_EINVAL;
    }
    if (d < 1) { err = "model needs a positive dimension"; return BHIP_EINVAL; }
    if (d_hint > 0 && d_hint != d) { err = "dimension does not match the model"; return BHIP_EINVAL; }
    if (npar != need) { err = "wrong number of model parameters"; return BHIP_EINVAL; }
    if (id == BHIP_MODEL_FHN || id == BHIP_MODEL_FHN2) {   // the kernels divide by eps with a hoisted reciprocal (UniformDivisor)
        const double ae = std::fabs(par[0]);
        if (!(ae > 0x1.0p-200 && ae < 0x1.0p200)) { err = "FitzHugh-Nagumo: eps must satisfy 2^-200 < |eps| < 2^200"; return BHIP_EINVAL; }
    }
    mh.id = id; mh.d = d; mh.mp = mp;
    mh.par.assign(par, par + npar);
    Mat S(d, mp);
    switch (id) {
    case BHIP_MODEL_WIENER: for (int k = 0; k < d; k++) S(k, k) = 1.0; break;
    case BHIP_MODEL_OU: S.a[0] = par[1]; break;
    case BHIP_MODEL_LINPRO: S = Mat(d, d, par + d * d + d); break;
    case BHIP_MODEL_FHN: S.a[1] = par[4]; break;
    case BHIP_MODEL_NCLAR: S.a[2] = par[2]; break;
    case BHIP_MODEL_INTDIFF: S.a[1] = par[0]; break;
    case BHIP_MODEL_LORENZ: for (int k = 0; k < 3; k++) S(k, k) = par[3 + k]; break;
    case BHIP_MODEL_FHN2: S(0, 0) = par[4]; S(1, 1) = par[5]; break;
    case BHIP_MODEL_PENDULUM: S.a[1] = par[1]; break;
    }
    mh.a = outer(S);   // src/types.jl:32  a = outer(sigma);  src/linpro.jl:72  a = sigma*sigma'
    mh.dpar = mh.par;
    switch (id) {
    // models with a square sigma also carry inv(sigma) (innovations!, src/euler.jl:371): scalar inv(x) = 1/x,
    // SMatrix inv (closed forms), SDiagonal inv = 1 ./ diag
    case BHIP_MODEL_OU: mh.dpar.push_back(mh.a.a[0]); mh.dpar.push_back(1.0 / par[1]); break;
    case BHIP_MODEL_LINPRO: {
        mh.dpar.insert(mh.dpar.end(), mh.a.a.begin(), mh.a.a.end());
        // (d <= 3: closed forms; 4 <= d <= 12, the path-per-lane kernels' range: LU.  The tile kernel builds its own constants.)
        if (d <= 12) { const Mat Si = det(S) != 0.0 ? inv(S) : Mat(d, d); mh.dpar.insert(mh.dpar.end(), Si.a.begin(), Si.a.end()); }
        break; }
    case BHIP_MODEL_FHN: case BHIP_MODEL_INTDIFF: case BHIP_MODEL_PENDULUM: mh.dpar.push_back(mh.a(1, 1)); break;
    case BHIP_MODEL_NCLAR: mh.dpar.push_back(mh.a(2, 2)); break;
    case BHIP_MODEL_LORENZ:
        for (int k = 0; k < 3; k++) mh.dpar.push_back(mh.a(k, k));
        for (int k = 0; k < 3; k++) mh.dpar.push_back(1.0 / par[3 + k]);
        break;
    case BHIP_MODEL_FHN2:
        mh.dpar.push_back(mh.a(0, 0)); mh.dpar.push_back(mh.a(1, 1));
        mh.dpar.push_back(1.0 / par[4]); mh.dpar.push_back(1.0 / par[5]);
        break;
    default: break;
    }
    return BHIP_OK;
}

// ---- per-step rows for the d <= 3 kernel (layout: RowLayout in bhip_path_kernel.h)
inline int row_stride(int gk, int d, int mo, bool constdiff = true)
{
    if (gk == BHIP_GUIDE_NONE) return 4;
    if (gk == 5 /* BHIP_GUIDE_QF, bhip_path_kernel.h: A_i, bv_i, P_i, q_i, c0_i */) return (3 + 2 * d * d + 2 * d + 1 + 1) & ~1;
    int glen = 0;
    if (gk == BHIP_GUIDE_HV) glen = d == 1 ? 3 : d == 2 ? 8 : 14;   // + the reciprocal of the row's divisor (bhip_smallmat.h sm_recip)
    else if (gk == BHIP_GUIDE_LMMU) glen = mo * d + mo + 2 * d * mo + (constdiff ? 0 : 2 * d * d);
    else glen = d * d + d;
    const int len = 3 + d * d + d + glen;
    return (len + 1) & ~1;
}

inline bool host_bderiv(const ModelHost &mh, const double *x, Mat &J)
{
    const int d = mh.d;
    const double *p = mh.par.data();
    J = Mat(d, d);
    switch (mh.id) {
    case BHIP_MODEL_LORENZ:
        J(0, 0) = -p[0];       J(0, 1) = p[0]; J(0, 2) = 0.0;
        J(1, 0) = p[1] - x[2]; J(1, 1) = -1.0; J(1, 2) = -x[0];
        J(2, 0) = x[1];        J(2, 1) = x[0]; J(2, 2) = -p[2];
        return true;
    case BHIP_MODEL_PENDULUM:
        J(0, 0) = 0.0;                      J(0, 1) = 1.0;
        J(1, 0) = -p[0] * det_cos(x[0]);   J(1, 1) = 0.0;
        return true;
    case BHIP_MODEL_LINPRO: J = Mat(d, d, p); return true;
    case BHIP_MODEL_WIENER: return true;
    }
    return false;
}
inline bool host_b(const ModelHost &mh, const double *x, Mat &o)
{
    const int d = mh.d;
    const double *p = mh.par.data();
    o = Mat(d, 1);
    switch (mh.id) {
    case BHIP_MODEL_LORENZ:   // src/Models.jl:47
        o.a[0] = p[0] * (x[1] - x[0]); o.a[1] = x[0] * (p[1] - x[2]) - x[1]; o.a[2] = x[0] * x[1] - p[2] * x[2];
        return true;
    case BHIP_MODEL_PENDULUM: o.a[0] = x[1]; o.a[1] = -p[0] * det_sin(x[0]); return true;   // :79
    case BHIP_MODEL_LINPRO: { Mat xm(d, 1); for (int k = 0; k < d; k++) xm.a[k] = x[k] - p[d * d + k]; o = Mat(d, d, p) * xm; return true; }
    case BHIP_MODEL_WIENER: return true;
    }
    return false;
}

inline void pack_rows(const std::vector<double> &tt, const ModelHost &mh, const Aux *Pt, const Guide &g, std::vector<double> &rows, int &rs)
{
    const int N = (int)tt.size(), d = mh.d;
    const int gk = g.kind == BHIP_GUIDE_NUH_INPLACE ? BHIP_GUIDE_NUH : g.kind;
    rs = row_stride(gk, d, g.m, mh.constdiff);
    rows.assign((size_t)(N - 1) * rs, 0.0);
    for (int i = 0; i < N - 1; i++) {
        double *r = &rows[(size_t)i * rs];
        r[0] = tt[i];
        r[1] = tt[i + 1] - tt[i];
        r[2] = std::sqrt(tt[i + 1] - tt[i]);   // rootdt of src/wiener.jl:27,53
        if (gk == BHIP_GUIDE_NONE) continue;
        // kernel evaluates b~ = B(x - mu~) + beta~: affine form has mu~ = 0, LinPro form B(x - mu) has beta~ = 0
        const Mat B = Pt->B(tt[i]);
        const Mat be = Pt->linpro_form() ? Mat(d, 1) : Pt->beta(tt[i]);
        std::memcpy(r + 3, B.a.data(), sizeof(double) * d * d);
        std::memcpy(r + 3 + d * d, be.a.data(), sizeof(double) * d);
        double *q = r + 3 + d * d + d;
        if (gk == BHIP_GUIDE_HV) {
            const Mat &A = g.Hd[i]; const Mat &V = g.V[i];
            auto a = [&](int ii, int jj) { return A(ii - 1, jj - 1); };
            if (d == 1) { q[0] = A.a[0]; q[1] = V.a[0]; q[2] = 1.0 / q[0]; }
            else if (d == 2) { std::memcpy(q, A.a.data(), 4 * sizeof(double)); q[4] = det(A); q[5] = V.a[0]; q[6] = V.a[1]; q[7] = 1.0 / q[4]; }
            else {   // cofactor rows of StaticArrays' 3x3 solve
                q[0] = a(2, 2) * a(3, 3) - a(2, 3) * a(3, 2); q[1] = a(1, 3) * a(3, 2) - a(1, 2) * a(3, 3); q[2] = a(1, 2) * a(2, 3) - a(1, 3) * a(2, 2);
                q[3] = a(2, 3) * a(3, 1) - a(2, 1) * a(3, 3); q[4] = a(1, 1) * a(3, 3) - a(1, 3) * a(3, 1); q[5] = a(1, 3) * a(2, 1) - a(1, 1) * a(2, 3);
                q[6] = a(2, 1) * a(3, 2) - a(2, 2) * a(3, 1); q[7] = a(1, 2) * a(3, 1) - a(1, 1) * a(3, 2); q[8] = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
                q[9] = det(A); q[10] = V.a[0]; q[11] = V.a[1]; q[12] = V.a[2]; q[13] = 1.0 / q[9];
            }
        } else if (gk == BHIP_GUIDE_LMMU) {
            const int m = g.m;
            const Mat &L = g.L[i]; const Mat &M = g.M[i];
            std::memcpy(q, L.a.data(), sizeof(double) * m * d);
            for (int j = 0; j < m; j++) q[m * d + j] = g.v.a[j] - g.mu[i].a[j];   // (v - mu[i])
            const Mat R = tr(L) * M;                 // r = L'*M*(...)            src/partialbridge.jl:57
            std::memcpy(q + m * d + m, R.a.data(), sizeof(double) * d * m);
            if (mh.constdiff) {
                const Mat G = (mh.a * tr(L)) * M;    // a*L'*M*(...)              src/partialbridge.jl:54
                std::memcpy(q + m * d + m + d * m, G.a.data(), sizeof(double) * d * m);
            } else {
                // a(t,x) is evaluated in the kernel: the slot holds M; then H = L'*M*L (src/partialbridge.jl:58)
                // and the auxiliary a~(t_i) for the extra log-likelihood terms (src/partialbridge.jl:79-84)
                std::memcpy(q + m * d + m + d * m, M.a.data(), sizeof(double) * m * m);
                const Mat H = R * L, at = Pt->a(tt[i]);
                std::memcpy(q + m * d + m + 2 * d * m, H.a.data(), sizeof(double) * d * d);
                std::memcpy(q + m * d + m + 2 * d * m + d * d, at.a.data(), sizeof(double) * d * d);
            }
        } else {
            std::memcpy(q, g.H[i].a.data(), sizeof(double) * d * d);
            std::memcpy(q + d * d, g.nu[i].a.data(), sizeof(double) * d);
        }
    }
}

}  // namespace bhip
