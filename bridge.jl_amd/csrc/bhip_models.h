// bhip_models.h -- device functors for the target processes (Bridge.b / Bridge.sigma / Bridge.a
// methods of the reference) and their host-side descriptions.
//
// A functor is constructed from the device parameter block `dp` = the user's `par` followed by
// derived constants the host computes once (a = sigma*sigma' in the reference's operation order).
// Member functions follow the reference expressions literally (operation order matters: results
// are compared bit-for-bit with the CPU oracle; device code is built with -ffp-contract=off).
//   b(t, x, o)          o = b(t,x,P)
//   sdw(t, x, dw, o)    o = sigma(t,x,P)*dw      (every built-in process has a constant sigma and ignores t, x;
//   amul(t, x, r, o)    o = a(t,x,P)*r            user processes compiled by hipRTC may depend on them)
#pragma once
#include "../../include/bridgehip.h"
#include <hip/hip_runtime.h>
#include "bhip_trig.h"

namespace bhip {

#define BHIP_DEV __device__ __forceinline__
}  // namespace bhip
#include "bhip_stream.h"
namespace bhip {

// x / c for a divisor c that is constant over the kernel (a model parameter).  This is the compiler's own
// correctly rounded fp64 division sequence -- reciprocal seed, two Newton steps, quotient, residual
// correction -- with the divisor-only part evaluated once in the constructor, so that a division in the
// time loop costs 3 VALU instructions instead of 11.  Identical bits to `x / c` whenever the hardware
// sequence would not pre-scale its operands (|c| and |x / c| far from the over/underflow thresholds; the
// sign of a zero quotient may differ); the host admits 2^-200 < |c| < 2^200 only (model_setup).
struct UniformDivisor {
    double c, y;
    BHIP_DEV explicit UniformDivisor(double c_) : c(c_)
    {
        const double y0 = __builtin_amdgcn_rcp(c_);
        const double e0 = __builtin_fma(-c_, y0, 1.0);
        const double y1 = __builtin_fma(y0, e0, y0);
        const double e1 = __builtin_fma(-c_, y1, 1.0);
        y = __builtin_fma(y1, e1, y1);
    }
    BHIP_DEV double div(double a) const
    {
        const double q0 = a * y;
        const double r = __builtin_fma(-c, q0, a);
        return __builtin_fma(r, y, q0);
    }
};

// ---- b = -beta*x, sigma, a = sigma^2                      test/guip.jl:21-23, README.md:75-77
struct MOU {
    static constexpr int D = 1, MP = 1, ID = BHIP_MODEL_OU;
    static constexpr bool noisy(int) { return true; }
    double beta, sig, a, isig;
    BHIP_DEV explicit MOU(const double *p) : beta(p[0]), sig(p[1]), a(p[2]), isig(p[3]) {}
    BHIP_DEV void b(double, const double *x, double *o) const { o[0] = -beta * x[0]; }
    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const { o[0] = sig * dw[0]; }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const { o[0] = a * r[0]; }
    BHIP_DEV void sinv_mul(const double *v, double *o) const { o[0] = isig * v[0]; }   // inv(sigma)*v
};

// ---- LinPro: b = B*(x - mu), sigma, a = sigma*sigma'       src/linpro.jl:65-87
// dp = B(D*D) mu(D) sigma(D*D) a(D*D), column-major
// PTR: where the parameter block is read from -- the kernel arguments (d <= 3), or, for 4 <= d <= 8, a device copy read
// through the constant address space and re-opened at every time step (STREAMED): B, sigma and a are then 3*d*d scalar-unit
// operands too many to keep resident next to the step's coefficient row (d = 4: the compiler parked them in spilled
// scalar registers and fetched every operand back with a v_readlane, 150 per step)
template <int D_, class PTR = const double *>
struct MLinPro {
    static constexpr int D = D_, MP = D_, ID = BHIP_MODEL_LINPRO;
    static constexpr bool STREAMED = D_ > 3;
    static constexpr bool noisy(int) { return true; }
    PTR p;
    BHIP_DEV explicit MLinPro(PTR p_) : p(p_) {}
    BHIP_DEV void b(double, const double *x, double *o) const
    {
        PTR q = p;
        if constexpr (STREAMED) bhip_after(q, x[D - 1]);
        double xm[D];
#pragma unroll
        for (int k = 0; k < D; k++) xm[k] = x[k] - q[D * D + k];
        if constexpr (STREAMED) { matvec_streamed<D, PTR>(q, xm, o); return; }
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = q[i] * xm[0];
#pragma unroll
            for (int j = 1; j < D; j++) s += q[i + D * j] * xm[j];
            o[i] = s;
        }
    }
    // Bridge.bderiv(t, x, P::LinPro) = P.B   src/linpro.jl:82
    BHIP_DEV void bderiv(double, const double *, double *J) const
    {
#pragma unroll
        for (int k = 0; k < D * D; k++) J[k] = p[k];
    }
    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const
    {
        PTR S = p + D * D + D;
        if constexpr (STREAMED) { bhip_after(S, dw[D - 1]); matvec_streamed<D, PTR>(S, dw, o); return; }
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = S[i] * dw[0];
#pragma unroll
            for (int j = 1; j < D; j++) s += S[i + D * j] * dw[j];
            o[i] = s;
        }
    }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const
    {
        PTR A = p + 2 * D * D + D;
        if constexpr (STREAMED) { bhip_after(A, r[D - 1]); matvec_streamed<D, PTR>(A, r, o); return; }
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = A[i] * r[0];
#pragma unroll
            for (int j = 1; j < D; j++) s += A[i + D * j] * r[j];
            o[i] = s;
        }
    }
    BHIP_DEV void sinv_mul(const double *v, double *o) const   // inv(P.sigma)*v
    {
        if constexpr (STREAMED) {
            PTR Sq = p + 3 * D * D + D;
            bhip_after(Sq, v[D - 1]);
            matvec_streamed<D, PTR>(Sq, v, o);
            return;
        }
        const PTR Si = p + 3 * D * D + D;
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = Si[i] * v[0];
#pragma unroll
            for (int j = 1; j < D; j++) s += Si[i + D * j] * v[j];
            o[i] = s;
        }
    }
};

// ---- FitzhughDiffusion                project_partialbridge/partialbridge_fitzhugh.jl:36-46
// b = ((x1-x2-x1^3+s)/eps, gamma*x1-x2+beta), sigma = (0, sigma)  (scalar noise, hypo-elliptic)
// dp = eps,s,gamma,beta,sigma, a22 = sigma*sigma
struct MFHN {
    static constexpr int D = 2, MP = 1, ID = BHIP_MODEL_FHN;
    static constexpr bool noisy(int k) { return k == 1; }   // sigma = (0, sigma): row 0 of sigma*dw is an exact zero
    double s, gam, beta, sig, a22;
    UniformDivisor eps;
    BHIP_DEV explicit MFHN(const double *p) : s(p[1]), gam(p[2]), beta(p[3]), sig(p[4]), a22(p[5]), eps(p[0]) {}
    BHIP_DEV void b(double, const double *x, double *o) const
    {
        o[0] = eps.div(x[0] - x[1] - x[0] * x[0] * x[0] + s);
        o[1] = gam * x[0] - x[1] + beta;
    }
    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const { o[0] = 0.0; o[1] = sig * dw[0]; }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const { o[0] = 0.0; o[1] = a22 * r[1]; }
};

// ---- NclarDiffusion                   project_partialbridge/partialbridge_nclar.jl:52-61
// b = (x2, x3, -alpha*sin(omega*x3)), sigma = (0,0,sigma);  dp = alpha,omega,sigma,a33
struct MNCLAR {
    static constexpr int D = 3, MP = 1, ID = BHIP_MODEL_NCLAR;
    static constexpr bool noisy(int k) { return k == 2; }
    double al, om, sig, a33;
    BHIP_DEV explicit MNCLAR(const double *p) : al(p[0]), om(p[1]), sig(p[2]), a33(p[3]) {}
    BHIP_DEV void b(double, const double *x, double *o) const
    {
        o[0] = x[1]; o[1] = x[2]; o[2] = -al * det_sin(om * x[2]);
    }
    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const { o[0] = 0.0; o[1] = 0.0; o[2] = sig * dw[0]; }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const { o[0] = 0.0; o[1] = 0.0; o[2] = a33 * r[2]; }
};

// ---- IntegratedDiffusion              test/partialbridge.jl:7-15
// b = (x2, -(x2+sin(x2)) + 1/2), sigma = (0,gamma);  dp = gamma, a22
struct MIntDiff {
    static constexpr int D = 2, MP = 1, ID = BHIP_MODEL_INTDIFF;
    static constexpr bool noisy(int k) { return k == 1; }
    double gam, a22;
    BHIP_DEV explicit MIntDiff(const double *p) : gam(p[0]), a22(p[1]) {}
    BHIP_DEV void b(double, const double *x, double *o) const
    {
        o[0] = x[1]; o[1] = -(x[1] + det_sin(x[1])) + 0.5;
    }
    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const { o[0] = 0.0; o[1] = gam * dw[0]; }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const { o[0] = 0.0; o[1] = a22 * r[1]; }
};

// ---- Lorenz                           src/Models.jl:41-58 (test/euler.jl:45-50 with s = 3,3,3)
// dp = th1,th2,th3,s1,s2,s3, a11,a22,a33
struct MLorenz {
    static constexpr int D = 3, MP = 3, ID = BHIP_MODEL_LORENZ;
    static constexpr bool noisy(int) { return true; }
    double t1, t2, t3, s1, s2, s3, a1, a2, a3, i1, i2, i3;
    BHIP_DEV explicit MLorenz(const double *p)
        : t1(p[0]), t2(p[1]), t3(p[2]), s1(p[3]), s2(p[4]), s3(p[5]), a1(p[6]), a2(p[7]), a3(p[8]), i1(p[9]), i2(p[10]), i3(p[11]) {}
    BHIP_DEV void sinv_mul(const double *v, double *o) const { o[0] = i1 * v[0]; o[1] = i2 * v[1]; o[2] = i3 * v[2]; }
    BHIP_DEV void b(double, const double *x, double *o) const
    {
        o[0] = t1 * (x[1] - x[0]);
        o[1] = x[0] * (t2 - x[2]) - x[1];
        o[2] = x[0] * x[1] - t3 * x[2];
    }
    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const { o[0] = s1 * dw[0]; o[1] = s2 * dw[1]; o[2] = s3 * dw[2]; }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const { o[0] = a1 * r[0]; o[1] = a2 * r[1]; o[2] = a3 * r[2]; }    // Bridge.bderiv(t, x, P::Lorenz)  src/Models.jl:49-53 (column-major 3 x 3): the linearisation of LinearAppr guides
    BHIP_DEV void bderiv(double, const double *x, double *J) const
    {
        J[0] = -t1;       J[3] = t1;   J[6] = 0.0;
        J[1] = t2 - x[2]; J[4] = -1.0; J[7] = -x[0];
        J[2] = x[1];      J[5] = x[0]; J[8] = -t3;
    }
};

// ---- Models.FitzHughNagumo            src/Models.jl:9-20  (diagonal 2-d noise)
// dp = eps,s,gamma,beta,s1,s2, a11,a22
struct MFHN2 {
    static constexpr int D = 2, MP = 2, ID = BHIP_MODEL_FHN2;
    static constexpr bool noisy(int) { return true; }
    double s, gam, beta, s1, s2, a1, a2, i1, i2;
    UniformDivisor eps;
    BHIP_DEV explicit MFHN2(const double *p)
        : s(p[1]), gam(p[2]), beta(p[3]), s1(p[4]), s2(p[5]), a1(p[6]), a2(p[7]), i1(p[8]), i2(p[9]), eps(p[0]) {}
    BHIP_DEV void sinv_mul(const double *v, double *o) const { o[0] = i1 * v[0]; o[1] = i2 * v[1]; }
    BHIP_DEV void b(double, const double *x, double *o) const
    {
        o[0] = eps.div(x[0] - x[0] * x[0] * x[0] - x[1] + s);
        o[1] = gam * x[0] - x[1] + beta;
    }
    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const { o[0] = s1 * dw[0]; o[1] = s2 * dw[1]; }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const { o[0] = a1 * r[0]; o[1] = a2 * r[1]; }
};

// ---- Pendulum                         src/Models.jl:69-88
// b = (x2, -theta2*sin(x1)), sigma = (0,gamma);  dp = theta2, gamma, a22
struct MPendulum {
    static constexpr int D = 2, MP = 1, ID = BHIP_MODEL_PENDULUM;
    static constexpr bool noisy(int k) { return k == 1; }
    double th2, gam, a22;
    BHIP_DEV explicit MPendulum(const double *p) : th2(p[0]), gam(p[1]), a22(p[2]) {}
    BHIP_DEV void b(double, const double *x, double *o) const { o[0] = x[1]; o[1] = -th2 * det_sin(x[0]); }
    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const { o[0] = 0.0; o[1] = gam * dw[0]; }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const { o[0] = 0.0; o[1] = a22 * r[1]; }
    // Bridge.bderiv(t, x, P::Pendulum)  src/Models.jl:81-84
    BHIP_DEV void bderiv(double, const double *x, double *J) const { J[0] = 0.0; J[1] = -th2 * det_cos(x[0]); J[2] = 1.0; J[3] = 0.0; }
};

// ---- Wiener{SVector{D}}: b = 0, sigma = a = I            src/wiener.jl:143-167
template <int D_>
struct MWiener {
    static constexpr int D = D_, MP = D_, ID = BHIP_MODEL_WIENER;
    static constexpr bool noisy(int) { return true; }
    BHIP_DEV explicit MWiener(const double *) {}
    // Bridge.bderiv(t, x, P::Wiener) = 0   src/wiener.jl:147
    BHIP_DEV void bderiv(double, const double *, double *J) const
    {
#pragma unroll
        for (int k = 0; k < D * D; k++) J[k] = 0.0;
    }
    BHIP_DEV void b(double, const double *, double *o) const
    {
#pragma unroll
        for (int k = 0; k < D; k++) o[k] = 0.0;
    }
    BHIP_DEV void sdw(double, const double *, const double *dw, double *o) const
    {
#pragma unroll
        for (int k = 0; k < D; k++) o[k] = dw[k];
    }
    BHIP_DEV void amul(double, const double *, const double *r, double *o) const
    {
#pragma unroll
        for (int k = 0; k < D; k++) o[k] = r[k];
    }
    BHIP_DEV void sinv_mul(const double *v, double *o) const
    {
#pragma unroll
        for (int k = 0; k < D; k++) o[k] = v[k];
    }
};

}  // namespace bhip
