// bhip_trig.h -- sin / cos as the drift functions of the built-in processes evaluate them (NclarDiffusion, the sin-drift
// integrated diffusion, Pendulum), ONE definition for the HIP kernels, the host C++ (linearappr of a Pendulum) and -- restated
// in C -- the oracle (bo_sin / bo_cos), so that all three agree bit for bit.
//
// Julia's sin(::Float64) / cos(::Float64) (base/special/trig.jl) are ports of fdlibm: reduction by pi/2 (Cody-Waite with a
// 33 + 33 + 53-bit split of pi/2 for |x| < 2^20*pi/2, Payne-Hanek beyond), then __kernel_sin / __kernel_cos on the reduced
// double-double argument.  This is that algorithm, branch-free: always two reduction steps (118 bits of pi/2 -- the closest
// approach of a double below 2^20*pi/2 to a multiple of pi/2 leaves more than 53 significant bits), quadrant by
// round-to-nearest-even, both kernels evaluated and selected.  ~50 VALU instructions against ~200 (plus scalar-register
// spills) of the device library's sin with its inlined Payne-Hanek path: the NCLAR kernels spend most of their time there.
// Error <= 0.75 ulp on the whole domain (checked against 200-bit arithmetic, tests/test_oracle.py).
// Domain: |x| < 2^20*pi/2 ~ 1.647e6; outside it (and for NaN / Inf) the result is NaN: a drift argument of that size means
// the path has blown up already.  (User processes compiled by hipRTC call whatever their text calls.)
#pragma once

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define BHIP_TRIG_HD __host__ __device__ inline
#else
#define BHIP_TRIG_HD inline
#endif

namespace bhip {

BHIP_TRIG_HD int trig_reduce(double x, double &y0, double &y1)
{
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double fn = __builtin_rint(x * invpio2);
    double r = x - fn * pio2_1;          // fn*pio2_1 is exact: 33 + 20 bits
    const double t = r;
    double w = fn * pio2_2;
    r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    y0 = r - w;
    y1 = (r - y0) - w;
    return (int)fn;
}
BHIP_TRIG_HD double trig_ksin(double x, double y)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
BHIP_TRIG_HD double trig_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
BHIP_TRIG_HD double det_sin(double x)
{
    double y0, y1;
    const int q = trig_reduce(x, y0, y1) & 3;
    const double s = trig_ksin(y0, y1), c = trig_kcos(y0, y1);
    const double r = (q & 1) ? c : s;
    const double v = (q & 2) ? -r : r;
    return __builtin_fabs(x) < 1647099.3291652855 ? v : __builtin_nan("");
}
BHIP_TRIG_HD double det_cos(double x)
{
    double y0, y1;
    const int q = trig_reduce(x, y0, y1) & 3;
    const double s = trig_ksin(y0, y1), c = trig_kcos(y0, y1);
    const double r = (q & 1) ? s : c;
    const double v = ((q + 1) & 2) ? -r : r;
    return __builtin_fabs(x) < 1647099.3291652855 ? v : __builtin_nan("");
}

}  // namespace bhip
