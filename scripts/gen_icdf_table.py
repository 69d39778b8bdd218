#!/usr/bin/env python3
"""Generates the constant table of the noise specification "bhip-philox-v4" (DESIGN.md section 4): a piecewise
polynomial inverse of the standard normal distribution function on 256 segments, and writes it as C hex-float literals:

    bridge.jl_amd/csrc/bhip_icdf_table.h      product (host + device + hipRTC)
    oracle/bo_icdf_table.h                    the CPU oracle's copy (spec DATA, not code)

One 32-bit Philox word w gives one standard normal:

    v = 2*(w mod 2^31) + 1          odd, 1 <= v < 2^32:  upper-tail probability  p = v * 2^-33  in (0, 1/2)
    d = (double) v                  exact;  hi = the high 32 bits of d  (11 exponent bits, 20 mantissa bits)
    R = (hi >> 17) & 255            the low five exponent bits and the top three mantissa bits: octave e = 31 - floor(log2 v) in 0..31
                                    (e = (30 - (R >> 3)) mod 32), eighth s = R & 7 of the octave:  2^(31-e) (1 + s/8) <= d < 2^(31-e) (1 + (s+1)/8)
    |z| = c0 + d*(c1 + d*(c2 + d*(c3 + d*c4)))     Horner, every step one fma, row R = {c0 .. c4} of the table
    z = |z| with the sign bit of w (bit 31)

The row is the degree-4 minimax polynomial (Remez, absolute error) of  d -> -Phi^-1(d * 2^-33)  on the row's interval, written
in powers of d itself (no local variable: the exact power-of-two scale of the octave sits in the coefficients), each
coefficient rounded to the nearest double.  Worst absolute error over all rows ~3.7e-9 (checked below against scipy's ndtri
on a dense grid of every row and written into the header); Kolmogorov distance of the marginal to N(0,1) <= ~1.3e-9.
|z| <= -Phi^-1(2^-33) = 6.34.
"""
import os
import sys
from fractions import Fraction

import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import ndtri

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEG = 4
ROWS = 256


def row_interval(R):
    e = (30 - (R >> 3)) & 31
    s = R & 7
    lo = Fraction(2) ** (31 - e) * Fraction(8 + s, 8)
    hi = Fraction(2) ** (31 - e) * Fraction(9 + s, 8)
    return e, s, lo, hi


def remez(f, deg, iters=20, ngrid=8001):
    """minimax polynomial of f on [-1, 1] in the Chebyshev basis (float64); returns (coefficients, max abs error on the grid)"""
    n = deg + 2
    k = np.arange(n)
    x = -np.cos(np.pi * k / (n - 1))
    grid = np.linspace(-1.0, 1.0, ngrid)
    fg = f(grid)
    c = None
    for _ in range(iters):
        A = np.zeros((n, n))
        A[:, :deg + 1] = C.chebvander(x, deg)
        A[:, deg + 1] = (-1.0) ** k
        sol = np.linalg.solve(A, f(x))
        c = sol[:deg + 1]
        err = C.chebval(grid, c) - fg
        idx = [0] + [i for i in range(1, ngrid - 1) if (err[i] - err[i - 1]) * (err[i + 1] - err[i]) <= 0] + [ngrid - 1]
        ext = []
        for i in idx:
            if ext and np.sign(err[i]) == np.sign(err[ext[-1]]):
                if abs(err[i]) > abs(err[ext[-1]]):
                    ext[-1] = i
            else:
                ext.append(i)
        while len(ext) > n:
            if abs(err[ext[0]]) < abs(err[ext[-1]]):
                ext.pop(0)
            else:
                ext.pop()
        if len(ext) < n:
            break
        xn = grid[ext]
        if np.allclose(xn, x, rtol=0, atol=1e-12):
            break
        x = xn
    err = C.chebval(grid, c) - fg
    return c, float(np.abs(err).max())


def monomial_in_d(cheb, lo, hi):
    """exact change of variable: sum_j cheb_j T_j(u), u = (d - mid)/half  ->  coefficients of d^k (Fractions)"""
    mono_u = [Fraction(float(x)) for x in C.cheb2poly(cheb)]   # powers of u; cheb2poly is exact enough at degree 4 (small integers)
    mid, half = (lo + hi) / 2, (hi - lo) / 2
    # u = a*d + b
    a, b = 1 / half, -mid / half
    out = [Fraction(0)] * len(mono_u)
    # expand (a d + b)^k
    pw = [Fraction(1)]   # coefficients of (a d + b)^0
    for k, ck in enumerate(mono_u):
        for j, pj in enumerate(pw):
            out[j] += ck * pj
        nxt = [Fraction(0)] * (len(pw) + 1)
        for j, pj in enumerate(pw):
            nxt[j] += pj * b
            nxt[j + 1] += pj * a
        pw = nxt
    return out


def build():
    rows, worst, worst_k = [], 0.0, 0.0
    for R in range(ROWS):
        e, s, lo, hi = row_interval(R)
        flo, fhi = float(lo), float(hi)
        f = lambda u: -ndtri((0.5 * (flo + fhi) + 0.5 * (fhi - flo) * u) * 2.0 ** -33)
        cheb, _ = remez(f, DEG)
        co = [float(q) for q in monomial_in_d(cheb, lo, hi)]   # float(Fraction) rounds to nearest
        # the error of the ROUNDED row as the generator evaluates it (Horner in d; fma vs two roundings is ~1e-16, immaterial here)
        d = np.linspace(flo, fhi, 4001)
        q = np.full_like(d, co[DEG])
        for k in range(DEG - 1, -1, -1):
            q = q * d + co[k]
        z = -ndtri(d * 2.0 ** -33)
        err = float(np.abs(q - z).max())
        phi = float(np.exp(-0.5 * z.min() ** 2) / np.sqrt(2 * np.pi))
        worst, worst_k = max(worst, err), max(worst_k, err * phi)
        rows.append(co)
    return rows, worst, worst_k


def emit(path, guard, prefix, banner, rows, worst, worst_k):
    with open(path, "w") as f:
        f.write(f"/* {banner}\n * GENERATED by scripts/gen_icdf_table.py (Remez in float64 against scipy.special.ndtri, exact change of variable,\n"
                " * every coefficient rounded to nearest) -- do not edit.\n"
                " * Constant table of the noise specification bhip-philox-v4 (DESIGN.md section 4): row R = {c0, c1, c2, c3, c4},\n"
                " * |z| = c0 + d (c1 + d (c2 + d (c3 + d c4))),  d = (double)(2 (w mod 2^31) + 1),  R = (highword(d) >> 17) & 255.\n"
                f" * max |row(d) + Phi^-1(d 2^-33)| over all rows: {worst:.3e};  times the normal density: {worst_k:.3e} */\n")
        f.write(f"#ifndef {guard}\n#define {guard}\n")
        f.write(f"#define {prefix}ICDF_ROWS {ROWS}\n#define {prefix}ICDF_DEG {DEG}\n")
        f.write(f"#define {prefix}ICDF_MAXERR {worst:.3e}\n")
        f.write(f"#define {prefix}ICDF_INIT {{ \\\n")
        for R, co in enumerate(rows):
            f.write("  " + ", ".join(float(c).hex() for c in co) + ("," if R < ROWS - 1 else "") + " \\\n")
        f.write("}\n#endif\n")


if __name__ == "__main__":
    rows, worst, worst_k = build()
    print(f"rows {ROWS} degree {DEG}: max abs error {worst:.3e}, times density {worst_k:.3e} (2^-24 = {2.0 ** -24:.3e})")
    if "--dry" in sys.argv:
        sys.exit(0)
    emit(os.path.join(ROOT, "bridge.jl_amd", "csrc", "bhip_icdf_table.h"), "BHIP_ICDF_TABLE_H", "BHIP_",
         "bhip_icdf_table.h -- product copy (host, device, hipRTC)", rows, worst, worst_k)
    emit(os.path.join(ROOT, "oracle", "bo_icdf_table.h"), "BO_ICDF_TABLE_H", "BO_",
         "bo_icdf_table.h -- the CPU oracle's copy (TEST INFRASTRUCTURE; specification data)", rows, worst, worst_k)
    print("tables written")
