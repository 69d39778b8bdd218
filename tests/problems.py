"""Shared problem definitions: each case builds the SAME proposal twice -- once through the CPU
oracle (tests/oracle.py) and once through the product's host mirror (bridgehip) -- from the
reference's own workload definitions.  Used by the CPU host-logic tests and the GPU parity tests.
"""
import math

import numpy as np

import oracle as o


def tau_grid(T, N):
    """tt = tau(T).(0:dt:T), tau(s) = s*(2 - s/T)   partialbridge_fitzhugh.jl:13-14"""
    s = np.linspace(0.0, T, N)
    return s * (2 - s / T)


class Case:
    def __init__(self, name, tt, x0, model, par, aux, apar, kind, d, mp, m=None, L=None, v=None, Sigma=None,
                 eps=None, hT=None, exact=True, rho=0.9):
        self.__dict__.update(locals())
        self.m = m if m is not None else d
        # sin / cos in the drift: since the fdlibm-form restatement shared by oracle, host and kernels (round 2) these cases are
        # bit-exact like the others; the goldens frozen BEFORE that (v1, v2) went through libm's sin and differ in the last place
        self.trig = model in (o.MODEL_NCLAR, o.MODEL_INTDIFF, o.MODEL_PENDULUM)
        self.tt = np.ascontiguousarray(tt, dtype=np.float64)
        self.x0 = np.atleast_1d(np.asarray(x0, dtype=np.float64))

    # ---- oracle side
    def oracle_guide(self):
        c = self
        if c.kind == o.GUIDE_HV:
            Hd, V = o.gp_hv(c.tt, c.d, c.mp, c.aux, c.apar, c.v, c.hT)
            return dict(Hd=Hd, V=V)
        if c.kind == o.GUIDE_LMMU:
            Lt, Mt, mut = o.partialbridge_ode(c.tt, c.d, c.mp, c.m, c.aux, c.apar, c.L, c.Sigma)
            return dict(L=Lt, M=Mt, mu=mut)
        nut, Ht, C = o.partialbridge_nuH(c.tt, c.d, c.mp, c.m, c.aux, c.apar, c.L, c.v, c.eps, c.Sigma,
                                         inplace=c.kind == o.GUIDE_NUH_INPLACE)
        return dict(nu=nut, H=Ht, C=C)

    def oracle_proposal(self, g=None):
        c = self
        if c.kind == o.GUIDE_NONE:
            return None
        g = g or c.oracle_guide()
        if c.kind == o.GUIDE_HV:
            return o.proposal_hv(c.tt, c.d, c.mp, c.model, c.par, c.aux, c.apar, g["Hd"], g["V"])
        if c.kind == o.GUIDE_LMMU:
            return o.proposal_lmmu(c.tt, c.d, c.mp, c.m, c.model, c.par, c.aux, c.apar, g["L"], g["M"], g["mu"], c.v)
        return o.proposal_nuh(c.tt, c.d, c.mp, c.model, c.par, c.aux, c.apar, g["nu"], g["H"],
                              inplace=c.kind == o.GUIDE_NUH_INPLACE)

    # ---- product side
    def bh_process(self, bh):
        c, p = self, self.par
        return {o.MODEL_OU: lambda: bh.OrnsteinUhlenbeck(*p),
                o.MODEL_LINPRO: lambda: bh.LinPro(o.uncm(p[:c.d * c.d], c.d, c.d), p[c.d * c.d:c.d * c.d + c.d],
                                                  o.uncm(p[c.d * c.d + c.d:], c.d, c.d)),
                o.MODEL_FHN: lambda: bh.FitzhughDiffusion(*p),
                o.MODEL_NCLAR: lambda: bh.NclarDiffusion(*p),
                o.MODEL_INTDIFF: lambda: bh.IntegratedDiffusion(*p),
                o.MODEL_LORENZ: lambda: bh.Lorenz(p[:3], p[3:]),
                o.MODEL_FHN2: lambda: bh.FitzHughNagumo(*p),
                o.MODEL_PENDULUM: lambda: bh.Pendulum(*p),
                o.MODEL_WIENER: lambda: bh.Wiener(c.d)}[c.model]()

    def bh_aux(self, bh):
        c, ap = self, np.asarray(self.apar, dtype=np.float64)
        d, mp = c.d, c.mp
        if c.aux == o.AUX_LINPRO:
            return bh.LinPro(o.uncm(ap[:d * d], d, d), ap[d * d:d * d + d], o.uncm(ap[d * d + d:], d, d))
        if c.aux == o.AUX_AFFINE:
            return bh.AffineAux(o.uncm(ap[:d * d], d, d), ap[d * d:d * d + d], o.uncm(ap[d * d + d:], d, mp))
        return bh.FitzhughDiffusionAuxStartEnd(*ap)

    def bh_proposal(self, bh, ctx=None):
        c = self
        P = c.bh_process(bh)
        if c.kind == o.GUIDE_NONE:
            return bh.PlainProcess(c.tt, P, ctx=ctx)
        Pt = c.bh_aux(bh)
        if c.kind == o.GUIDE_HV:
            return bh.GuidedBridge(c.tt, P, Pt, c.v, c.hT, ctx=ctx)
        if c.kind == o.GUIDE_LMMU:
            return bh.PartialBridge(c.tt, P, Pt, c.L, c.v, c.Sigma, ctx=ctx)
        if c.kind == o.GUIDE_NUH:
            return bh.PartialBridgeNuH(c.tt, P, Pt, c.L, c.v, c.eps, c.Sigma, ctx=ctx)
        return bh.PartialBridgeInplace(c.tt, P, Pt, c.L, c.v, c.eps, c.Sigma, ctx=ctx)


def fhn_aux_end(eps, s, gamma, beta, sigma, v):
    """Bridge.B / Bridge.beta of FitzhughDiffusionAux "linearised_end" (partialbridge_fitzhugh.jl:99-100)"""
    B = [[1 / eps - 3 * (v * v) / eps, -1 / eps], [gamma, -1.0]]
    be = [s / eps + 2 * (v * v * v) / eps, beta]
    return o.affine_par(B, be, [[0.0], [sigma]])


def cases(N=201):
    cs = []
    # ---- C2: 1-d OU target (LinPro(-0.8, 0, sqrt(.7))), aux LinPro(-0.8, 0.2, sqrt(.7)), GuidedBridge (test/guip.jl:117-120,248)
    a, beta = 0.7, 0.8
    par = o.linpro_par([[-beta]], [0.0], [[math.sqrt(a)]])
    apar = o.linpro_par([[-beta]], [0.2], [[math.sqrt(a)]])
    cs.append(Case("ou_guidedbridge", tau_grid(2.0, N), [0.5], o.MODEL_LINPRO, par, o.AUX_LINPRO, apar, o.GUIDE_HV, 1, 1, v=[0.1]))
    cs.append(Case("ou_guidedbridge_free_end", np.linspace(0, 2.0, N), [0.5], o.MODEL_LINPRO, par, o.AUX_LINPRO, apar,
                   o.GUIDE_HV, 1, 1, v=[0.1], hT=[[0.05]]))
    cs.append(Case("ouproc_nuh", np.linspace(0, 1.0, N), [0.1], o.MODEL_OU, [2.0, 1.0], o.AUX_AFFINE,
                   o.affine_par([[-1.5]], [0.1], [[1.0]]), o.GUIDE_NUH, 1, 1, m=1, L=[[1.0]], v=[0.3], Sigma=[[0.01]], eps=0.0))
    # ---- C3: FitzHugh-Nagumo partial bridge (partialbridge_fitzhugh.jl:31-33,48-50,88-101)
    fpar = [0.1, 0.0, 1.5, 0.8, 0.3]
    x0 = [-0.5, -0.6]
    L = [[1.0, 0.0]]
    Sg = [[1e-10]]
    for nm, v in (("first", -1.0), ("extreme", 1.1)):
        ap = fhn_aux_end(*fpar, v)
        cs.append(Case(f"fhn_partialbridge_{nm}", tau_grid(2.0, N), x0, o.MODEL_FHN, fpar, o.AUX_AFFINE, ap,
                       o.GUIDE_LMMU, 2, 1, m=1, L=L, v=[v], Sigma=Sg, rho=0.0 if nm == "first" else 0.9))
    ap = fhn_aux_end(*fpar, -1.0)
    # (nu,H) parametrisation: H+ grows like exp(40 (T-t)) for this contracting auxiliary, so inv(H+) is only
    # well defined on a short horizon (on T = 2 the reference's formulas give H = Inf at some grid points)
    cs.append(Case("fhn_nuh", tau_grid(0.25, N), x0, o.MODEL_FHN, fpar, o.AUX_AFFINE, ap, o.GUIDE_NUH, 2, 1, m=1, L=L,
                   v=[-1.0], Sigma=[[1e-2]], eps=1e-3))
    cs.append(Case("fhn_inplace", tau_grid(0.25, N), x0, o.MODEL_FHN, fpar, o.AUX_AFFINE, ap, o.GUIDE_NUH_INPLACE, 2, 1, m=1,
                   L=L, v=[-1.0], Sigma=[[1e-2]], eps=1e-3))
    cs.append(Case("fhn_startend", tau_grid(2.0, N), x0, o.MODEL_FHN, fpar, o.AUX_FHN_STARTEND,
                   fpar + [0.0, x0[0], 2.0, 1.1], o.GUIDE_LMMU, 2, 1, m=1, L=L, v=[1.1], Sigma=Sg, rho=0.98))
    # ---- NCLAR 3-d (partialbridge_nclar.jl:13,43-45,52-86)
    npar = [6.0, 2 * math.pi, 1.0]
    nap = o.affine_par([[0, 1, 0], [0, 0, 1], [0, 0, 0]], [0, 0, 0], [[0.0], [0.0], [1.0]])
    cs.append(Case("nclar_firstcomponent", tau_grid(0.5, N), [0, 0, 0], o.MODEL_NCLAR, npar, o.AUX_AFFINE, nap, o.GUIDE_LMMU,
                   3, 1, m=1, L=[[1.0, 0, 0]], v=[5 / 128], Sigma=[[1e-10]], rho=0.95))
    cs.append(Case("nclar_full", tau_grid(0.5, N), [0, 0, 0], o.MODEL_NCLAR, npar, o.AUX_AFFINE, nap, o.GUIDE_LMMU,
                   3, 1, m=3, L=np.eye(3), v=[5 / 128, 3 / 8, 2], Sigma=1e-10 * np.eye(3), rho=0.85))
    # ---- IntegratedDiffusion (test/partialparam.jl, test/partialbridge.jl)
    iap = o.affine_par([[0.0, 1.0], [0.0, -1.0]], [0.0, 0.5], [[0.0], [0.7]])
    tti = np.linspace(0, 1.5, N)
    cs.append(Case("intdiff_partialbridge", tti, [2.0, 1.0], o.MODEL_INTDIFF, [0.7], o.AUX_AFFINE, iap, o.GUIDE_LMMU, 2, 1,
                   m=1, L=[[1.0, 0.0]], v=[2.5], Sigma=[[0.1]]))
    cs.append(Case("intdiff_nuh", tti, [2.0, 1.0], o.MODEL_INTDIFF, [0.7], o.AUX_AFFINE, iap, o.GUIDE_NUH, 2, 1,
                   m=1, L=[[1.0, 0.0]], v=[2.5], Sigma=[[0.1]], eps=1e-2))
    # ---- 2-d / 3-d LinPro GuidedBridge (test/linpro.jl:8-17 matrices)
    B2 = np.array([[-1, 0.1], [-0.2, -1]])
    s2 = 2 * np.array([[-0.212887, 0.0687025], [0.193157, 0.388997]])
    p2 = o.linpro_par(B2, [0.02, 0.03], s2)
    a2 = o.linpro_par(B2 * 0.9, [0.0, 0.0], s2)
    cs.append(Case("linpro2_guidedbridge", np.linspace(0, 1.0, N), [1.0, 0.0], o.MODEL_LINPRO, p2, o.AUX_LINPRO, a2,
                   o.GUIDE_HV, 2, 2, v=[0.5, 0.0]))
    rng = np.random.default_rng(7)
    B3 = -np.eye(3) + 0.2 * rng.standard_normal((3, 3))
    s3 = 0.5 * np.eye(3) + 0.1 * rng.standard_normal((3, 3))
    p3 = o.linpro_par(B3, [0.1, -0.1, 0.0], s3)
    a3 = o.linpro_par(-np.eye(3), [0.0, 0.0, 0.0], s3)
    cs.append(Case("linpro3_guidedbridge", np.linspace(0, 1.0, N), [0.2, 0.0, -0.1], o.MODEL_LINPRO, p3, o.AUX_LINPRO, a3,
                   o.GUIDE_HV, 3, 3, v=[0.5, 0.1, 0.0]))
    cs.append(Case("linpro3_partial_m2", np.linspace(0, 1.0, N), [0.2, 0.0, -0.1], o.MODEL_LINPRO, p3, o.AUX_LINPRO, a3,
                   o.GUIDE_LMMU, 3, 3, m=2, L=[[1.0, 0, 0], [0, 1.0, 0.5]], v=[0.5, 0.1], Sigma=0.01 * np.eye(2)))
    # ---- Models.FitzHughNagumo with diagonal 2-d noise, fully observed, nuH
    f2 = [0.1, 0.0, 1.5, 0.8, 0.25, 0.2]
    f2a = o.affine_par([[1 / 0.1 - 3 / 0.1, -1 / 0.1], [1.5, -1.0]], [0.0 / 0.1 + 2 / 0.1, 0.8], [[0.25, 0.0], [0.0, 0.2]])
    cs.append(Case("fhn2_nuh_full", tau_grid(0.25, N), [-0.5, -0.6], o.MODEL_FHN2, f2, o.AUX_AFFINE, f2a, o.GUIDE_NUH, 2, 2,
                   m=2, L=np.eye(2), v=[-1.0, -0.5], Sigma=1e-6 * np.eye(2), eps=0.0))
    # ---- Pendulum partial bridge (supplements/smoothing model, src/Models.jl:69-88)
    pa = o.affine_par([[0.0, 1.0], [0.0, 0.0]], [0.0, 0.0], [[0.0], [0.5]])
    cs.append(Case("pendulum_partialbridge", np.linspace(0, 1.0, N), [1.0, 0.5], o.MODEL_PENDULUM, [4.0, 0.5], o.AUX_AFFINE, pa,
                   o.GUIDE_LMMU, 2, 1, m=1, L=[[1.0, 0.0]], v=[0.8], Sigma=[[0.01]]))
    return cs


def linpro_big_case(d=32, N=201, kind=None):
    """config C5: LinPro d=32, B = -I + 0.1 G, G ~ N(0,1)/sqrt(d) (seed 5), mu = 0,
    sigma = 0.5 I + 0.05 G2 (dense a), auxiliary LinPro with B~ = -I and the same sigma; T = 1, u = 0, v = 0.5*1"""
    rng = np.random.default_rng(5)
    G = rng.standard_normal((d, d)) / math.sqrt(d)
    G2 = rng.standard_normal((d, d)) / math.sqrt(d)
    B = -np.eye(d) + 0.1 * G
    sig = 0.5 * np.eye(d) + 0.05 * G2
    par = o.linpro_par(B, np.zeros(d), sig)
    apar = o.linpro_par(-np.eye(d), np.zeros(d), sig)
    kind = o.GUIDE_HV if kind is None else kind
    if kind == o.GUIDE_HV:
        return Case(f"linpro{d}_guidedbridge", np.linspace(0, 1.0, N), np.zeros(d), o.MODEL_LINPRO, par, o.AUX_LINPRO, apar,
                    o.GUIDE_HV, d, d, v=0.5 * np.ones(d), exact=False)
    return Case(f"linpro{d}_nuh", np.linspace(0, 1.0, N), np.zeros(d), o.MODEL_LINPRO, par, o.AUX_LINPRO, apar,
                o.GUIDE_NUH, d, d, m=d, L=np.eye(d), v=0.5 * np.ones(d), Sigma=0.01 * np.eye(d), eps=0.0, exact=False)


def forward_cases(N=201):
    """unguided Euler-Maruyama (config C1 and test/euler.jl)"""
    return [
        Case("ou_readme", np.arange(0, N) * 0.01, [0.1], o.MODEL_OU, [2.0, 1.0], o.AUX_AFFINE, [], o.GUIDE_NONE, 1, 1),
        Case("lorenz", np.linspace(0, 1.0, N), [1.0, 0.0, 0.0], o.MODEL_LORENZ, [10.0, 28.0, 8 / 3, 3.0, 3.0, 3.0], o.AUX_AFFINE, [],
             o.GUIDE_NONE, 3, 3),
        Case("fhn_forward", np.linspace(0, 1.0, N), [-0.5, -0.6], o.MODEL_FHN, [0.1, 0.0, 1.5, 0.8, 0.3], o.AUX_AFFINE, [],
             o.GUIDE_NONE, 2, 1),
        Case("nclar_forward", np.linspace(0, 0.5, N), [0.0, 0.0, 0.0], o.MODEL_NCLAR, [6.0, 2 * math.pi, 1.0], o.AUX_AFFINE, [],
             o.GUIDE_NONE, 3, 1),
    ]
