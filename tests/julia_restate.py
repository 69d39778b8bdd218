"""A SECOND, independent restatement of the guided path of Bridge.jl -- numpy, generic linear algebra, written straight
from the Julia sources (NOT from oracle/bridge_oracle.c and not sharing a line with it).  TEST INFRASTRUCTURE.

Purpose ("weak" 1): the C oracle and the product were written by the same hands from the same reading of
the reference; a transcription slip shared by both would pass every GPU-vs-oracle `==`.  This file re-derives the
guide ODEs, `_b`, `r`, `solve!(Euler(), ...)` and `llikelihood(LeftRule(), ...)` a second time, with numpy's generic
`@`, `np.linalg.inv`, `np.linalg.solve`, `np.outer`, so that the two restatements can only agree if both follow the
Julia.  Agreement is to rounding (1e-12 relative: generic LAPACK-style inverses vs StaticArrays' closed forms), not
bit-level.

Every function cites the reference lines it transliterates (relative to /root/reference).
"""
import numpy as np


# ------------------------------------------------------------------ processes: Bridge.b / sigma / a methods
class FitzhughDiffusion:
    """project_partialbridge/partialbridge_fitzhugh.jl:36-46"""

    def __init__(self, eps, s, gamma, beta, sigma):
        self.eps, self.s, self.gamma, self.beta, self.sigma = eps, s, gamma, beta, sigma

    def b(self, t, x):
        return np.array([(x[0] - x[1] - x[0] * x[0] * x[0] + self.s) / self.eps, self.gamma * x[0] - x[1] + self.beta])

    def sig(self, t, x):
        return np.array([[0.0], [self.sigma]])


class FitzhughAuxEnd:
    """FitzhughDiffusionAux, aux_choice "linearised_end": partialbridge_fitzhugh.jl:58-62, 99-100, 107-116"""

    def __init__(self, P, v):
        self.P, self.v = P, v

    def B(self, t):
        P, v = self.P, self.v
        return np.array([[1 / P.eps - 3 * v ** 2 / P.eps, -1 / P.eps], [P.gamma, -1.0]])

    def beta(self, t):
        P, v = self.P, self.v
        return np.array([P.s / P.eps + 2 * v ** 3 / P.eps, P.beta])

    def sig(self, t):
        return np.array([[0.0], [self.P.sigma]])

    def b(self, t, x):            # Bridge.b(t, x, P::FitzhughDiffusionAux) = B(t,P)*x + beta(t,P)
        return self.B(t) @ x + self.beta(t)


class LinPro:
    """src/linpro.jl:65-87: b = B*(x - mu), sigma, a = sigma*sigma'"""

    def __init__(self, B, mu, sigma):
        self.Bm, self.mu, self.sg = np.atleast_2d(np.asarray(B, float)), np.atleast_1d(np.asarray(mu, float)), np.atleast_2d(np.asarray(sigma, float))

    def b(self, t, x):
        return self.Bm @ (x - self.mu)

    def sig(self, t, x=None):
        return self.sg

    def B(self, t):
        return self.Bm

    def beta(self, t):            # src/linpro.jl: beta(t, P::LinPro) = -P.B*P.mu
        return -self.Bm @ self.mu


def a_of(sig):
    """a = outer(sigma) = sigma*sigma'   src/types.jl:32, src/misc.jl:63"""
    return sig @ sig.T


# ------------------------------------------------------------------ src/ode.jl:44-49
def kernelr3(f, t, y, dt):
    k1 = f(t, y)
    k2 = f(t + 1 / 2 * dt, y + 1 / 2 * dt * k1)
    k3 = f(t + 3 / 4 * dt, y + 3 / 4 * dt * k2)
    return y + dt * (2 / 9 * k1 + 1 / 3 * k2 + 4 / 9 * k3)


# ------------------------------------------------------------------ src/partialbridge.jl:1-22
def partialbridgeode(tt, L, Sigma, Pt):
    N = len(tt)
    L = np.atleast_2d(np.asarray(L, float))
    Sigma = np.atleast_2d(np.asarray(Sigma, float))
    m = L.shape[0]
    Lt, Mt, mut = [None] * N, [None] * N, [None] * N
    Lt[-1] = L
    Mt[-1] = np.linalg.inv(Sigma)
    Mp = Sigma
    mut[-1] = mu = np.zeros(m)
    for i in range(N - 2, -1, -1):
        dt = tt[i] - tt[i + 1]
        L = kernelr3(lambda t, y: -y @ Pt.B(t), tt[i + 1], L, dt)
        Mp = kernelr3(lambda t, y: -a_of(L @ Pt.sig(t)), tt[i + 1], Mp, dt)       # uses the L already advanced
        mu = kernelr3(lambda t, y: -L @ Pt.beta(t), tt[i + 1], mu, dt)
        Lt[i], Mt[i], mut[i] = L, np.linalg.inv(Mp), mu
    return Lt, Mt, mut


# ------------------------------------------------------------------ src/gode.jl:2-3,13,21 + src/ode.jl:88-97
def gp_hv(tt, Pt, v, hT=None):
    N, d = len(tt), len(v)
    K = np.zeros((d, d)) if hT is None else np.atleast_2d(np.asarray(hT, float))
    V = np.asarray(v, float)
    Hd, Vs = [None] * N, [None] * N
    Hd[-1], Vs[-1] = K, V
    for i in range(N - 2, -1, -1):
        dt = tt[i] - tt[i + 1]
        K = kernelr3(lambda t, y: Pt.B(t) @ y + y @ Pt.B(t).T - a_of(Pt.sig(t)), tt[i + 1], K, dt)
        V = kernelr3(lambda t, y: Pt.B(t) @ y + Pt.beta(t), tt[i + 1], V, dt)
        Hd[i], Vs[i] = K, V
    return Hd, Vs


# ------------------------------------------------------------------ src/partialbridgenuH.jl:1-8, 21-55
def partialbridge_nuH(tt, L, Sigma, v, eps, Pt):
    L = np.atleast_2d(np.asarray(L, float))
    Sigma = np.atleast_2d(np.asarray(Sigma, float))
    v = np.atleast_1d(np.asarray(v, float))
    d = L.shape[1]
    H = L.T @ np.linalg.inv(Sigma) @ L + eps * np.eye(d)
    Hp = np.linalg.inv(H)
    nu = Hp @ L.T @ np.linalg.inv(Sigma) @ v
    N = len(tt)
    nut, Ht = [None] * N, [None] * N
    Ht[-1] = H = np.linalg.inv(Hp)
    nut[-1] = nu
    for i in range(N - 2, -1, -1):
        dt = tt[i] - tt[i + 1]
        Hp = kernelr3(lambda t, y: Pt.B(t) @ y + (Pt.B(t) @ y).T - a_of(Pt.sig(t)), tt[i + 1], Hp, dt)
        nu = kernelr3(lambda t, y: Pt.B(t) @ y + Pt.beta(t), tt[i + 1], nu, dt)
        nut[i] = nu
        Ht[i] = H = np.linalg.inv(Hp)
    return nut, Ht


# ------------------------------------------------------------------ proposals: _b, r
class PartialBridge:
    """src/partialbridge.jl:33-58"""

    def __init__(self, tt, P, Pt, L, v, Sigma):
        self.tt, self.P, self.Pt, self.v = np.asarray(tt, float), P, Pt, np.atleast_1d(np.asarray(v, float))
        self.L, self.M, self.mu = partialbridgeode(self.tt, L, Sigma, Pt)

    def r(self, i, x):
        return self.L[i].T @ self.M[i] @ (self.v - self.mu[i] - self.L[i] @ x)

    def _b(self, i, x):
        t = self.tt[i]
        return self.P.b(t, x) + a_of(self.P.sig(t, x)) @ self.L[i].T @ self.M[i] @ (self.v - self.mu[i] - self.L[i] @ x)

    def endpoint(self, y):
        return y


class GuidedBridge:
    """src/guip.jl:165-194, endpoint src/euler.jl:241-242"""

    def __init__(self, tt, P, Pt, v, hT=None):
        self.tt, self.P, self.Pt = np.asarray(tt, float), P, Pt
        self.Hd, self.V = gp_hv(self.tt, Pt, np.atleast_1d(np.asarray(v, float)), hT)

    def r(self, i, x):
        return np.linalg.solve(self.Hd[i], self.V[i] - x)

    def _b(self, i, x):
        t = self.tt[i]
        return self.P.b(t, x) + a_of(self.P.sig(t, x)) @ np.linalg.solve(self.Hd[i], self.V[i] - x)

    def endpoint(self, y):
        return self.V[-1] if np.abs(self.Hd[-1]).sum() < np.finfo(float).eps else y   # norm(A, 1) of a matrix = the entrywise 1-norm (LinearAlgebra / StaticArrays; opnorm is the induced one)


class PartialBridgeNuH:
    """src/partialbridgenuH.jl:122-162"""

    def __init__(self, tt, P, Pt, L, v, eps, Sigma):
        self.tt, self.P, self.Pt = np.asarray(tt, float), P, Pt
        self.nu, self.H = partialbridge_nuH(self.tt, L, Sigma, v, eps, Pt)

    def r(self, i, x):
        return self.H[i] @ (self.nu[i] - x)

    def _b(self, i, x):
        t = self.tt[i]
        return self.P.b(t, x) + a_of(self.P.sig(t, x)) @ (self.H[i] @ (self.nu[i] - x))

    def endpoint(self, y):
        return y


# ------------------------------------------------------------------ src/euler.jl:247-268
def solve_euler(u, W, Po):
    """W: [N, m'] driving Wiener path sampled on Po.tt; returns X [N, d]"""
    tt = Po.tt
    N = len(tt)
    y = np.asarray(u, float)
    X = np.zeros((N, len(y)))
    for i in range(N - 1):
        X[i] = y
        y = y + Po._b(i, y) * (tt[i + 1] - tt[i]) + Po.P.sig(tt[i], y) @ (W[i + 1] - W[i])
    X[N - 1] = Po.endpoint(y)
    return X


# ------------------------------------------------------------------ src/partialbridge.jl:67-77 (= guip.jl:429-438, partialbridgenuH.jl:171-181)
def llikelihood(X, Po, skip=0):
    tt = Po.tt
    som = 0.0
    for i in range(len(tt) - 1 - skip):
        s, x = tt[i], X[i]
        r = Po.r(i, x)
        som += np.dot(Po.P.b(s, x) - Po.Pt.b(s, x), r) * (tt[i + 1] - tt[i])
    return som
