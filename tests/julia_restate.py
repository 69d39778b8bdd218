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


# ====================================================================================================================
# Round 3: the application loop around the path -- chained segments, gpupdate, LinearAppr, the index-based Heun guide,
# mcnext! and supplements/smoothing/smoothing.jl:99-213 -- restated a second time (the C oracle's twins are
# bo_gpupdate, bo_linearappr, bo_gp_hv_heuni, bo_mcnext, bo_smooth_mcmc, bo_smooth_adaptive).
# ====================================================================================================================
class Lorenz:
    """src/Models.jl:41-58: b, bderiv, sigma (SDiagonal)"""

    def __init__(self, theta, sigma):
        self.th, self.sg = np.asarray(theta, float), np.diag(np.asarray(sigma, float))

    def b(self, t, x):
        th = self.th
        return np.array([th[0] * (x[1] - x[0]), x[0] * (th[1] - x[2]) - x[1], x[0] * x[1] - th[2] * x[2]])

    def bderiv(self, t, x):
        th = self.th
        return np.array([[-th[0], th[0], 0.0], [th[1] - x[2], -1.0, -x[0]], [x[1], x[0], -th[2]]])

    def sig(self, t, x=None):
        return self.sg


def gpupdate(Hd, V, L, Sigma, v):
    """src/guip.jl:221-231 (matrix Sigma)"""
    Hd, V, L, Sigma, v = (np.atleast_2d(np.asarray(Hd, float)), np.atleast_1d(np.asarray(V, float)), np.atleast_2d(np.asarray(L, float)),
                          np.atleast_2d(np.asarray(Sigma, float)), np.atleast_1d(np.asarray(v, float)))
    if np.all(np.diag(Hd) == np.inf):
        A = L.T @ np.linalg.inv(Sigma) @ L
        return np.linalg.inv(A), np.linalg.solve(A, L.T @ np.linalg.inv(Sigma) @ v)
    Z = np.eye(Hd.shape[0]) - Hd @ L.T @ np.linalg.inv(Sigma @ np.eye(Sigma.shape[0]) + L @ Hd @ L.T) @ L
    return Z @ Hd, Z @ Hd @ L.T @ np.linalg.inv(Sigma) @ v + Z @ V


class LinearAppr:
    """src/linpro.jl:181-204: linearappr(Y, P) / linearappr!(Pt, Y, P); _b, B, beta, a by grid INDEX"""

    def __init__(self, tt, yy, P):
        self.assign(tt, yy, P)

    def assign(self, tt, yy, P):          # linearappr!
        self.tt = np.asarray(tt, float).copy()
        self.xx = [np.asarray(x, float).copy() for x in yy]
        self.Bs = [P.bderiv(t, x) for t, x in zip(self.tt, self.xx)]
        self.bs = [P.b(t, x) for t, x in zip(self.tt, self.xx)]
        self.Sg = [P.sig(t, x) for t, x in zip(self.tt, self.xx)]
        return self

    def b_i(self, i, x):                  # _b((i,s), x, P::LinearAppr) = P.B[i]*(x - P.xx[i]) + P.b[i]
        return self.Bs[i] @ (x - self.xx[i]) + self.bs[i]

    def a_i(self, i):                     # a((i,s), P) = outer(P.Sigma[i])
        return self.Sg[i] @ self.Sg[i].T


def kerneli_heun(f, i, y, dt):
    """src/ode.jl:98-102 with `i` = the loop index of solvebackwardi! (as committed the name is unbound):
    k1 = f((i,t), y); k2 = f((i+1,t+dt), y + dt*k1); y + dt/2*(k1 + k2)"""
    k1 = f(i, y)
    k2 = f(i + 1, y + dt * k1)
    return y + dt / 2 * (k1 + k2)


class GuidedBridgeLA:
    """GuidedBridge(tt, P, Pt::LinearAppr, v, h)  src/guip.jl:181-189 (solvebackwardi!, src/ode.jl:104-113), then the
    methods of src/guip.jl:192-194 and the end-point rule of src/euler.jl:241-242"""

    def __init__(self, tt, P, Pt, v, h):
        self.tt, self.P, self.Pt = np.asarray(tt, float), P, Pt
        N = len(self.tt)
        K, V = np.atleast_2d(np.asarray(h, float)), np.atleast_1d(np.asarray(v, float))
        self.Hd, self.V = [None] * N, [None] * N
        self.Hd[-1], self.V[-1] = K, V
        for i in range(N - 2, -1, -1):
            dt = self.tt[i] - self.tt[i + 1]
            K = kerneli_heun(lambda j, Y: Pt.Bs[j] @ Y + Y @ Pt.Bs[j].T - Pt.a_i(j), i, K, dt)
            self.Hd[i] = K
        for i in range(N - 2, -1, -1):
            dt = self.tt[i] - self.tt[i + 1]
            V = kerneli_heun(lambda j, y: Pt.b_i(j, y), i, V, dt)
            self.V[i] = V

    def r(self, i, x):
        return np.linalg.solve(self.Hd[i], self.V[i] - x)

    def _b(self, i, x):
        t = self.tt[i]
        return self.P.b(t, x) + a_of(self.P.sig(t, x)) @ np.linalg.solve(self.Hd[i], self.V[i] - x)

    def btilde(self, i, x):
        return self.Pt.b_i(i, x)

    def endpoint(self, y):
        return self.V[-1] if np.abs(self.Hd[-1]).sum() < np.finfo(float).eps else y


def llikelihood_indexed(X, Po, skip=0):
    """src/guip.jl:429-438, constant-diffusivity form, with the auxiliary drift taken by grid index (btilde((i,s), x, P))"""
    tt = Po.tt
    som = 0.0
    for i in range(len(tt) - 1 - skip):
        x = X[i]
        som += np.dot(Po.P.b(tt[i], x) - Po.btilde(i, x), Po.r(i, x)) * (tt[i + 1] - tt[i])
    return som


def mcstart(yy):
    """src/mclog.jl:22"""
    yy = np.asarray(yy, float)
    return [np.zeros_like(yy), np.zeros(yy.shape + (yy.shape[1],)), 0]


def mcnext_(mc, xx):
    """mcnext!  src/mclog.jl:48-56"""
    m, m2, n = mc
    for i in range(len(m2)):
        delta = xx[i] - m[i]
        m[i] = m[i] + delta / (n + 1)
        m2[i] = m2[i] + np.outer(delta, xx[i] - m[i])
    mc[2] = n + 1
    return mc


def smooth(pi0, tts, P, Po, ll_of, noise, iterations, w_new, w_old, L=None, Sigma=None, obs=None, HT=None, vT=None,
           adaptit=0, adaptmax=0, smoothmean=False, hwindow=20, skip=0):
    """supplements/smoothing/smoothing.jl:95-213 for one chain: initialisation (:99-106) and `smooth` (:110-213).

    pi0 = (mu, Sigma) of Gaussian(v, Hermitian(Hdiamond)); Po: the m proposals; ll_of(X, Po_i): llikelihood(LeftRule(), X, Po_i);
    noise: .wiener(i, it) = sample(tt_i, Wiener) for segment i at iteration it (it = 0: the initialisation), .randn(it) the
    d normals of rand(pi0), .rand(it) the uniform; w_new[it-1] = sqrt(rho_), w_old[it-1] = sqrt(1 - rho_) (the script draws
    rho_ = exp(-alpha*randexp())); adaptation (:130-160) for LinearAppr proposals when adaptit > 0."""
    m = len(Po)
    mu, Sig0 = np.asarray(pi0[0], float), np.asarray(pi0[1], float)
    # initialize  (:99-106)
    XX, WW = [None] * m, [None] * m
    y = mu
    for i in range(m):
        WW[i] = noise.wiener(i, 0)
        XX[i] = solve_euler(y, WW[i], Po[i])
        y = XX[i][-1]                                     # bridge! returns yy[N]   src/euler.jl:267
    mcstate = [mcstart(XX[i]) for i in range(m)]
    acc, y0, newblock = 0, mu, False
    for it in range(1, iterations + 1):
        doaccept = False
        if adaptit and it < adaptmax and it % adaptit == 0:      # adaptive smoothing  (:130-160)
            H, v = np.asarray(HT, float), np.asarray(vT, float)
            for i in range(m - 1, -1, -1):
                xx = mcstate[i][0]
                if smoothmean:
                    Y = [np.mean(xx[max(0, j - hwindow):min(len(xx), j + hwindow + 1)], axis=0) for j in range(len(xx))]
                else:
                    Y = [x.copy() for x in xx]
                Po[i].Pt.assign(tts[i], Y, P)                    # linearappr!(Po[i].Pt, Y, P)
                Po[i] = GuidedBridgeLA(tts[i], P, Po[i].Pt, v, H)
                H, v = gpupdate(Po[i].Hd[0], Po[i].V[0], L, Sigma, obs[i])
            mu, Sig0 = v, np.triu(H) + np.triu(H, 1).T           # Gaussian(v, Hermitian(H))
            newblock = True
            if it == adaptit:
                doaccept = True
        rho_new, rho_old = w_new[it - 1], w_old[it - 1]
        if newblock:
            y0o = y0
        else:
            C = np.linalg.cholesky(Sig0)                         # rand(pi0) = mu + chol(Sigma)'*randn   src/gaussian.jl:54
            y0o = mu + rho_new * ((mu + C @ noise.randn(it)) - mu) + rho_old * (y0 - mu)
        y = y0o
        XXo, WWo = [None] * m, [None] * m
        for i in range(m):
            WWo[i] = rho_new * noise.wiener(i, it) + rho_old * WW[i]
            XXo[i] = solve_euler(y, WWo[i], Po[i])
            y = XXo[i][-1]
        ll = 0.0
        for i in range(m):
            ll += ll_of(XXo[i], Po[i], skip) - ll_of(XX[i], Po[i], skip)
        if doaccept or noise.rand(it) < np.exp(ll):
            acc += 1
            y0 = y0o
            XX, WW = XXo, WWo
            newblock = False
        for i in range(m):
            mcstate[i] = mcnext_(mcstate[i], XX[i])
    lls = [ll_of(XX[i], Po[i], skip) for i in range(m)]
    return dict(X=np.stack(XX), W=np.stack(WW), y0=y0, acc=acc, mean=np.stack([s[0] for s in mcstate]), m2=np.stack([s[1] for s in mcstate]),
                ll=np.array(lls), mu=mu, H=Sig0, Po=Po)
