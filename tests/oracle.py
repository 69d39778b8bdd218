"""ctypes binding of the CPU oracle (oracle/libbridge_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package (bridge.jl_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ODIR = os.path.join(os.path.dirname(_HERE), "oracle")
# BRIDGE_ORACLE_SO: another build of the same source (the sanitizer leg, tests/test_oracle_sanitizers.py: `make -C oracle san`)
_SO = os.environ.get("BRIDGE_ORACLE_SO") or os.path.join(_ODIR, "libbridge_oracle.so")

MODEL_WIENER, MODEL_OU, MODEL_LINPRO, MODEL_FHN, MODEL_NCLAR, MODEL_INTDIFF, MODEL_LORENZ, MODEL_FHN2, MODEL_PENDULUM = range(9)
MODEL_SDIFF1, MODEL_SDIFF2 = 9, 10        # state-dependent sigma (oracle-side stand-ins for hipRTC user processes)
MODEL_LORENZ96 = 11                       # Lorenz-96 with a dense constant sigma: stand-in for a component-wise user drift at d > 3
AUX_AFFINE, AUX_LINPRO, AUX_FHN_STARTEND = range(3)
GUIDE_NONE, GUIDE_HV, GUIDE_LMMU, GUIDE_NUH, GUIDE_NUH_INPLACE = range(5)

dp = C.POINTER(C.c_double)


def _build():
    src = os.path.join(_ODIR, "bridge_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ODIR, "-s"] + (["san"] if _SO.endswith("_san.so") else []))


def load():
    _build()
    lib = C.CDLL(_SO)
    lib.bo_log.restype = C.c_double
    lib.bo_icdf_normal.restype = C.c_double
    lib.bo_log.argtypes = [C.c_double]
    lib.bo_uniform_accept.restype = C.c_double
    lib.bo_uniform_accept.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    lib.bo_det.restype = C.c_double
    lib.bo_logpdfnormal.restype = C.c_double
    lib.bo_traceB.restype = C.c_double
    lib.bo_r3_forward.restype = C.c_double
    lib.bo_partialbridge_nuH.restype = C.c_double
    lib.bo_llikelihood_flat.restype = C.c_double
    lib.bo_ensemble_proposals.restype = C.c_double
    lib.bo_ensemble_mcmc.restype = C.c_double
    lib.bo_girsanov.restype = C.c_double
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def _d(a):
    """contiguous float64 array + pointer (None -> NULL)"""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(dp)


def cm(A):
    """flatten a matrix (or stack of matrices, leading index = time) column-major per matrix"""
    A = np.asarray(A, dtype=np.float64)
    if A.ndim <= 1:
        return np.ascontiguousarray(A)
    if A.ndim == 2:
        return np.ascontiguousarray(A.T).ravel()
    return np.ascontiguousarray(np.swapaxes(A, -1, -2)).reshape(A.shape[0], -1)


def uncm(a, r, c):
    a = np.asarray(a)
    if a.ndim == 1:
        return a.reshape(c, r).T.copy()
    return np.swapaxes(a.reshape(a.shape[0], c, r), -1, -2).copy()


# ---- RNG ----
def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().bo_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def bo_log(x):
    return lib().bo_log(C.c_double(x))


def sincos2pi(u):
    s, c = C.c_double(), C.c_double()
    K = int(u * 2.0 ** 53)          # u = K*2^-53; the generator hands over the top 32 of its 64 source bits (K >> 21)
    assert K * 2.0 ** -53 == u
    lib().bo_sincos2pi(C.c_double(u), C.c_uint32(K >> 21), C.byref(s), C.byref(c))
    return s.value, c.value


class noise_spec:
    """`with o.noise_spec(2): ...` -- the oracle draws the full-resolution stream bhip-philox-v2 (3: bhip-philox-v3) inside the block
    (bo_set_noise_spec), the default bhip-philox-v4 outside.  The twin of the product's BHIP_OPT_NOISE_SPEC."""

    def __init__(self, spec):
        self.spec = spec

    def __enter__(self):
        self.old = lib().bo_get_noise_spec()
        lib().bo_set_noise_spec(C.c_int(self.spec))
        return self

    def __exit__(self, *exc):
        lib().bo_set_noise_spec(C.c_int(self.old))
        return False


def icdf_normal(w):
    """specification v4: the standard normal of one 32-bit word (scalar), or of an array of words"""
    if np.ndim(w) == 0:
        return lib().bo_icdf_normal(C.c_uint32(int(w)))
    w = np.ascontiguousarray(w, dtype=np.uint32)
    z = np.empty(w.shape)
    lib().bo_icdf_normals(w.ctypes.data_as(C.c_void_p), C.c_long(w.size), z.ctypes.data_as(dp))
    return z


def normals(seed, path, it, n0, n):
    z = np.empty(n)
    lib().bo_normals(C.c_uint64(seed), C.c_uint32(path), C.c_uint32(it), C.c_int(n0), C.c_int(n), z.ctypes.data_as(dp))
    return z


def uniform_accept(seed, path, it):
    return lib().bo_uniform_accept(C.c_uint64(seed), C.c_uint32(path), C.c_uint32(it))


# ---- linear algebra ----
def det(A):
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    a, p = _d(cm(A))
    return lib().bo_det(C.c_int(A.shape[0]), p)


def inv(A):
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    n = A.shape[0]
    a, p = _d(cm(A))
    o = np.empty(n * n)
    lib().bo_inv(C.c_int(n), p, o.ctypes.data_as(dp))
    return uncm(o, n, n)


def solve(A, b):
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    n = A.shape[0]
    a, p = _d(cm(A))
    bb, pb = _d(b)
    o = np.empty(n)
    lib().bo_solve(C.c_int(n), p, pb, o.ctypes.data_as(dp))
    return o


def logpdfnormal(x, Sigma):
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    S = np.atleast_2d(np.asarray(Sigma, dtype=np.float64))
    xx, px = _d(x)
    ss, ps = _d(cm(S))
    return lib().bo_logpdfnormal(C.c_int(len(x)), px, ps)


# ---- models ----
def b(model, d, par, t, x):
    pp, p = _d(par)
    xx, px = _d(x)
    o = np.empty(d)
    lib().bo_b(C.c_int(model), C.c_int(d), p, C.c_double(t), px, o.ctypes.data_as(dp))
    return o


def a(model, d, mp, par, t=0.0, x=None):
    pp, p = _d(par)
    xx, px = _d(np.zeros(d) if x is None else x)
    o = np.empty(d * d)
    lib().bo_a(C.c_int(model), C.c_int(d), C.c_int(mp), p, C.c_double(t), px, o.ctypes.data_as(dp))
    return uncm(o, d, d)


def aux_b(aux, d, apar, t, x):
    pp, p = _d(apar)
    xx, px = _d(x)
    o = np.empty(d)
    lib().bo_aux_b(C.c_int(aux), C.c_int(d), p, C.c_double(t), px, o.ctypes.data_as(dp))
    return o


def aux_B(aux, d, apar, t):
    pp, p = _d(apar)
    o = np.empty(d * d)
    lib().bo_aux_B(C.c_int(aux), C.c_int(d), p, C.c_double(t), o.ctypes.data_as(dp))
    return uncm(o, d, d)


def aux_beta(aux, d, apar, t):
    pp, p = _d(apar)
    o = np.empty(d)
    lib().bo_aux_beta(C.c_int(aux), C.c_int(d), p, C.c_double(t), o.ctypes.data_as(dp))
    return o


def aux_a(aux, d, mp, apar, t):
    pp, p = _d(apar)
    o = np.empty(d * d)
    lib().bo_aux_a(C.c_int(aux), C.c_int(d), C.c_int(mp), p, C.c_double(t), o.ctypes.data_as(dp))
    return uncm(o, d, d)


def linpro_par(B, mu, sigma):
    B = np.atleast_2d(np.asarray(B, dtype=np.float64))
    d = B.shape[0]
    return np.concatenate([cm(B), np.atleast_1d(np.asarray(mu, dtype=np.float64)).ravel(),
                           cm(np.asarray(sigma, dtype=np.float64).reshape(d, -1))])


def affine_par(B, beta, sigma):
    return linpro_par(B, beta, sigma)


# ---- guides ----
def gp_hv(tt, d, mp, aux, apar, v, hT=None):
    """GuidedBridge(tt, P, Pt, v, hT) -> Hd [N,d,d], V [N,d]"""
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    N = len(tt)
    pp, p = _d(apar)
    vv, pv = _d(np.atleast_1d(v))
    hh, ph = _d(None if hT is None else cm(np.atleast_2d(hT)))
    Hd = np.empty((N, d * d))
    V = np.empty((N, d))
    lib().bo_gp_hv(tt.ctypes.data_as(dp), C.c_int(N), C.c_int(d), C.c_int(mp), C.c_int(aux), p, pv, ph,
                   Hd.ctypes.data_as(dp), V.ctypes.data_as(dp))
    return uncm(Hd, d, d), V


def partialbridge_ode(tt, d, mp, m, aux, apar, L, Sigma):
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    N = len(tt)
    pp, p = _d(apar)
    ll, pl = _d(cm(np.asarray(L, dtype=np.float64).reshape(m, d)))
    ss, ps = _d(cm(np.asarray(Sigma, dtype=np.float64).reshape(m, m)))
    Lt = np.empty((N, m * d))
    Mt = np.empty((N, m * m))
    mut = np.empty((N, m))
    lib().bo_partialbridge_ode(tt.ctypes.data_as(dp), C.c_int(N), C.c_int(d), C.c_int(mp), C.c_int(m),
                               C.c_int(aux), p, pl, ps, Lt.ctypes.data_as(dp), Mt.ctypes.data_as(dp),
                               mut.ctypes.data_as(dp))
    return uncm(Lt, m, d), uncm(Mt, m, m), mut


def partialbridge_nuH(tt, d, mp, m, aux, apar, L, v, eps, Sigma, inplace=False):
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    N = len(tt)
    pp, p = _d(apar)
    ll, pl = _d(cm(np.asarray(L, dtype=np.float64).reshape(m, d)))
    vv, pv = _d(np.atleast_1d(v))
    ss, ps = _d(cm(np.asarray(Sigma, dtype=np.float64).reshape(m, m)))
    nut = np.empty((N, d))
    Ht = np.empty((N, d * d))
    if inplace:
        lib().bo_partialbridge_inplace(tt.ctypes.data_as(dp), C.c_int(N), C.c_int(d), C.c_int(mp), C.c_int(m),
                                       C.c_int(aux), p, pl, pv, C.c_double(eps), ps,
                                       nut.ctypes.data_as(dp), Ht.ctypes.data_as(dp))
        Cc = 0.0
    else:
        Cc = lib().bo_partialbridge_nuH(tt.ctypes.data_as(dp), C.c_int(N), C.c_int(d), C.c_int(mp), C.c_int(m),
                                        C.c_int(aux), p, pl, pv, C.c_double(eps), ps,
                                        nut.ctypes.data_as(dp), Ht.ctypes.data_as(dp))
    return nut, uncm(Ht, d, d), Cc


def traceB(tt, d, aux, apar):
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    pp, p = _d(apar)
    return lib().bo_traceB(tt.ctypes.data_as(dp), C.c_int(len(tt)), C.c_int(d), C.c_int(aux), p)


def r3_forward(tt, d, mp, aux, apar, what, y0):
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    pp, p = _d(apar)
    y0 = np.ascontiguousarray(np.atleast_1d(y0), dtype=np.float64).ravel()
    out = np.empty_like(y0)
    lib().bo_r3_forward(tt.ctypes.data_as(dp), C.c_int(len(tt)), C.c_int(d), C.c_int(mp), C.c_int(aux), p,
                        C.c_int(what), y0.ctypes.data_as(dp), C.c_int(len(y0)), out.ctypes.data_as(dp))
    return out


# ---- hot path ----
def wiener_sample(tt, mp, seed, path, it):
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    W = np.empty((len(tt), mp))
    lib().bo_wiener_sample(tt.ctypes.data_as(dp), C.c_int(len(tt)), C.c_int(mp), C.c_uint64(seed),
                           C.c_uint32(path), C.c_uint32(it), W.ctypes.data_as(dp))
    return W


def solve_em(model, d, mp, par, tt, u, W):
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    N = len(tt)
    pp, p = _d(par)
    uu, pu = _d(np.atleast_1d(u))
    ww, pw = _d(np.asarray(W).reshape(N, mp))
    X = np.empty((N, d))
    lib().bo_solve_em(C.c_int(model), C.c_int(d), C.c_int(mp), p, tt.ctypes.data_as(dp), C.c_int(N), pu, pw,
                      X.ctypes.data_as(dp))
    return X


class Proposal:
    """flat description of a guided proposal: kind + arrays in the oracle's layout"""

    def __init__(self, kind, tt, d, mp, m, model, par, aux, apar, A1=None, A2=None, A3=None, A4=None):
        self.kind, self.d, self.mp, self.m, self.model, self.aux = kind, d, mp, m, model, aux
        self.tt = np.ascontiguousarray(tt, dtype=np.float64)
        self.N = len(self.tt)
        self.par = np.ascontiguousarray(par, dtype=np.float64)
        self.apar = np.ascontiguousarray(apar, dtype=np.float64)
        self.A = [None if A is None else np.ascontiguousarray(A, dtype=np.float64) for A in (A1, A2, A3, A4)]

    def _args(self):
        ptr = [None if A is None else A.ctypes.data_as(dp) for A in self.A]
        return [C.c_int(self.kind), C.c_int(self.N), C.c_int(self.d), C.c_int(self.mp), C.c_int(self.m),
                C.c_int(self.model), self.par.ctypes.data_as(dp), C.c_int(self.aux),
                self.apar.ctypes.data_as(dp), self.tt.ctypes.data_as(dp)] + ptr


def proposal_hv(tt, d, mp, model, par, aux, apar, Hd, V):
    return Proposal(GUIDE_HV, tt, d, mp, d, model, par, aux, apar, cm(Hd), V)


def proposal_lmmu(tt, d, mp, m, model, par, aux, apar, Lt, Mt, mut, v):
    return Proposal(GUIDE_LMMU, tt, d, mp, m, model, par, aux, apar, cm(Lt), cm(Mt), mut, np.atleast_1d(v))


def proposal_nuh(tt, d, mp, model, par, aux, apar, nut, Ht, inplace=False):
    return Proposal(GUIDE_NUH_INPLACE if inplace else GUIDE_NUH, tt, d, mp, d, model, par, aux, apar, nut, cm(Ht))


def solve_guided(P, u, W):
    uu, pu = _d(np.atleast_1d(u))
    ww, pw = _d(np.asarray(W).reshape(P.N, P.mp))
    X = np.empty((P.N, P.d))
    lib().bo_solve_guided_flat(*P._args(), pu, pw, X.ctypes.data_as(dp))
    return X


def llikelihood(P, X, skip=0):
    xx, px = _d(np.asarray(X).reshape(P.N, P.d))
    return lib().bo_llikelihood_flat(*P._args(), px, C.c_int(skip))


def guided_r(P, i, x):
    """r((i,t), x, Po) at grid index i (0-based)"""
    xx, px = _d(np.atleast_1d(x))
    out = np.empty(P.d)
    lib().bo_guided_terms_flat(*P._args(), C.c_int(i), px, out.ctypes.data_as(dp), None)
    return out


def guided_drift(P, i, x):
    """_b((i,t), x, Po) = b + a*r"""
    xx, px = _d(np.atleast_1d(x))
    out = np.empty(P.d)
    lib().bo_guided_terms_flat(*P._args(), C.c_int(i), px, None, out.ctypes.data_as(dp))
    return out


def gpupdate(Hd, V, L, Sigma, v):
    Hd = np.atleast_2d(np.asarray(Hd, dtype=np.float64))
    d = Hd.shape[0]
    L = np.asarray(L, dtype=np.float64).reshape(-1, d)
    m = L.shape[0]
    a1, p1 = _d(cm(Hd))
    a2, p2 = _d(np.atleast_1d(V))
    a3, p3 = _d(cm(L))
    a4, p4 = _d(cm(np.asarray(Sigma, dtype=np.float64).reshape(m, m)))
    a5, p5 = _d(np.atleast_1d(v))
    Ho, Vo = np.empty(d * d), np.empty(d)
    lib().bo_gpupdate(C.c_int(d), C.c_int(m), p1, p2, p3, p4, p5, Ho.ctypes.data_as(dp), Vo.ctypes.data_as(dp))
    return uncm(Ho, d, d), Vo


def innovations(P, X, model=None, d=None, mp=None, par=None, tt=None):
    """innovations!(EulerMaruyama(), W, X, P): P = Proposal, or None with (model, d, mp, par, tt) for the plain target"""
    if P is not None:
        xx, px = _d(np.asarray(X).reshape(P.N, P.d))
        W = np.empty((P.N, P.d))
        lib().bo_innovations_flat(*P._args(), px, W.ctypes.data_as(dp))
        return W
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    N = len(tt)
    pp, p = _d(par)
    xx, px = _d(np.asarray(X).reshape(N, d))
    W = np.empty((N, d))
    lib().bo_innovations_flat(C.c_int(GUIDE_NONE), C.c_int(N), C.c_int(d), C.c_int(mp), C.c_int(d), C.c_int(model), p,
                              C.c_int(0), None, tt.ctypes.data_as(dp), None, None, None, None, px, W.ctypes.data_as(dp))
    return W


def girsanov(model, d, mp, par, par_t, tt, X):
    """girsanov(X, P, Pt) (src/diffusion.jl:109-123); par_t None -> Pt = Wiener"""
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    N = len(tt)
    pp, p = _d(par)
    pt, ptp = _d(par_t)
    xx, px = _d(np.asarray(X).reshape(N, d))
    return lib().bo_girsanov(C.c_int(model), C.c_int(d), C.c_int(mp), p, ptp, tt.ctypes.data_as(dp), C.c_int(N), px)


class _Res(C.Structure):
    _fields_ = [("acc", C.c_long), ("ll", C.c_double)]


def mcmc(P, x0, rho, iters, seed, path, skip=0):
    uu, pu = _d(np.atleast_1d(x0))
    W = np.empty((P.N, P.mp))
    X = np.empty((P.N, P.d))
    llt = np.empty(iters)
    acct = np.empty(iters, dtype=np.int32)
    res = _Res()
    lib().bo_mcmc_flat(*P._args(), pu, C.c_double(rho), C.c_int(iters), C.c_int(skip), C.c_uint64(seed),
                       C.c_uint32(path), W.ctypes.data_as(dp), X.ctypes.data_as(dp), llt.ctypes.data_as(dp),
                       acct.ctypes.data_as(C.POINTER(C.c_int)), C.byref(res))
    return dict(W=W, X=X, ll_trace=llt, acc_trace=acct, acc=res.acc, ll=res.ll)


def ensemble_proposals(P, x0, npaths, path0, seed, it, threads=1, want_last=False):
    uu, pu = _d(np.atleast_1d(x0))
    ll = np.empty(npaths)
    last = np.empty((npaths, P.d)) if want_last else None
    n = lib().bo_ensemble_proposals(*P._args(), pu, C.c_int(npaths), C.c_uint32(path0), C.c_uint64(seed),
                                    C.c_uint32(it), C.c_int(threads), ll.ctypes.data_as(dp),
                                    None if last is None else last.ctypes.data_as(dp))
    return n, ll, last


def ensemble_mcmc(P, x0, rho, iters, nchains, path0, seed, threads=1):
    uu, pu = _d(np.atleast_1d(x0))
    ll = np.empty(nchains)
    acc = np.empty(nchains, dtype=np.int64)
    n = lib().bo_ensemble_mcmc(*P._args(), pu, C.c_double(rho), C.c_int(iters), C.c_int(nchains),
                               C.c_uint32(path0), C.c_uint64(seed), C.c_int(threads), ll.ctypes.data_as(dp),
                               acc.ctypes.data_as(C.POINTER(C.c_long)))
    return n, ll, acc


AUX_LINEARAPPR = 4


def linearappr(model, d, mp, par, tt, Y):
    """bo_linearappr: (B [N,d,d], b [N,d], Sigma [N,d,mp]) of the target along Y   src/linpro.jl:196-204"""
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    N = len(tt)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    par = np.ascontiguousarray(par, dtype=np.float64)
    B, b, S = np.empty((N, d * d)), np.empty((N, d)), np.empty((N, d * mp))
    lib().bo_linearappr(C.c_int(model), C.c_int(d), C.c_int(mp), par.ctypes.data_as(dp), tt.ctypes.data_as(dp), C.c_int(N),
                        Y.ctypes.data_as(dp), B.ctypes.data_as(dp), b.ctypes.data_as(dp), S.ctypes.data_as(dp))
    return np.swapaxes(B.reshape(N, d, d), -1, -2).copy(), b, np.swapaxes(S.reshape(N, mp, d), -1, -2).copy()


def linearappr_par(tt, xx, B, b, Sigma):
    """parameter block of BO_AUX_LINEARAPPR: N, tt, xx, B (column-major per index), b, Sigma (column-major per index)"""
    cmN = lambda A: np.ravel(np.swapaxes(np.asarray(A, dtype=np.float64), -1, -2))
    return np.concatenate([[float(len(tt))], np.ravel(tt), np.ravel(xx), cmN(B), np.ravel(b), cmN(Sigma)])


def gp_hv_heuni(tt, d, mp, xx, B, b, Sigma, v, hT=None):
    """bo_gp_hv_heuni: (Hd [N,d,d], V [N,d]) of GuidedBridge(tt, P, Pt::LinearAppr, v, hT)   src/guip.jl:181-189"""
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    N = len(tt)
    cmN = lambda A: np.ascontiguousarray(np.swapaxes(np.asarray(A, dtype=np.float64), -1, -2))
    xx, b_ = np.ascontiguousarray(xx, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    Bc, Sc = cmN(B), cmN(Sigma)
    v = np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64)
    h = None if hT is None else np.ascontiguousarray(cm(np.atleast_2d(hT)), dtype=np.float64)
    Hd, V = np.empty((N, d * d)), np.empty((N, d))
    lib().bo_gp_hv_heuni(tt.ctypes.data_as(dp), C.c_int(N), C.c_int(d), C.c_int(mp), xx.ctypes.data_as(dp), Bc.ctypes.data_as(dp),
                         b_.ctypes.data_as(dp), Sc.ctypes.data_as(dp), v.ctypes.data_as(dp), None if h is None else h.ctypes.data_as(dp),
                         Hd.ctypes.data_as(dp), V.ctypes.data_as(dp))
    return np.swapaxes(Hd.reshape(N, d, d), -1, -2).copy(), V


def smooth_mcmc(props, mu, chol, w_old, w_new, seed, path, skip=0, stats=False):
    """bo_smooth_mcmc: joint MH over the chained proposals `props` (supplements/smoothing/smoothing.jl:99-213).
    Returns dict(X [m,N,d], W [m,N,mp], y0 [d], ll [m], acc, and with stats: mean [m,N,d], m2 [m,N,d,d], n)"""
    m, P0 = len(props), props[0]
    N, d, mp = P0.N, P0.d, P0.mp
    cat = lambda k: (None if P0.A[k] is None else np.ascontiguousarray(np.concatenate([np.ravel(P.A[k]) for P in props])))
    A = [cat(k) for k in range(4)]
    ptr = [None if a is None else a.ctypes.data_as(dp) for a in A]
    tts = np.ascontiguousarray(np.concatenate([P.tt for P in props]))
    apars = np.ascontiguousarray(np.concatenate([P.apar for P in props]))
    mu = np.ascontiguousarray(np.atleast_1d(mu), dtype=np.float64)
    ch = np.ascontiguousarray(cm(np.atleast_2d(chol)), dtype=np.float64)
    w_old, w_new = np.ascontiguousarray(w_old, dtype=np.float64), np.ascontiguousarray(w_new, dtype=np.float64)
    X, W = np.empty((m, N, d)), np.empty((m, N, mp))
    y0, ll, acc = np.empty(d), np.empty(m), C.c_long()
    mean = np.empty((m, N, d)) if stats else None
    m2 = np.empty((m, N, d * d)) if stats else None
    ns = C.c_long()
    lib().bo_smooth_mcmc_flat(C.c_int(m), C.c_int(P0.kind), C.c_int(N), C.c_int(d), C.c_int(mp), C.c_int(P0.m), C.c_int(P0.model),
                              P0.par.ctypes.data_as(dp), C.c_int(P0.aux), apars.ctypes.data_as(dp), C.c_int(len(P0.apar)),
                              tts.ctypes.data_as(dp), *ptr, mu.ctypes.data_as(dp), ch.ctypes.data_as(dp),
                              w_old.ctypes.data_as(dp), w_new.ctypes.data_as(dp), C.c_int(len(w_old)), C.c_int(skip),
                              C.c_uint64(seed), C.c_uint32(path), X.ctypes.data_as(dp), W.ctypes.data_as(dp), y0.ctypes.data_as(dp),
                              ll.ctypes.data_as(dp), C.byref(acc), None if mean is None else mean.ctypes.data_as(dp),
                              None if m2 is None else m2.ctypes.data_as(dp), C.byref(ns))
    out = dict(X=X, W=W, y0=y0, ll=ll, acc=acc.value)
    if stats:
        out.update(mean=mean, m2=np.swapaxes(m2.reshape(m, N, d, d), -1, -2).copy(), n=ns.value)
    return out


def chol_lower(A):
    """cholupper(Hermitian(A))' -- the lower factor from A's upper triangle (StaticArrays closed forms at n <= 3, the column-by-column
    factorisation above)"""
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    n = A.shape[0]
    a = np.ascontiguousarray(cm(A))
    out = np.empty(n * n)
    lib().bo_chol_lower(C.c_int(n), a.ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out.reshape(n, n).T.copy()


def lna_path(model, d, par, tt, x, direction):
    """LinearNoiseAppr's deterministic path: R3 of the target drift forward (1) / backward (-1) from x, zeros (0)"""
    tt = np.ascontiguousarray(tt, dtype=np.float64); par = np.ascontiguousarray(par, dtype=np.float64)
    x = np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
    Y = np.empty((len(tt), d))
    lib().bo_lna_path(C.c_int(model), C.c_int(d), par.ctypes.data_as(dp), tt.ctypes.data_as(dp), C.c_int(len(tt)), x.ctypes.data_as(dp),
                      C.c_int(direction), Y.ctypes.data_as(dp))
    return Y


def lna_coeffs(model, d, mp, par, tt, Y):
    """(xx, B, b, Sigma) of the LinearAppr that carries LinearNoiseAppr with deterministic path Y"""
    tt = np.ascontiguousarray(tt, dtype=np.float64); par = np.ascontiguousarray(par, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    N = len(tt)
    xx, B, b, S = np.empty((N, d)), np.empty((N, d * d)), np.empty((N, d)), np.empty((N, d * mp))
    lib().bo_lna_coeffs(C.c_int(model), C.c_int(d), C.c_int(mp), par.ctypes.data_as(dp), tt.ctypes.data_as(dp), C.c_int(N), Y.ctypes.data_as(dp),
                        xx.ctypes.data_as(dp), B.ctypes.data_as(dp), b.ctypes.data_as(dp), S.ctypes.data_as(dp))
    return xx, np.swapaxes(B.reshape(N, d, d), -1, -2).copy(), b, np.swapaxes(S.reshape(N, mp, d), -1, -2).copy()


def smooth_adaptive(model, d, mp, par, tts, Y0, L, Sigma, obs, HT, vT, w_old, w_new, adaptit, adaptmax, seed, path, hwindow=0, skip=0, lna=0):
    """bo_smooth_adaptive: the smoothing loop of supplements/smoothing/smoothing.jl:75-213 for one chain, adaptation included.
    tts [m,N]; Y0 [m,N,d] first linearisation paths; obs [m,mo] (obs[i] at the left end of segment i); (HT, vT) at the right end.
    Returns dict(X, W, y0, ll, acc, mean, m2, mu, H, Hd [m,N,d,d], V [m,N,d])"""
    tts = np.ascontiguousarray(tts, dtype=np.float64)
    m, N = tts.shape
    Y0 = np.ascontiguousarray(Y0, dtype=np.float64)
    L = np.atleast_2d(np.asarray(L, dtype=np.float64)); mo = L.shape[0]
    Lc, Sc = np.ascontiguousarray(cm(L)), np.ascontiguousarray(cm(np.atleast_2d(np.asarray(Sigma, dtype=np.float64))))
    obs = np.ascontiguousarray(np.asarray(obs, dtype=np.float64).reshape(m, mo))
    HTc = np.ascontiguousarray(cm(np.atleast_2d(np.asarray(HT, dtype=np.float64))))
    vT = np.ascontiguousarray(np.atleast_1d(vT), dtype=np.float64)
    par = np.ascontiguousarray(par, dtype=np.float64)
    w_old, w_new = np.ascontiguousarray(w_old, dtype=np.float64), np.ascontiguousarray(w_new, dtype=np.float64)
    X, W = np.empty((m, N, d)), np.empty((m, N, mp))
    y0, ll, acc = np.empty(d), np.empty(m), C.c_long()
    mean, m2 = np.empty((m, N, d)), np.empty((m, N, d * d))
    mu, H = np.empty(d), np.empty(d * d)
    Hd, V = np.empty((m, N, d * d)), np.empty((m, N, d))
    P = lambda a: a.ctypes.data_as(dp)
    lib().bo_smooth_adaptive(C.c_int(m), C.c_int(N), C.c_int(d), C.c_int(mp), C.c_int(mo), C.c_int(model), P(par), P(tts), P(Y0),
                             P(Lc), P(Sc), P(obs), P(HTc), P(vT), P(w_old), P(w_new), C.c_int(len(w_old)), C.c_int(adaptit),
                             C.c_int(adaptmax), C.c_int(hwindow), C.c_int(skip), C.c_uint64(seed), C.c_uint32(path),
                             P(X), P(W), P(y0), P(ll), C.byref(acc), P(mean), P(m2), P(mu), P(H), P(Hd), P(V), C.c_int(lna))
    return dict(X=X, W=W, y0=y0, ll=ll, acc=acc.value, mean=mean, m2=np.swapaxes(m2.reshape(m, N, d, d), -1, -2).copy(),
                mu=mu, H=H.reshape(d, d).T.copy(), Hd=np.swapaxes(Hd.reshape(m, N, d, d), -1, -2).copy(), V=V)


def mcnext(mean, m2, n, x):
    """in-place Welford update; mean [E,d], m2 [E,d*d] (column-major per entry), returns n+1"""
    E, d = mean.shape
    nn = C.c_long(n)
    xx, px = _d(x)
    lib().bo_mcnext(C.c_int(E), C.c_int(d), mean.ctypes.data_as(dp), m2.ctypes.data_as(dp), C.byref(nn), px)
    return nn.value
