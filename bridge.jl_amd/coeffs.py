"""Host-side coefficient functions with the reference's names, for inspection, plotting and tests:

    b(t, x, P), sigma(t, x, P), a(t, x, P), Gamma(t, x, P), constdiff(P)      src/types.jl:30-36 and the
                                                                              per-process methods cited below
    B(t, Pt), beta(t, Pt), a(t, Pt), sigma(t, Pt)                             the auxiliary's 2-argument methods
    r(i, x, Po), H(i, Po), guided_b(i, x, Po)                                  src/guip.jl:192-194, src/partialbridge.jl:53-58,
                                                                              src/partialbridgenuH.jl:157-162  (i is 0-based)

They evaluate single points with numpy from a process object's parameters / a proposal's guide arrays.  The
ensemble work (sample!, solve!, llikelihood, the MCMC step) never goes through here -- that is the device
library; these are the small accessor methods a Bridge.jl user calls interactively.
"""
import math

import numpy as np

from . import api as _api


def _vec(x, d):
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    if x.shape != (d,):
        raise _api.BridgeError("point has the wrong dimension")
    return x


def b(t, x, P):
    """Bridge.b(t, x, P): drift of a target process (or of an auxiliary process: B(t)*x + beta(t))"""
    if isinstance(P, _api._Proposal):
        raise _api.BridgeError("b(t, x, Po): use guided_b(i, x, Po) for a proposal")
    if hasattr(P, "aux_kind") and not isinstance(P, _api.LinPro):
        return B(t, P) @ _vec(x, P.d) + beta(t, P)
    x = _vec(x, P.d)
    if isinstance(P, _api.Wiener):                       # src/wiener.jl:143
        return np.zeros(P.d)
    if isinstance(P, _api.OrnsteinUhlenbeck):            # test/guip.jl:21
        return np.array([-P.beta * x[0]])
    if isinstance(P, _api.LinPro):                       # src/linpro.jl:80
        return P.B @ (x - P.mu)
    if isinstance(P, _api.FitzhughDiffusion):            # partialbridge_fitzhugh.jl:44
        return np.array([(x[0] - x[1] - x[0] * x[0] * x[0] + P.s) / P.eps, P.gamma * x[0] - x[1] + P.beta])
    if isinstance(P, _api.NclarDiffusion):               # partialbridge_nclar.jl:58
        return np.array([x[1], x[2], -P.alpha * math.sin(P.omega * x[2])])
    if isinstance(P, _api.IntegratedDiffusion):          # test/partialbridge.jl:11-12
        return np.array([x[1], -(x[1] + math.sin(x[1])) + 0.5])
    if isinstance(P, _api.Lorenz):                       # src/Models.jl:47
        th = P.theta
        return np.array([th[0] * (x[1] - x[0]), x[0] * (th[1] - x[2]) - x[1], x[0] * x[1] - th[2] * x[2]])
    if isinstance(P, _api.FitzHughNagumo):               # src/Models.jl:18
        eps, s, gam, bet = P.p[:4]
        return np.array([(x[0] - x[0] * x[0] * x[0] - x[1] + s) / eps, gam * x[0] - x[1] + bet])
    if isinstance(P, _api.Pendulum):                     # src/Models.jl:79
        return np.array([x[1], -P.theta2 * math.sin(x[0])])
    raise _api.BridgeError(f"b(t, x, P): no host method for {type(P).__name__} (user texts only run on the device)")


def sigma(t, x=None, P=None):
    """Bridge.sigma(t, x, P) as a d x m' matrix; sigma(t, Pt) for an auxiliary process"""
    if P is None:
        x, P = None, x
    if hasattr(P, "aux_kind") and not isinstance(P, _api.LinPro):
        if isinstance(P, _api.FitzhughDiffusionAuxStartEnd):
            return np.array([[0.0], [P.p[4]]])
        if isinstance(P, _api.CallbackAux):
            raise _api.BridgeError("a callback auxiliary defines a(t) only")
        return P.sigma
    if isinstance(P, _api.Wiener):
        return np.eye(P.d)
    if isinstance(P, _api.OrnsteinUhlenbeck):
        return np.array([[P.sigma]])
    if isinstance(P, _api.LinPro):
        return P.sigma
    if isinstance(P, _api.FitzhughDiffusion):
        return np.array([[0.0], [P.sigma]])
    if isinstance(P, _api.NclarDiffusion):
        return np.array([[0.0], [0.0], [P.sigma]])
    if isinstance(P, _api.IntegratedDiffusion):
        return np.array([[0.0], [P.gamma]])
    if isinstance(P, _api.Lorenz):
        return np.diag(P.sigma)
    if isinstance(P, _api.FitzHughNagumo):
        return np.diag(P.p[4:6])
    if isinstance(P, _api.Pendulum):
        return np.array([[0.0], [P.gamma]])
    if isinstance(P, _api.UserProcess) and P.sigma is not None:
        return P.sigma
    raise _api.BridgeError(f"sigma: no host method for {type(P).__name__}")


def a(t, x=None, P=None):
    """Bridge.a = sigma*sigma'  (src/types.jl:32); a(t, Pt) for an auxiliary process"""
    if P is None:
        x, P = None, x
    if isinstance(P, _api.CallbackAux):
        return np.atleast_2d(np.asarray(P.fn(t)[2], dtype=np.float64))
    s = sigma(t, x, P)
    return s @ s.T


def Gamma(t, x, P):
    """Bridge.Gamma = inv(a)  (src/types.jl:33)"""
    return np.linalg.inv(a(t, x, P))


def constdiff(P):
    """Bridge.constdiff(P)  (src/types.jl:36): False only for user processes with a sigma text"""
    if isinstance(P, _api._Proposal):
        P = P.Target
    return not (isinstance(P, _api.UserProcess) and P.sigma is None)


def B(t, Pt):
    """Bridge.B(t, Pt) of an auxiliary (linear) process"""
    if isinstance(Pt, _api.LinPro) or isinstance(Pt, _api.AffineAux):
        return Pt.B
    if isinstance(Pt, _api.FitzhughDiffusionAuxStartEnd):     # partialbridge_fitzhugh.jl:70-73,103
        eps, s, gam, bet, sig, t0, u, T, v = Pt.p
        lam = (t - t0) / (T - t0)
        uv = v * lam + u * (1 - lam)
        return np.array([[1 / eps - 3 * (uv * uv) / eps, -1 / eps], [gam, -1.0]])
    if isinstance(Pt, _api.CallbackAux):
        return np.atleast_2d(np.asarray(Pt.fn(t)[0], dtype=np.float64))
    raise _api.BridgeError(f"B(t, Pt): {type(Pt).__name__} is not an auxiliary process")


def beta(t, Pt):
    """Bridge.beta(t, Pt); for LinPro  beta = -B*mu  (src/linpro.jl:82)"""
    if isinstance(Pt, _api.LinPro):
        return -Pt.B @ Pt.mu
    if isinstance(Pt, _api.AffineAux):
        return Pt.beta
    if isinstance(Pt, _api.FitzhughDiffusionAuxStartEnd):     # partialbridge_fitzhugh.jl:104
        eps, s, gam, bet, sig, t0, u, T, v = Pt.p
        lam = (t - t0) / (T - t0)
        uv = v * lam + u * (1 - lam)
        return np.array([s / eps + 2 * (uv * uv * uv) / eps, bet])
    if isinstance(Pt, _api.CallbackAux):
        return np.atleast_1d(np.asarray(Pt.fn(t)[1], dtype=np.float64))
    raise _api.BridgeError(f"beta(t, Pt): {type(Pt).__name__} is not an auxiliary process")


def r(i, x, Po):
    """Bridge.r((i,t), x, Po): the guiding term's gradient at grid index i (0-based)"""
    x = _vec(x, Po.d)
    if Po.kind == _api.GUIDE_HV:                          # Hd[i] \\ (V[i] - x)            src/guip.jl:193
        return np.linalg.solve(Po.Hd[i], Po.V[i] - x)
    if Po.kind == _api.GUIDE_LMMU:                        # L'M(v - mu - Lx)              src/partialbridge.jl:57
        return Po.L[i].T @ (Po.M[i] @ (Po.v - Po.mu[i] - Po.L[i] @ x))
    if Po.kind in (_api.GUIDE_NUH, _api.GUIDE_NUH_INPLACE):   # H(nu - x)                  src/partialbridgenuH.jl:161
        return Po.H[i] @ (Po.nu[i] - x)
    raise _api.BridgeError("r(i, x, Po): not a guided proposal")


def H(i, Po):
    """Bridge.H((i,t), x, Po)"""
    if Po.kind == _api.GUIDE_HV:                          # inv(Hd[i])                    src/guip.jl:194
        return np.linalg.inv(Po.Hd[i])
    if Po.kind == _api.GUIDE_LMMU:                        # L'ML                          src/partialbridge.jl:58
        return Po.L[i].T @ Po.M[i] @ Po.L[i]
    if Po.kind in (_api.GUIDE_NUH, _api.GUIDE_NUH_INPLACE):
        return Po.H[i]
    raise _api.BridgeError("H(i, Po): not a guided proposal")


def guided_b(i, x, Po):
    """Bridge._b((i,t), x, Po) = b(t_i, x, P) + a(t_i, x, P) * r((i,t), x, Po)"""
    t = float(Po.tt[i])
    return b(t, x, Po.Target) + a(t, x, Po.Target) @ r(i, x, Po)
