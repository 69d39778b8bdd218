"""girsanov(X, P, Pt) (src/diffusion.jl:109-123) on stored ensembles: bit-for-bit against the oracle at small
sizes, exact algebraic properties at 262 144 paths."""
import numpy as np
import pytest
import torch

import bridgehip as bh
import oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return bh.default_context(0)


def _linpro(B, mu, sg):
    return bh.LinPro(np.array(B), np.array(mu), np.array(sg))


CASES = [
    # name, process, perturbed process, oracle model id, x0, T
    ("ou", bh.OrnsteinUhlenbeck(2.0, 1.0), bh.OrnsteinUhlenbeck(1.5, 1.0), o.MODEL_OU, [0.1], 1.0),
    ("linpro1", _linpro([[-0.8]], [0.2], [[0.7]]), _linpro([[-1.1]], [0.0], [[0.7]]), o.MODEL_LINPRO, [0.5], 2.0),
    ("linpro2", _linpro([[-1.0, 0.3], [-0.2, -0.8]], [0.1, -0.2], [[0.8, 0.1], [-0.3, 0.6]]),
     _linpro([[-0.5, 0.1], [0.0, -1.2]], [0.0, 0.3], [[0.8, 0.1], [-0.3, 0.6]]), o.MODEL_LINPRO, [0.2, -0.1], 1.0),
    ("linpro3", _linpro([[-1.0, 0.3, 0.0], [-0.2, -0.8, 0.1], [0.0, 0.2, -0.9]], [0.1, -0.2, 0.0],
                        [[0.8, 0.1, 0.0], [-0.3, 0.6, 0.1], [0.0, 0.2, 0.7]]),
     _linpro([[-0.9, 0.2, 0.1], [-0.1, -0.7, 0.0], [0.1, 0.1, -1.0]], [0.0, 0.0, 0.1],
             [[0.8, 0.1, 0.0], [-0.3, 0.6, 0.1], [0.0, 0.2, 0.7]]), o.MODEL_LINPRO, [0.2, -0.1, 0.3], 1.0),
    ("lorenz", bh.Lorenz((10.0, 28.0, 8 / 3), (3.0, 3.0, 3.0)), bh.Lorenz((9.0, 27.0, 2.5), (3.0, 3.0, 3.0)), o.MODEL_LORENZ,
     [1.0, 0.0, 0.0], 0.5),
    ("fhn2", bh.FitzHughNagumo(0.1, 0.0, 1.5, 0.8, 0.3, 0.4), bh.FitzHughNagumo(0.12, 0.1, 1.4, 0.7, 0.3, 0.4), o.MODEL_FHN2,
     [-0.5, -0.6], 1.0),
]


@pytest.mark.parametrize("name,P,Pt,model,x0,T", CASES, ids=[c[0] for c in CASES])
def test_girsanov_matches_oracle(ctx, name, P, Pt, model, x0, T):
    N, npaths = 257, 70
    tt = np.linspace(0.0, T, N)
    proc = bh.PlainProcess(tt, P, ctx=ctx)
    X, _, _ = bh.sample_solve(x0, proc, npaths, seed=21)
    Xh = X.paths()
    par, par_t = np.asarray(P.params(), dtype=float), np.asarray(Pt.params(), dtype=float)
    g = bh.girsanov(X, P, Pt).cpu().numpy()
    gw = bh.girsanov(X, proc, bh.Wiener(P.d)).cpu().numpy()     # a proposal on X's grid is accepted for P
    for p in range(npaths):
        assert g[p] == o.girsanov(model, P.d, P.mp, par, par_t, tt, Xh[p]), (name, p)
        assert gw[p] == o.girsanov(model, P.d, P.mp, par, None, tt, Xh[p]), (name, p)
    assert np.all(np.isfinite(g)) and np.any(g != 0.0)


def test_girsanov_full_size_properties(ctx):
    """262 144 stored paths: girsanov(X,P,P) = 0 and girsanov(X,P,Pt) = -girsanov(X,Pt,P) hold exactly when
    sigma is shared; mean(exp(girsanov(X, Pt, P))) = 1 for X ~ P up to Monte-Carlo and O(dt) error."""
    P, Pt = bh.FitzHughNagumo(0.5, 0.0, 1.5, 0.8, 0.3, 0.4), bh.FitzHughNagumo(0.5, 0.05, 1.45, 0.8, 0.3, 0.4)
    tt = np.linspace(0.0, 1.0, 1001)
    n = 262144
    X, _, _ = bh.sample_solve([-0.5, -0.6], bh.PlainProcess(tt, P, ctx=ctx), n, seed=22)
    g = bh.girsanov(X, P, Pt)
    assert torch.equal(bh.girsanov(X, Pt, P), -g)
    assert not bool(bh.girsanov(X, P, P).any())
    w = torch.exp(-g)                                  # dPt/dP along X ~ P
    se = float(w.std()) / np.sqrt(n)
    assert abs(float(w.mean()) - 1) < 4 * se + 5e-3
    # a spot check against the oracle at this size
    for p in (0, 131071, n - 1):
        assert float(g[p]) == o.girsanov(o.MODEL_FHN2, 2, 2, P.params(), Pt.params(), tt, X.paths(p, 1)[0])


def test_girsanov_argument_checks(ctx):
    tt = np.linspace(0.0, 1.0, 11)
    X = bh.EnsemblePath(tt, 2, 4, ctx)
    X.data.zero_()
    with pytest.raises(bh.BridgeError):      # hypo-elliptic: a is singular, Gamma does not exist
        bh.girsanov(X, bh.FitzhughDiffusion(0.1, 0.0, 1.5, 0.8, 0.3), bh.FitzhughDiffusion(0.2, 0.0, 1.5, 0.8, 0.3))
    with pytest.raises(bh.BridgeError):      # different process types
        bh.girsanov(X, bh.FitzHughNagumo(0.1, 0.0, 1.5, 0.8, 0.3, 0.4), bh.OrnsteinUhlenbeck(1.0, 1.0))
    with pytest.raises(bh.BridgeError):      # dimension
        bh.girsanov(X, bh.OrnsteinUhlenbeck(1.0, 1.0), bh.OrnsteinUhlenbeck(2.0, 1.0))
    with pytest.raises(bh.BridgeError):      # grid mismatch
        bh.girsanov(X, bh.PlainProcess(np.linspace(0, 2, 11), bh.FitzHughNagumo(0.1, 0.0, 1.5, 0.8, 0.3, 0.4), ctx=ctx),
                    bh.FitzHughNagumo(0.2, 0.0, 1.5, 0.8, 0.3, 0.4))
