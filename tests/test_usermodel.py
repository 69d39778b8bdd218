"""User-defined target drift b(t,x,P) compiled at run time with hipRTC (bhip_model_define): the
reference's extension point "add a Bridge.b method for your own process type" (README.md:69-77).

CPU part: definition, validation and error reporting (hipRTC compiles without a GPU).
GPU part: a user copy of the FitzHugh-Nagumo drift must reproduce the built-in functor bit for bit
in every kernel mode, and a process that is NOT in the registry (double-well drift) is checked against
a plain-Python restatement of the reference loops (src/euler.jl:262-265, src/guip.jl:192-193,429-438).
"""
import numpy as np
import pytest
import torch

import bridgehip as bh
import oracle as o
import problems

FHN_SRC = "o[0] = (x[0] - x[1] - x[0]*x[0]*x[0] + par[1]) / par[0];  o[1] = par[2]*x[0] - x[1] + par[3];"


def test_define_validate_and_errors():
    h = bh.Context(-1)
    P = bh.UserProcess(2, FHN_SRC, [0.1, 0.0, 1.5, 0.8], [[0.0], [0.3]], ctx=h)
    assert P.model_id >= 1000 and P.mp == 1
    with pytest.raises(bh.BridgeError, match="undeclared identifier"):
        bh.UserProcess(1, "o[0] = nope * x[0];", [1.0], [[1.0]], ctx=h)
    # (round 6: the full-form body runs above d = 3 as a component-wise model when m' = d; what stays refused is m' != d there)
    assert bh.UserProcess(4, "o[0] = 0;", [1.0], np.eye(4), ctx=h).model_id >= 1000
    with pytest.raises(bh.BridgeError, match="1..3"):
        bh.UserProcess(4, "o[0] = 0;", [1.0], np.ones((4, 2)), ctx=h)
    # the host side (guide ODEs) treats a user model like any other: a = sigma*sigma'
    c = [k for k in problems.cases(51) if k.name == "fhn_partialbridge_first"][0]
    Po = bh.PartialBridge(c.tt, P, c.bh_aux(bh), c.L, c.v, c.Sigma, ctx=h)
    g = c.oracle_guide()
    assert np.array_equal(Po.L, g["L"]) and np.array_equal(Po.M, g["M"]) and np.array_equal(Po.mu, g["mu"])
    # wrong parameter count for the model (npar + d*mp expected)
    import ctypes as C
    out = C.c_void_p()
    tt = np.linspace(0, 1, 5)
    par = np.zeros(3)
    assert h.lib.bhip_proposal_create(h.h, bh.api._dptr(tt), 5, P.model_id, 2, bh.api._dptr(par), 3, C.byref(out)) == -1
    assert h.lib.bhip_proposal_create(h.h, bh.api._dptr(tt), 5, 5000, 2, bh.api._dptr(par), 3, C.byref(out)) == -1


@pytest.mark.gpu
def test_user_fhn_reproduces_builtin_bitwise():
    ctx = bh.default_context(0)
    c = [k for k in problems.cases(201) if k.name == "fhn_partialbridge_extreme"][0]
    Pu = bh.UserProcess(2, FHN_SRC, c.par[:4], [[0.0], [c.par[4]]], ctx=ctx)
    Pt = c.bh_aux(bh)
    Po_u = bh.PartialBridge(c.tt, Pu, Pt, c.L, c.v, c.Sigma, ctx=ctx)
    Po_b = c.bh_proposal(bh, ctx)
    Xu, Wu, llu = bh.sample_solve(c.x0, Po_u, 300, seed=8, store_W=True)
    Xb, Wb, llb = bh.sample_solve(c.x0, Po_b, 300, seed=8, store_W=True)
    assert torch.equal(Wu.data, Wb.data) and torch.equal(Xu.data, Xb.data) and torch.equal(llu, llb)
    assert torch.equal(bh.llikelihood(bh.LeftRule(), Xu, Po_u), llb)
    assert torch.equal(bh.solve(bh.Euler(), c.x0, Wb, Po_u).data, Xb.data)
    chu, chb = bh.Chains(Po_u, c.x0, 200, seed=3), bh.Chains(Po_b, c.x0, 200, seed=3)
    chu.step(0.9, 10)
    chb.step(0.9, 10)
    assert np.array_equal(chu.ll(), chb.ll()) and np.array_equal(chu.acc(), chb.acc())
    # plain Euler-Maruyama of the user process and the (nu,H) / PartialBridge! guides as well
    Wf = bh.sample(c.tt, bh.Wiener(1), npaths=64, seed=2, ctx=ctx)
    Xf = bh.solve(bh.EulerMaruyama(), c.x0, Wf, bh.PlainProcess(c.tt, Pu, ctx=ctx))
    Xg = bh.solve(bh.EulerMaruyama(), c.x0, Wf, bh.PlainProcess(c.tt, c.bh_process(bh), ctx=ctx))
    assert torch.equal(Xf.data, Xg.data)
    c2 = [k for k in problems.cases(201) if k.name == "fhn_inplace"][0]
    P2u = bh.PartialBridgeInplace(c2.tt, Pu, c2.bh_aux(bh), c2.L, c2.v, c2.eps, c2.Sigma, ctx=ctx)
    _, _, l2u = bh.sample_solve(c2.x0, P2u, 128, seed=1)
    _, _, l2b = bh.sample_solve(c2.x0, c2.bh_proposal(bh, ctx), 128, seed=1)
    assert torch.equal(l2u, l2b)


@pytest.mark.gpu
def test_unregistered_process_double_well_guided_bridge():
    """dX = theta*(X - X^3) dt + sigma dW, guided to v by a LinPro auxiliary (GuidedBridge)"""
    ctx = bh.default_context(0)
    theta, sig, u, v = 1.3, 0.6, -0.9, 0.8
    tt = problems.tau_grid(1.0, 151)
    P = bh.UserProcess(1, "o[0] = par[0] * (x[0] - x[0]*x[0]*x[0]);", [theta], [[sig]], ctx=ctx)
    Pt = bh.LinPro([[-0.5]], [0.1], [[sig]])
    Po = bh.GuidedBridge(tt, P, Pt, [v], ctx=ctx)
    npaths = 70
    X, W, ll = bh.sample_solve([u], Po, npaths, seed=6, store_W=True)
    Xh, Wh, llh = X.paths()[:, :, 0], W.paths()[:, :, 0], ll.cpu().numpy()
    a = sig * sig
    B, mu = -0.5, 0.1
    for p in (0, 33, 69):
        assert np.array_equal(Wh[p], o.wiener_sample(tt, 1, 6, p, 0)[:, 0])
        y, som = u, 0.0
        for i in range(len(tt) - 1):
            assert Xh[p, i] == y
            dt = tt[i + 1] - tt[i]
            r = (Po.V[i, 0] - y) / Po.Hd[i, 0, 0]                    # Hd[i] \ (V[i] - x)
            b = theta * (y - y * y * y)
            som += ((b - B * (y - mu)) * r) * dt                      # dot(b - btilde, r)*dt
            y = y + (b + a * r) * dt + sig * (Wh[p, i + 1] - Wh[p, i])
        assert Xh[p, -1] == v and llh[p] == som


@pytest.mark.gpu
def test_user_lorenz_three_dimensional_noise_reproduces_builtin_bitwise():
    """a hipRTC process with noise dimension 3 (d = 3, diagonal sigma) runs on the padded line layout and the
    wave-specialised kernels (6 Philox blocks per 4-grid-point chunk, carry over the chunk boundary) like the built-in Lorenz:
    fresh proposals and pCN chains bit-identical, at ragged ensemble sizes and a grid that is not a multiple of the chunk"""
    ctx = bh.default_context(0)
    th, sg = (10.0, 20.0, 8 / 3), (3.0, 3.0, 3.0)
    src = "o[0] = par[0]*(x[1] - x[0]); o[1] = x[0]*(par[1] - x[2]) - x[1]; o[2] = x[0]*x[1] - par[2]*x[2];"
    tt = np.linspace(0.0, 0.2, 83)
    Pb = bh.Lorenz(th, sg)
    Pu = bh.UserProcess(3, src, list(th), np.diag(sg), ctx=ctx)
    Pt = bh.LinPro(-np.eye(3), np.zeros(3), np.diag(sg))
    v, x0 = [1.2, -1.0, 24.0], [1.5, -1.5, 25.0]
    Po_b = bh.GuidedBridge(tt, Pb, Pt, v, 0.2 * np.eye(3), ctx=ctx)
    Po_u = bh.GuidedBridge(tt, Pu, Pt, v, 0.2 * np.eye(3), ctx=ctx)
    for n in (70, 300):
        Xu, Wu, llu = bh.sample_solve(x0, Po_u, n, seed=8, store_W=True)
        Xb, Wb, llb = bh.sample_solve(x0, Po_b, n, seed=8, store_W=True)
        assert torch.equal(Wu.data, Wb.data) and torch.equal(Xu.data, Xb.data) and torch.equal(llu, llb)
        chu, chb = bh.Chains(Po_u, x0, n, seed=3), bh.Chains(Po_b, x0, n, seed=3)
        chu.step(0.9, 5)
        chb.step(0.9, 5)
        assert np.array_equal(chu.ll(), chb.ll()) and np.array_equal(chu.acc(), chb.acc())
        Xcu, Wcu = chu.paths(0, n)
        Xcb, Wcb = chb.paths(0, n)
        assert np.array_equal(Xcu, Xcb) and np.array_equal(Wcu, Wcb)
