"""bhip_chains_step_group / bhip_chains_stats_group (include/bridgehip.h): ONE C-ABI call that steps the ensembles of every
device.  A gpurun box has one GPU, so the ensembles here live on several CONTEXTS of device 0 (which is how a node's devices
look to the library: one context each); the contract is that the group call gives exactly what stepping every ensemble by
itself gives -- decisions, log-likelihoods, Wiener states, proposal paths, statistics -- and that its argument errors are those of
bhip_chains_step.  The loop being replaced: project_partialbridge/partialbridge_fitzhugh.jl:143-176, once per device."""
import ctypes as C
import time

import numpy as np
import pytest
import torch

import bridgehip as bh
import problems
from bridgehip import dist as bdist

pytestmark = pytest.mark.gpu


def _case(name, N=129):
    return [c for c in problems.cases(N) if c.name == name][0]


@pytest.mark.parametrize("name", ["fhn_partialbridge_extreme", "ou_guidedbridge", "nclar_firstcomponent", "linpro3_guidedbridge"])
def test_group_step_equals_per_ensemble_stepping(name):
    names = {c.name for c in problems.cases(129)}
    if name not in names:
        pytest.skip(f"no problem named {name}")
    case = _case(name)
    n, P = 3, 320
    ctxs = [bh.Context(0) for _ in range(n)]
    solo = [bh.Chains(case.bh_proposal(bh, c), case.x0, P, seed=9, path0=k * P) for k, c in enumerate(ctxs)]
    grp = [bh.Chains(case.bh_proposal(bh, c), case.x0, P, seed=9, path0=k * P) for k, c in enumerate(ctxs)]
    for ch in solo:
        ch.step(0.9, 4)
        ch.step(0.8, 1)
    g = bdist.ChainsGroup(grp)
    g.step(0.9, 4)
    g.step(0.8, 1)
    torch.cuda.synchronize()
    for a, b in zip(solo, grp):
        assert np.array_equal(a.ll(), b.ll()) and np.array_equal(a.acc(), b.acc())
        Xa, Wa = a.paths()
        Xb, Wb = b.paths()
        assert np.array_equal(Xa, Xb) and np.array_equal(Wa, Wb)
        assert torch.equal(a.proposal_X(), b.proposal_X())      # the last iteration of a call stores Xo in both forms
        assert b.iterations == 5
    outs = [c.empty(bh.STATS_LEN) for c in ctxs]
    g.stats(outs)
    torch.cuda.synchronize()
    for a, o in zip(solo, outs):
        assert torch.equal(a.stats(), o)
    # sharded by global id the three ensembles ARE one ensemble of 3 P chains
    one = bh.Chains(case.bh_proposal(bh, ctxs[0]), case.x0, n * P, seed=9)
    one.step(0.9, 4)
    one.step(0.8, 1)
    assert np.array_equal(one.ll(), np.concatenate([b.ll() for b in grp]))
    assert np.array_equal(one.acc(), np.concatenate([b.acc() for b in grp]))


def test_group_argument_errors():
    case = _case("fhn_partialbridge_extreme")
    ctx = bh.default_context(0)
    lib = ctx.lib
    a = bh.Chains(case.bh_proposal(bh, ctx), case.x0, 64, seed=1)
    b = bh.Chains(case.bh_proposal(bh, ctx), case.x0, 64, seed=2)
    hs = (C.c_void_p * 2)(a.h.value, b.h.value)
    assert lib.bhip_chains_step_group(0, hs, 0.9, 1, 0) == -1              # BHIP_EINVAL
    assert lib.bhip_chains_step_group(2, None, 0.9, 1, 0) == -1
    assert lib.bhip_chains_step_group(2, hs, 1.5, 1, 0) == -1              # rho outside [-1, 1]: nothing was launched
    assert "rho" in lib.bhip_last_error(ctx.h).decode()
    assert lib.bhip_chains_step_group(2, hs, 0.9, -1, 0) == -1
    twice = (C.c_void_p * 2)(a.h.value, a.h.value)
    assert lib.bhip_chains_step_group(2, twice, 0.9, 1, 0) == -1
    assert "twice" in lib.bhip_last_error(ctx.h).decode()
    assert a.acc().sum() == 0 and b.acc().sum() == 0
    assert lib.bhip_chains_step_group(2, hs, 0.9, 0, 0) == 0               # zero iterations: a no-op
    assert lib.bhip_chains_step_group(2, hs, 0.9, 2, -1) == 0              # BHIP_SKIP_OF_INIT
    torch.cuda.synchronize()
    a2 = bh.Chains(case.bh_proposal(bh, ctx), case.x0, 64, seed=1)
    a2.step(0.9, 2)
    assert np.array_equal(a.ll(), a2.ll())
    outs = (C.c_void_p * 2)(ctx.empty(bh.STATS_LEN).data_ptr(), None)
    assert lib.bhip_chains_stats_group(2, hs, outs) == -1


def test_group_host_issue_time_eight_contexts():
    """what a single host thread pays to keep 8 devices busy: the time inside ONE bhip_chains_step_group call of 8 ensembles
    (8 contexts on this box's one device).  The bound is loose (a loaded test box): the bench line reports the figure."""
    case = _case("fhn_partialbridge_extreme", 1001)
    ctxs = [bh.Context(0) for _ in range(8)]
    chs = [bh.Chains(case.bh_proposal(bh, c), case.x0, 4096, seed=5, path0=k * 4096) for k, c in enumerate(ctxs)]
    g = bdist.ChainsGroup(chs)
    g.step(0.9, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 20
    g.step(0.9, iters)            # asynchronous: this is issue time, not kernel time
    dt = (time.perf_counter() - t0) / iters * 1e6
    torch.cuda.synchronize()
    print(f"bhip_chains_step_group, 8 contexts: {dt:.1f} us of host time per iteration")
    assert dt < 400.0
