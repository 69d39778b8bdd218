#!/usr/bin/env python
"""Writes tests/golden/guided_paths_v5.npz: for every problem of tests/problems.py (N = 101) the Wiener paths of the noise
specification bhip-philox-v4 (one normal per 32-bit Philox word through the piecewise inverse distribution function, round 5),
the guided paths, the log-likelihoods and a short pCN chain, as computed by the CPU oracle (oracle/bridge_oracle.c) AFTER it
passed its pins (tests/test_oracle.py, K1..K14).

guided_paths_v4.npz (round 3) holds the same under the noise specification bhip-philox-v3: the oracle still reproduces it FROM
ITS SEEDS with that specification selected (bo_set_noise_spec(3); tests/test_oracle.py), as it does _v3.npz under specification v2.

guided_paths_v2.npz / _v3.npz (noise specification v2; v3 = v2 + the shared fdlibm-form sin / cos of the drift functions)
stay committed for what v1 is kept for: the paths and log-likelihoods GIVEN their stored Wiener paths.

guided_paths_v1.npz (round 1, noise specification v1) stays committed: its Wiener paths are no longer what the
generator draws, but the guided paths and log-likelihoods GIVEN those stored Wiener paths do not involve the
generator, and both the oracle and the kernels must still reproduce them bit for bit (the tests do that), so the
frozen round-1 arithmetic keeps guarding the solver across the change of the noise specification.  This script
refuses to write a new file unless that holds.

The reference (Julia) stores no guided paths or llikelihood values and cannot run here, so these
vectors do not come from Bridge.jl: they freeze the oracle + noise specification, so that a later
change to BOTH the oracle and the kernels cannot drift unnoticed.  Regenerate only with a new version suffix.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle as o          # noqa: E402
import problems             # noqa: E402

N, NPATHS, SEED, CHAIN_ITERS, RHO = 101, 3, 2026, 5, 0.9


def build():
    out = {"meta": np.array([N, NPATHS, SEED, CHAIN_ITERS], dtype=np.int64), "rho": np.array(RHO)}
    for c in problems.cases(N) + problems.forward_cases(N):
        W = np.stack([o.wiener_sample(c.tt, c.mp, SEED, p, 0) for p in range(NPATHS)])
        out[c.name + "/W"] = W
        if c.kind == o.GUIDE_NONE:
            out[c.name + "/X"] = np.stack([o.solve_em(c.model, c.d, c.mp, c.par, c.tt, c.x0, W[p]) for p in range(NPATHS)])
            continue
        ref = c.oracle_proposal()
        X = np.stack([o.solve_guided(ref, c.x0, W[p]) for p in range(NPATHS)])
        out[c.name + "/X"] = X
        out[c.name + "/ll"] = np.array([o.llikelihood(ref, X[p]) for p in range(NPATHS)])
        r = o.mcmc(ref, c.x0, RHO, CHAIN_ITERS, SEED, 1)
        out[c.name + "/chain_W"], out[c.name + "/chain_X"] = r["W"], r["X"]
        out[c.name + "/chain_ll_acc"] = np.array([r["ll"], float(r["acc"])])
    return out


def check_given_W(version):
    """the noise-independent part of an earlier file: X and ll from ITS stored W"""
    g = np.load(os.path.join(HERE, f"guided_paths_{version}.npz"))
    n, npaths = int(g["meta"][0]), int(g["meta"][1])
    for c in problems.cases(n) + problems.forward_cases(n):
        W = g[c.name + "/W"]
        tol = 0.0 if (c.exact and not (c.trig and version in ("v1", "v2"))) else 1e-12     # v1, v2 evaluated the sin drifts through libm
        if c.kind == o.GUIDE_NONE:
            X = np.stack([o.solve_em(c.model, c.d, c.mp, c.par, c.tt, c.x0, W[p]) for p in range(npaths)])
        else:
            ref = c.oracle_proposal()
            X = np.stack([o.solve_guided(ref, c.x0, W[p]) for p in range(npaths)])
            ll = np.array([o.llikelihood(ref, X[p]) for p in range(npaths)])
            assert np.abs(ll - g[c.name + "/ll"]).max() <= tol * (1 + np.abs(ll).max()), c.name
        assert np.abs(X - g[c.name + "/X"]).max() <= tol * (1 + np.abs(X).max()), c.name


if __name__ == "__main__":
    for v in ("v1", "v2", "v3", "v4"):
        check_given_W(v)
    assert o.lib().bo_get_noise_spec() == 4
    data = build()
    fn = os.path.join(HERE, "guided_paths_v5.npz")
    np.savez_compressed(fn, **data)
    print(fn, os.path.getsize(fn), "bytes,", len(data), "arrays")
