#!/usr/bin/env python
"""Writes the inputs bridge.jl_amd/julia/bridgejl_fixtures.jl feeds to Bridge.jl itself: for three test problems the time
grid and the Wiener paths of tests/golden/guided_paths_v2.npz as CSV with 17 significant digits (exact round trip).
Output: tests/golden/julia_in/<case>_{tt,W}.csv.  (Only needed by someone who has Julia; see that script.)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import problems  # noqa: E402

g = np.load(os.path.join(HERE, "guided_paths_v2.npz"))
N = int(g["meta"][0])
out = os.path.join(HERE, "julia_in")
os.makedirs(out, exist_ok=True)
for c in problems.cases(N):
    if c.name in ("fhn_partialbridge_extreme", "fhn_partialbridge_first", "ou_guidedbridge"):
        W = g[c.name + "/W"]                       # [npaths, N, m']
        np.savetxt(os.path.join(out, c.name + "_tt.csv"), c.tt, fmt="%.17g", delimiter=",")
        np.savetxt(os.path.join(out, c.name + "_W.csv"), W.reshape(-1, W.shape[-1]), fmt="%.17g", delimiter=",")
print("written:", out)
