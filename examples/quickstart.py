"""The README example: Bridge.jl's call sequence on ensembles (run from the repository root on an MI355X)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bridgehip as bh

npaths = int(os.environ.get("QUICKSTART_PATHS", "65536"))
tt = np.linspace(0, 2, 1001); tt = tt * (2 - tt / 2)                      # the scripts' time change
P = bh.FitzhughDiffusion(0.1, 0.0, 1.5, 0.8, 0.3)                         # or bh.UserProcess(d, "HIP C++ text of b(t,x,P)", ...)
Pt = bh.fitzhugh_aux_linearised_end(P, 1.1)
Po = bh.PartialBridge(tt, P, Pt, [[1.0, 0.0]], [1.1], [[1e-10]])          # guide ODEs on the host, as in the reference

W = bh.sample(tt, bh.Wiener(1), npaths=npaths, seed=1)                    # sample(tt, Wiener())
X = bh.solve(bh.Euler(), [-0.5, -0.6], W, Po)                             # solve(Euler(), x0, W, Po)
ll = bh.llikelihood(bh.LeftRule(), X, Po)                                 # one value per path (device tensor)

ch = bh.Chains(Po, [-0.5, -0.6], nchains=4 * npaths, seed=44)             # the MCMC loop of partialbridge_fitzhugh.jl
ch.step(0.9, iters=100)                                                   # rho = 0.9
Xcur = ch.current_X()                                                     # current paths (EnsemblePath)
print(f"proposals: mean ll {float(ll.mean()):.3f};  chains: acceptance {ch.acc().mean() / 100:.3f}, mean ll {ch.ll().mean():.3f}, "
      f"endpoint x1 {float(Xcur.data[-1, 0].mean()):.4f}")
