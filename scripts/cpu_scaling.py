"""thread scaling of the CPU oracle's pCN loop on the GPU box's host cores"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as o, problems
N = 1001
FHN = (0.1, 0.0, 1.5, 0.8, 0.3)
s = np.linspace(0, 2.0, N); tt = s * (2 - s / 2.0)
ap = problems.fhn_aux_end(*FHN, 1.1)
Lt, Mt, mut = o.partialbridge_ode(tt, 2, 1, 1, o.AUX_AFFINE, ap, [[1.0, 0.0]], [[1e-10]])
Po = o.proposal_lmmu(tt, 2, 1, 1, o.MODEL_FHN, list(FHN), o.AUX_AFFINE, ap, Lt, Mt, mut, [1.1])
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (1, 8, 32, 64, 128, 256):
    nch = max(64, th * 16)
    t0 = time.perf_counter(); n, _, _ = o.ensemble_mcmc(Po, (-0.5, -0.6), 0.9, 4, nch, 0, 1, threads=th); dt = time.perf_counter() - t0
    print(th, "threads", nch, "chains", "%.3e path-steps/s" % (n / dt), "%.2fs" % dt)
