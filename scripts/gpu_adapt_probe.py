"""Times of the adaptive smoothing loop on one MI355X: Lorenz, m segments of M steps, n chains.
  step (shared guides)       one joint MH iteration over the m segments, guides shared by the ensemble
  adapt_device               per-chain re-linearisation + guide ODE + gpupdate chain + pi0 for ALL chains on the device
  step (per-chain guides)    one iteration when every chain reads its own coefficient rows
  host guide                 ONE chain's guides on the host (bhip_linearappr + bhip_proposal_guide_hv + bhip_gpupdate per segment)
Prints ms, the algorithmic bytes and GB/s of the two device paths."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import bridgehip as bh

ctx = bh.default_context(0)
m = int(os.environ.get("PROBE_M", 4))
M = int(os.environ.get("PROBE_STEPS", 250))
n = int(os.environ.get("PROBE_CHAINS", 32768))
par = dict(theta=(10.0, 20.0, 8 / 3), sigma=(3.0, 3.0, 3.0))
P = bh.Lorenz(par["theta"], par["sigma"])
tgrid = np.linspace(0.0, 0.02 * m * M / 10, m * M + 1)


def drift_path(tt, x0):
    Y = np.zeros((len(tt), 3)); y = np.array(x0, dtype=np.float64)
    for i in range(len(tt)):
        Y[i] = y
        if i + 1 < len(tt):
            y = y + np.array([10 * (y[1] - y[0]), y[0] * (20 - y[2]) - y[1], y[0] * y[1] - 8 / 3 * y[2]]) * (tt[i + 1] - tt[i])
    return Y


truth = drift_path(tgrid, (1.5, -1.5, 25.0))
L, Sig = np.eye(3), 0.25 * np.eye(3)
obs = truth[::M] + 0.5 * np.random.default_rng(0).standard_normal((m + 1, 3))
HT, vT = bh.gpupdate(1e3 * np.eye(3), np.zeros(3), L, Sig, obs[m])


def build(paths):
    H, v, segs = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        segs[i] = bh.GuidedBridge(tgrid[i * M:(i + 1) * M + 1].copy(), P, bh.linearappr(paths[i]), v, H, ctx=ctx)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    return segs, v, H


first = [truth[i * M:(i + 1) * M + 1] for i in range(m)]
t0 = time.perf_counter()
segs, mu, H0 = build(first)
t_host = time.perf_counter() - t0
sc = bh.SegChains(segs, mu, bh.cholupper_t(H0), n, seed=1, mcnext=True)
rho = 0.9
wo, wn = rho, math.sqrt(1 - rho * rho)


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for k in range(reps):
        ev[k].record(); fn()
    ev[reps].record(); torch.cuda.synchronize()
    ts = [ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]
    return float(np.mean(ts)), float(np.min(ts))


N = M + 1
ps = n * m * M                                   # path-steps per iteration
t_sh = timeit(lambda: sc.step(wo, wn, 1), 10)
t_ad = timeit(lambda: sc.adapt_device(L, Sig, obs[:m], HT, vT), 5)
t_pc = timeit(lambda: sc.step(wo, wn, 1), 10)
ll, acc, _ = sc.state()
assert np.isfinite(ll).all()
b_sh = (2 * 16 * 3 + 24 + 24 + 24) * ps          # W slots r+w (16 B x m' = 3 each way), Xo store, commit: Xo read + Xc write (accepted) -- upper figure
b_pc = b_sh + 15 * 8 * ps
b_ad = (24 + 15 * 8) * n * m * N
print(f"Lorenz smoothing, m = {m} segments x {M} steps, {n} chains ({ps / 1e6:.1f} M path-steps per iteration)")
print(f"step, shared guides     : {t_sh[0]:8.3f} ms (min {t_sh[1]:.3f})   {ps / t_sh[0] / 1e6:8.2f} G path-steps/s")
print(f"adapt_device (per chain): {t_ad[0]:8.3f} ms (min {t_ad[1]:.3f})   {n * m / t_ad[0] / 1e3:8.2f} M guides/s, {b_ad / t_ad[0] / 1e6:7.1f} GB/s (reads 24 B + writes 120 B per chain and grid point)")
print(f"step, per-chain guides  : {t_pc[0]:8.3f} ms (min {t_pc[1]:.3f})   {ps / t_pc[0] / 1e6:8.2f} G path-steps/s, + 120 B of compact guide rows per path-step: >= {15 * 8 * ps / t_pc[0] / 1e6:7.1f} GB/s")
print(f"host: ONE chain's {m} guides (linearappr + Heun + gpupdate, upload): {t_host * 1e3:.2f} ms  -> {n} chains would take {t_host * n:.1f} s on one core")
