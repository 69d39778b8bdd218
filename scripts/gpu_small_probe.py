"""Kernel times of the named BASELINE configurations that are NOT the bench default:
C2 (OU GuidedBridge, 65 536 paths), the SURVEY-C4 shard (FHN pCN, 32 768 chains), NCLAR 3-d at 262 144 in modes E and M,
plus the bench default for calibration of this box.  Prints ms per launch, path-steps/s and the algorithmic GB/s."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import bench
import bridgehip as bh

ctx = bh.default_context(0)
N = bench.N_GRID


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for k in range(n):
        ev[k].record()
        fn()
    ev[n].record()
    torch.cuda.synchronize()
    ts = [ev[k].elapsed_time(ev[k + 1]) for k in range(n)]
    return float(np.mean(ts)), float(np.min(ts))


def ou_proposal():
    a, beta = 0.7, 0.8
    return bh.GuidedBridge(bench.tau_grid(2.0, N), bh.LinPro([[-beta]], [0.0], [[math.sqrt(a)]]),
                           bh.LinPro([[-beta]], [0.2], [[math.sqrt(a)]]), [0.1], ctx=ctx)


def nclar_proposal():
    P = bh.NclarDiffusion(6.0, 2 * math.pi, 1.0)
    Pt = bh.AffineAux([[0, 1, 0], [0, 0, 1], [0, 0, 0]], [0, 0, 0], [[0.0], [0.0], [1.0]])
    return bh.PartialBridge(bench.tau_grid(0.5, N), P, Pt, [[1.0, 0, 0]], [5 / 128], [[1e-10]], ctx=ctx)


def fresh(Po, d, x0, P):
    X = bh.EnsemblePath(Po.tt, d, P, ctx)
    ll = ctx.empty(P)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    it = [0]

    def step():
        it[0] += 1
        ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, bh.api._dptr(x0), None, None, P, X.ptr(), P,
                                            bh.api.vp(ll.data_ptr()), 0, P, 7, it[0], 0))
    return step, (X, ll)


def chains(Po, x0, P, rho):
    ch = bh.Chains(Po, x0, P, seed=4, store_X=True)
    return (lambda: ch.step(rho, 1)), ch


rows = []
sizes = [int(s) for s in os.environ.get("PROBE_SIZES", "").split(",") if s]
Pfhn = bench.build_proposal(ctx)
for P in (sizes or [262144, 32768]):
    st, keep = chains(Pfhn, bench.X0, P, 0.9)
    rows.append((f"FHN pCN chains (mode M)  P={P}", st, P, 32, keep))
for P in (sizes or [262144, 65536]):
    st, keep = fresh(Pfhn, 2, bench.X0, P)
    rows.append((f"FHN proposals (mode E)   P={P}", st, P, 16, keep))
Pou = ou_proposal()
for P in (sizes or [65536, 262144]):
    st, keep = fresh(Pou, 1, [0.5], P)
    rows.append((f"C2 OU GuidedBridge (E)   P={P}", st, P, 8, keep))
    st, keep = chains(Pou, [0.5], P, 0.9)
    rows.append((f"C2 OU GuidedBridge (M)   P={P}", st, P, 24, keep))
Pn = nclar_proposal()
for P in (sizes or [262144]):
    st, keep = fresh(Pn, 3, [0, 0, 0], P)
    rows.append((f"NCLAR 3-d proposals (E)  P={P}", st, P, 24, keep))
    st, keep = chains(Pn, [0, 0, 0], P, 0.95)
    rows.append((f"NCLAR 3-d pCN chains (M) P={P}", st, P, 40, keep))
AB = os.environ.get("PROBE_AB") == "1"   # also time the one-lane-does-everything kernels (same process, same box)
for name, st, P, b, keep in rows:
    ms, mn = timeit(st)
    ps = P * (N - 1)
    line = (f"{name:42s} {ms:8.4f} ms (min {mn:7.4f})  {ps / ms * 1e3:10.3e} path-steps/s  "
            f"{b * ps / ms / 1e6:7.0f} GB/s of {b} B/path-step = {b * ps / ms / 1e6 / 8000:5.3f} of 8 TB/s")
    if AB:
        ctx.set_option(bh.OPT_WAVE_SPECIALISED, 0)
        ms0, mn0 = timeit(st)
        ctx.set_option(bh.OPT_WAVE_SPECIALISED, 1)
        line += f"   | monolithic {ms0:8.4f} ms (min {mn0:7.4f}) = {b * ps / ms0 / 1e6 / 8000:5.3f}  -> x{ms0 / ms:5.2f}"
    print(line, flush=True)
