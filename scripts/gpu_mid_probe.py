"""LinPro GuidedBridge at 4 <= d <= 8: one path per lane (BHIP_OPT_MID_VALU = 1) vs zero padded on the 16-row MFMA tile kernel;
262 144 fresh proposals x 1000 steps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import bridgehip as bh
ctx = bh.default_context(0)
N, P = 1001, 262144
for d in [int(x) for x in os.environ.get("PROBE_DIMS", "4 5 6 8").split()]:
    rng = np.random.default_rng(5)
    G = rng.standard_normal((d, d)) / np.sqrt(d); G2 = rng.standard_normal((d, d)) / np.sqrt(d)
    sig = 0.5 * np.eye(d) + 0.05 * G2
    Po = bh.GuidedBridge(np.linspace(0, 1, N), bh.LinPro(-np.eye(d) + 0.1 * G, np.zeros(d), sig), bh.LinPro(-np.eye(d), np.zeros(d), sig), 0.5 * np.ones(d), ctx=ctx)
    X = bh.EnsemblePath(Po.tt, d, P, ctx); ll = ctx.empty(P); x0 = np.zeros(d)
    def step(it=[0]):
        it[0] += 1
        ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, bh.api._dptr(x0), None, None, P, X.ptr(), P, bh.api.vp(ll.data_ptr()), 0, P, 4, it[0], 0))
    for valu in (12, 0):
        ctx.set_option(bh.OPT_MID_VALU, valu)
        step(); step(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): step()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"d={d} {'path per lane' if valu else 'padded MFMA tile'}: {ms:.3f} ms  {P * (N - 1) / ms * 1e3:.3e} path-steps/s  X store {8 * d * P * (N - 1) / ms / 1e6:.0f} GB/s", flush=True)
    ctx.set_option(bh.OPT_MID_VALU, 1)
