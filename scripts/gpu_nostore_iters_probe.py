"""pCN chains stepped several iterations per call: all but the last run the instantiation WITHOUT the path store (bhip_api.hip chains_step_once) --
the small-ensemble k_pc<.., 7, 0, 2|4> kernels; ms per iteration next to the one-iteration call (with the store)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import bridgehip as bh, problems
ctx = bh.default_context(0)
for name in ("fhn_partialbridge_extreme", "nclar_firstcomponent", "ou_guidedbridge"):
    c = [c for c in problems.cases(1001) if c.name == name][0]
    Po = c.bh_proposal(bh, ctx)
    for n in (32768, 65536):
        ch = bh.Chains(Po, c.x0, n, seed=1)
        def t(iters, reps=10):
            ch.step(0.9, iters); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): ch.step(0.9, iters)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps / iters
        print(f"{name:28s} {n:6d} chains: 1 iteration per call (path store) {t(1):.4f} ms   20 per call (19 without the store) {t(20, 3):.4f} ms per iteration", flush=True)
        del ch
