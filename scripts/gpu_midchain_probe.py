#!/usr/bin/env python3
"""pCN chains of a LinPro GuidedBridge at d = 4..8: the path-per-lane kernel on 16-byte slots vs the zero-padded MFMA tile kernel
(BHIP_OPT_MID_VALU = 0).  ms per MH iteration, 65 536 chains x 1000 steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bridgehip as bh
import problems

ctx = bh.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536

def t(ch, k=8):
    ch.step(0.9, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ch.step(0.9, k); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k

for d in [int(x) for x in os.environ.get("PROBE_DIMS", "4 5 6 8").split()]:
    c = problems.linpro_big_case(d, 1001)
    Po = c.bh_proposal(bh, ctx)
    out = []
    for opt in (12, 0):
        ctx.set_option(bh.OPT_MID_VALU, opt)
        ch = bh.Chains(Po, c.x0, n, seed=1)
        out.append(t(ch))
        del ch
        torch.cuda.empty_cache()
    ctx.set_option(bh.OPT_MID_VALU, 1)
    print(f"d = {d}: one path per lane (slots) {out[0]:8.3f} ms   zero padded on the tile kernel {out[1]:8.3f} ms   x{out[1] / out[0]:.2f}")
