"""Round 5: the context's piece map at work -- six chain ensembles alive in one process, at two sizes: per ensemble the set-up time,
Xo candidates tested, the two-stream rates (one piece / kept pair), the pieces, and ms per pCN iteration afterwards."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bridgehip as bh
import problems

case = [c for c in problems.cases(1001) if c.name == "fhn_partialbridge_extreme"][0]
for n in ([int(a) for a in sys.argv[1:]] or [65536, 262144]):
    ctx = bh.Context(0)
    Po = case.bh_proposal(bh, ctx)
    ens = []
    print(f"== {n} chains per ensemble, six alive in one context")
    for k in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ch = bh.Chains(Po, case.x0, n, seed=9 + k)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ens.append(ch)
        i = ch.placement()
        for _ in range(30): ch.step(0.9, 1)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): ch.step(0.9, 1)
        e1.record(); torch.cuda.synchronize()
        print(f"  ensemble {k}: set-up {1e3 * (t1 - t0):7.1f} ms  tries {i['tries']:2d}  one piece {i['gbs_same_piece']:6.0f} GB/s  kept {i['gbs_kept']:6.0f} "
              f"({i['gbs_kept'] / max(i['gbs_same_piece'], 1):.2f}x)  pieces W {i['piece_w']} Xo {i['piece_xo']}   {e0.elapsed_time(e1) / 100:.4f} ms per iteration")
    del ens
    torch.cuda.empty_cache()
