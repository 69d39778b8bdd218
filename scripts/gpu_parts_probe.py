"""Fresh proposals with X in one buffer against X in three parts lying in three pieces of the device memory (bhip_sample_solve_parts):
the same values, and the time per launch."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bridgehip as bh
import problems

case = [c for c in problems.cases(1001) if c.name == "fhn_partialbridge_extreme"][0]
ctx = bh.Context(0)
Po = case.bh_proposal(bh, ctx)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144


def timed(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


X1 = bh.EnsemblePath(Po.tt, Po.d, n, ctx)
ll1 = ctx.empty(n)
one = lambda: ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, bh.api._dptr(bh.api._x0(case.x0, Po.d)), None, None, n, X1.ptr(), n, bh.api.vp(ll1.data_ptr()), 0, n, 7, 1, 0))
t1 = timed(one)
for nparts in (2, 3):
    XP = bh.EnsembleParts(Po.tt, Po.d, n, nparts, ctx)
    llp = ctx.empty(n)
    parts = lambda: ctx.check(ctx.lib.bhip_sample_solve_parts(ctx.h, Po.h, bh.api._dptr(bh.api._x0(case.x0, Po.d)), XP.nparts, XP._ptrs, XP.part_paths, XP.part_paths,
                                                              bh.api.vp(llp.data_ptr()), 0, n, 7, 1, 0))
    tp = timed(parts)
    same_ll = torch.equal(ll1, llp)
    ok = all(np.array_equal(X1.paths(p, 3), XP.paths(p, 3)) for p in (0, XP.part_paths - 2, XP.part_paths, 2 * XP.part_paths - 1 if nparts > 2 else n - 3, n - 3))
    bytes_ = n * (len(Po.tt) - 1) * 8 * Po.d
    print(f"{n} paths: one buffer {t1:.4f} ms ({bytes_ / t1 / 1e6:.0f} GB/s)   {nparts} parts (apart {XP.apart}) {tp:.4f} ms ({bytes_ / tp / 1e6:.0f} GB/s)   ll equal {same_ll}  paths equal {ok}")
    XP.free()
