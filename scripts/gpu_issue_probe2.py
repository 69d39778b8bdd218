# where the host time of a fresh-proposal launch goes (c2: 105 us per bhip_sample_solve against 3 us per bhip_chains_step)
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench, bridgehip as bh
ctx = bh.default_context(0)
def timeit(f, n=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return 1e6 * (t1 - t0) / n, 1e6 * (time.perf_counter() - t0) / n
for mode in ("c2", "proposals"):
    w = bench.Workload(mode, ctx, 0, 0)
    for _ in range(30): w.step()
    P = w.P
    x0p, Xp, llp = bh.api._dptr(w.x0), w.X.ptr(), bh.api.vp(w.ll.data_ptr())
    lib = ctx.lib
    it = [100]
    def raw():
        it[0] += 1
        lib.bhip_sample_solve(ctx.h, w.Po.h, x0p, None, None, P, Xp, P, llp, 0, P, 7, it[0], 0)
    def raw_nox():
        it[0] += 1
        lib.bhip_sample_solve(ctx.h, w.Po.h, x0p, None, None, P, None, P, llp, 0, P, 7, it[0], 0)
    def raw_noll():
        it[0] += 1
        lib.bhip_sample_solve(ctx.h, w.Po.h, x0p, None, None, P, Xp, P, None, 0, P, 7, it[0], 0)
    print(mode, "w.step            issue %.1f us, period %.1f us" % timeit(w.step), flush=True)
    print(mode, "raw ctypes call   issue %.1f us, period %.1f us" % timeit(raw), flush=True)
    print(mode, "raw, X not stored issue %.1f us, period %.1f us" % timeit(raw_nox), flush=True)
    print(mode, "raw, no ll        issue %.1f us, period %.1f us" % timeit(raw_noll), flush=True)
    for n in (4096, 16384, 65536 * 2):
        X2 = bh.EnsemblePath(w.Po.tt, w.X.dim, n, ctx); ll2 = ctx.empty(n)
        X2p, ll2p = X2.ptr(), bh.api.vp(ll2.data_ptr())
        def rawn():
            it[0] += 1
            lib.bhip_sample_solve(ctx.h, w.Po.h, x0p, None, None, n, X2p, n, ll2p, 0, n, 7, it[0], 0)
        for _ in range(10): rawn()
        print(mode, "raw, %6d paths issue %.1f us, period %.1f us" % ((n,) + timeit(rawn)), flush=True)
    del w
