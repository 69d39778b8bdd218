# where the path buffers of a multi-segment ensemble lie relative to its W (BHIP_SEG_SPACER_GB, measurement hook of bhip_segchains_create):
# the smoothing loop with 0 / 60 / 90 / 120 GiB held while Xc, Xtb, mean, m2 are allocated; same box, fresh processes
import os, sys, subprocess
sys.path.insert(0, "."); sys.path.insert(0, "tests")
if len(sys.argv) > 1:
    import bench, bridgehip as bh
    ctx = bh.Context(0)
    r = bench.smoothing_record(ctx)
    print("spacer %4s GiB: shared %.4f  k1 %.4f  means-only %.4f  per-chain %.4f  adapt %.4f ms" % (
        sys.argv[1], r["iteration_shared_guides"]["ms"], r["iteration_shared_guides_stats_every_iteration"]["ms"], r["iteration_shared_guides_means_only"]["ms"],
        r["iteration_per_chain_guides"]["ms"], r["adapt_device"]["ms"]), flush=True)
else:
    for rep in range(2):
        for gb in ("0", "60", "90", "120", "180"):
            subprocess.run([sys.executable, __file__, gb], env={**os.environ, "BHIP_SEG_SPACER_GB": gb})
