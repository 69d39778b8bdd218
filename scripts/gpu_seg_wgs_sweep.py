# workgroups of the deferred statistics pass (BHIP_SEG_MCNEXT_WGS; default: one per CU) beside the proposal launches, with the 8 + 8 ring
import os, sys, subprocess
sys.path.insert(0, "."); sys.path.insert(0, "tests")
if len(sys.argv) > 1:
    import bench, bridgehip as bh
    ctx = bh.Context(0)
    r = bench.smoothing_record(ctx)
    print("workgroups %-4s shared %.4f  means-only %.4f  per-chain %.4f ms" % (sys.argv[1], r["iteration_shared_guides"]["ms"],
          r["iteration_shared_guides_means_only"]["ms"], r["iteration_per_chain_guides"]["ms"]), flush=True)
else:
    for rep in range(2):
        for w in ("128", "192", "256", "384", "512", "1024"):
            subprocess.run([sys.executable, __file__, w], env={**os.environ, "BHIP_SEG_MCNEXT_WGS": w})
