# the ring of path buffers of the smoothing loop (mcnext! every K iterations, L spare buffers: BHIP_SEG_DEFER="K L") under the round-5 kernels;
# same box, fresh processes, two rounds
import os, sys, subprocess
sys.path.insert(0, "."); sys.path.insert(0, "tests")
if len(sys.argv) > 1:
    import bench, bridgehip as bh
    ctx = bh.Context(0)
    r = bench.smoothing_record(ctx)
    print("K L = %-5s shared %.4f  means-only %.4f  per-chain %.4f ms" % (sys.argv[1], r["iteration_shared_guides"]["ms"],
          r["iteration_shared_guides_means_only"]["ms"], r["iteration_per_chain_guides"]["ms"]), flush=True)
else:
    for rep in range(2):
        for kl in ("4 4", "4 2", "8 4", "8 8", "12 4", "2 2"):
            subprocess.run([sys.executable, __file__, kl], env={**os.environ, "BHIP_SEG_DEFER": kl})
