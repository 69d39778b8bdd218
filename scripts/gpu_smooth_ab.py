# smoothing loop A/B on one box: time-blocked paths in parity halves + mcnext! on the second stream (default) / plain SoA paths with the
# commit copy on the second stream (BHIP_SEG_PLAIN_X=1) / the same on one stream (+ BHIP_SEG_ONE_STREAM=1: the round-3 loop)
import os, sys, subprocess
sys.path.insert(0, "."); sys.path.insert(0, "tests")
if len(sys.argv) > 1:
    import bench, bridgehip as bh
    ctx = bh.Context(0)
    r = bench.smoothing_record(ctx)
    print(sys.argv[1], " shared %.4f  means-only %.4f  per-chain %.4f  adapt %.4f ms" % (r["iteration_shared_guides"]["ms"], r["iteration_shared_guides_means_only"]["ms"],
          r["iteration_per_chain_guides"]["ms"], r["adapt_device"]["ms"]), flush=True)
else:
    for rep in range(2):
        for tag, env in (("time-blocked, two streams", {}), ("plain, two streams       ", {"BHIP_SEG_PLAIN_X": "1"}),
                         ("time-blocked, one stream ", {"BHIP_SEG_ONE_STREAM": "1"}),
                         ("plain, one stream        ", {"BHIP_SEG_PLAIN_X": "1", "BHIP_SEG_ONE_STREAM": "1"})):
            subprocess.run([sys.executable, __file__, tag], env={**os.environ, **env})
