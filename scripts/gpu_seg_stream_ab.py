# the smoothing loop's statistics pass beside the proposals (second stream, the default) against behind them (BHIP_SEG_ONE_STREAM=1), and the
# pass's own workgroup count (BHIP_SEG_MCNEXT_WGS); same box, fresh processes, two rounds
import os, sys, subprocess
sys.path.insert(0, "."); sys.path.insert(0, "tests")
if len(sys.argv) > 1:
    import bench, bridgehip as bh
    ctx = bh.Context(0)
    r = bench.smoothing_record(ctx)
    print("%-40s shared %.4f  means-only %.4f  per-chain %.4f  k1 %.4f ms" % (sys.argv[1], r["iteration_shared_guides"]["ms"],
          r["iteration_shared_guides_means_only"]["ms"], r["iteration_per_chain_guides"]["ms"], r["iteration_shared_guides_stats_every_iteration"]["ms"]), flush=True)
else:
    for rep in range(2):
        combos = [("default (second stream, 8 8)", {})]
        for kl in ("15 1", "14 2", "12 4"):
            combos.append((f"second stream + DEFER {kl}", {"BHIP_SEG_DEFER": kl}))
            combos.append((f"one stream + DEFER {kl}", {"BHIP_SEG_ONE_STREAM": "1", "BHIP_SEG_DEFER": kl}))
        for name, env in combos:
            subprocess.run([sys.executable, __file__, name], env={**os.environ, **env})
