import sys, json; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import bench, bridgehip as bh
ctx = bh.Context(0)
r = bench.smoothing_record(ctx)
print("shared %.4f  means-only %.4f  per-chain %.4f  adapt %.4f ms" % (r["iteration_shared_guides"]["ms"], r["iteration_shared_guides_means_only"]["ms"],
      r["iteration_per_chain_guides"]["ms"], r["adapt_device"]["ms"]))
