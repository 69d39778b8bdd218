import sys, json; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import bench, bridgehip as bh
ctx = bh.Context(0)
r = bench.smoothing_record(ctx)
print("adapt_device", r["adapt_device"]["ms"])
