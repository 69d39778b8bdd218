# host time to ISSUE a launch against the launch's period on the device, per mode: is a short kernel's period the host's?
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench, bridgehip as bh
ctx = bh.default_context(0)
for mode in (sys.argv[1:] or ["c2", "c4shard", "proposals", "mcmc"]):
    w = bench.Workload(mode, ctx, 0, 0)
    for _ in range(50):
        w.step()
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for _ in range(n):
        w.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # single launches, each waited for: launch latency + kernel
    ts = []
    for _ in range(20):
        a = time.perf_counter(); w.step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
    print(f"{mode:10s} host issue {1e6 * (t1 - t0) / n:8.1f} us per launch   period with the queue full {1e6 * (t2 - t0) / n:8.1f} us   one launch + wait {1e6 * min(ts):8.1f} us", flush=True)
    del w
