"""N > 1 path on CPU: two processes, gloo backend, rendezvous on 127.0.0.1.

The hot path has no data-path collective (chains are independent, sharded by global id); what the
multi-GPU run adds is (i) the shard arithmetic, (ii) ONE all-gather of the statistics block and its
combination, (iii) the merge of per-rank pointwise Welford states.  Those run here exactly as in
bench.py --gpus N, with CPU tensors standing in for the device blocks.
"""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bridgehip as bh
    from bridgehip import dist as bdist
    r, l, w = bdist.init("gloo")
    assert (r, w) == (rank, world)
    total = 1000 + 7                        # not divisible by world
    lo, hi = bdist.shard(total, rank, world)
    # fabricate the block k_chain_stats would produce for this shard: chain g has ll = sin(g), acc = g % 5, 20 iterations
    gid = np.arange(lo, hi)
    ll, acc = np.sin(gid), (gid % 5).astype(float)
    stats = torch.tensor([hi - lo, 20, acc.sum(), ll.sum(), (ll ** 2).sum(), ll.min(), ll.max(), (acc ** 2).sum()], dtype=torch.float64)
    g = bdist.allgather_stats(stats, world)
    summary = bdist.combine_stats(g)
    # pointwise Welford states of this shard's "paths" x_g[i] = (g, i) -> merged over ranks
    rng = np.random.default_rng(1)
    allx = rng.standard_normal((total, 6, 2))
    x = allx[lo:hi]
    mean = x.mean(0)
    dev = x - mean
    m2 = np.einsum("pir,pic->irc", dev, dev)
    n, gm, gm2 = bdist.allgather_pathstats(hi - lo, mean, m2, world)
    torch.distributed.barrier()
    q.put((rank, lo, hi, g.numpy(), summary, n, gm, gm2))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_allgather_and_merge():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, g0, s0, n0, m0, q0), (r1, lo1, hi1, g1, s1, n1, m1, q1) = res
    total = 1007
    assert lo0 == 0 and hi0 == lo1 and hi1 == total and (hi0 - lo0) - (hi1 - lo1) in (0, 1)      # contiguous, balanced shards
    assert np.array_equal(g0, g1) and g0.shape == (2, 8)                                           # every rank holds every block
    gid = np.arange(total)
    ll, acc = np.sin(gid), gid % 5
    assert s0 == s1 and s0["chains"] == total and s0["iterations"] == 20
    assert abs(s0["mean_ll"] - ll.mean()) < 1e-14 and abs(s0["var_ll"] - ll.var(ddof=1)) < 1e-12
    assert abs(s0["acceptance_rate"] - acc.mean() / 20) < 1e-15
    assert s0["min_ll"] == ll.min() and s0["max_ll"] == ll.max()
    allx = np.random.default_rng(1).standard_normal((total, 6, 2))
    assert n0 == n1 == total and np.allclose(m0, allx.mean(0), atol=1e-14) and np.array_equal(m0, m1)
    dev = allx - allx.mean(0)
    assert np.allclose(q0, np.einsum("pir,pic->irc", dev, dev), rtol=1e-12)


def test_shard_partitions_any_world_size():
    import bridgehip as bh
    from bridgehip import dist as bdist
    for total in (0, 1, 7, 262144, 262145):
        for world in (1, 2, 3, 4, 8):
            parts = [bdist.shard(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
