"""N > 1 path on CPU: two processes, gloo backend, rendezvous on 127.0.0.1.

The hot path has no data-path collective (chains are independent, sharded by global id); what the
multi-GPU run adds is (i) the shard arithmetic, (ii) ONE all-gather of the statistics block and its
combination, (iii) the merge of per-rank pointwise Welford states.  Those run here exactly as in
bench.py --gpus N, with CPU tensors standing in for the device blocks.
"""
import os
import socket

import numpy as np
import pytest
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


@pytest.mark.gpu
def test_bench_two_ranks_equal_one_rank_with_twice_the_chains(tmp_path):
    """bench.py's N > 1 path end to end on a one-GPU box (BENCH_SINGLE_DEVICE=1: both ranks on GPU 0, gloo collectives):
    rank r owns the global chain ids [r*P, (r+1)*P) and the noise is keyed by the global id, so 2 ranks x 4096 chains must
    report exactly the acceptance rate and mean log-weight of 1 rank x 8192 chains."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SINGLE_DEVICE="1")
    common = ["--steps", "4", "--warmup", "0", "--no-cpu-baseline", "--no-other-modes"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29561", os.path.join(root, "bench.py"), "--gpus", "2", "--chains", "4096"] + common,
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--chains", "8192"] + common,
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["config"]["chains_total"] == 8192 and j1["config"]["chains_total"] == 8192
    assert j2["config"]["acceptance_rate"] == j1["config"]["acceptance_rate"]
    assert abs(j2["config"]["mean_ll"] - j1["config"]["mean_ll"]) <= 1e-12 * abs(j1["config"]["mean_ll"])
    assert j2["scaling"] == "weak" and j2["config"]["path_steps_per_step"] == 8192 * 1000
    assert len(j2["per_gpu_ms_per_step"]) == 2 and all(0 < t <= j2["ms_per_step"] * 1.0001 for t in j2["per_gpu_ms_per_step"])   # every rank's device time
