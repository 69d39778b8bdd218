"""GPU tests of the collective inside the product: bhip_comm_* (RCCL over xGMI, include/bridgehip.h), SURVEY 8(e).

A gpurun box has ONE GPU, and RCCL refuses two ranks on one device, so what can run here is the real RCCL code path with
a world of one (communicator creation from a unique id, the all-gather on the context's stream, the single-process
ncclCommInitAll form) plus, where the box has two or more GPUs, the 2-rank equality test through the same entry points.
The N > 1 arithmetic (sharding by global id, merging the gathered blocks) is covered on CPU by tests/test_dist_gloo.py.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import bridgehip as bh
import problems
from bridgehip import dist as bdist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    return bh.default_context(0)


def test_rccl_allgather_of_the_statistics_block_world_of_one(ctx):
    case = [c for c in problems.cases(129) if c.name == "fhn_partialbridge_extreme"][0]
    Po = case.bh_proposal(bh, ctx)
    ch = bh.Chains(Po, case.x0, 512, seed=3)
    ch.step(0.9, 5)
    stats = ch.stats()
    comm = bdist.Comm.from_torch_dist(ctx)          # no process group: a world of one, id drawn locally
    assert (comm.nranks, comm.rank) == (1, 0)
    n, r = C.c_int(-1), C.c_int(-1)
    ctx.check(ctx.lib.bhip_comm_info(comm.h, C.byref(n), C.byref(r)))
    assert (n.value, r.value) == (1, 0)
    # ... and what RCCL itself says about the communicator (bhip_comm_query: ncclGetVersion / ncclCommCount / ncclCommUserRank)
    info = comm.info()
    assert info["nranks"] == info["rccl_nranks"] == 1 and info["rank"] == info["rccl_rank"] == 0 and info["rccl_version"] >= 2000
    g = bdist.allgather_stats(stats, 1, comm)       # bhip_comm_allgather -> ncclAllGather on the context's stream
    torch.cuda.synchronize()
    assert g.shape == (1, bh.STATS_LEN) and torch.equal(g[0], stats)
    s = bdist.combine_stats(g)
    assert s["chains"] == 512 and s["iterations"] == 5 and s["acceptance_rate"] == ch.acc().sum() / (5 * 512)
    # a longer payload (pointwise Welford state) through the generic entry point
    _, mean, m2 = ch.pathstats()
    payload = torch.tensor(np.concatenate([np.ravel(mean), np.ravel(m2)]), dtype=torch.float64, device=ctx.device)
    out = comm.allgather(payload)
    torch.cuda.synchronize()
    assert torch.equal(out[0], payload)
    comm.destroy()


def test_single_process_init_all_and_group_gather(ctx):
    """bhip_comm_init_all over the process's devices (here: as many as the box has) + bhip_comm_allgather_group"""
    ndev = torch.cuda.device_count()
    ctxs = [ctx] + [bh.Context(k) for k in range(1, ndev)]
    hs = (C.c_void_p * ndev)(*[c.h for c in ctxs])
    comms = (C.c_void_p * ndev)()
    ctx.check(ctx.lib.bhip_comm_init_all(ndev, hs, comms))
    send = [torch.arange(8, dtype=torch.float64, device=c.device) + 100 * k for k, c in enumerate(ctxs)]
    recv = [torch.zeros(8 * ndev, dtype=torch.float64, device=c.device) for c in ctxs]
    sp = (C.c_void_p * ndev)(*[t.data_ptr() for t in send])
    rp = (C.c_void_p * ndev)(*[t.data_ptr() for t in recv])
    ctx.check(ctx.lib.bhip_comm_allgather_group(ndev, comms, sp, rp, 8))
    for c in ctxs:
        c.sync()
    want = torch.cat([torch.arange(8, dtype=torch.float64) + 100 * k for k in range(ndev)])
    for t in recv:
        assert torch.equal(t.cpu(), want)
    for k in range(ndev):
        ctx.lib.bhip_comm_destroy(comms[k])
    # argument checking
    assert ctx.lib.bhip_comm_allgather_group(0, comms, sp, rp, 8) == -1
    two = (C.c_void_p * 2)(ctx.h, ctx.h)
    out2 = (C.c_void_p * 2)()
    assert ctx.lib.bhip_comm_init_all(2, two, out2) == -1   # two ranks on one device


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_bench_two_ranks_over_rccl_equal_one_rank(tmp_path):
    """bench.py --gpus 2 over real RCCL (torch 'nccl' group for the handshake, bhip_comm_allgather_stats for the data):
    2 ranks x 4096 chains == 1 rank x 8192 chains, exactly"""
    common = ["--steps", "4", "--warmup", "0", "--no-cpu-baseline", "--no-other-modes"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--chains", "4096"] + common,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--chains", "8192"] + common,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["config"]["acceptance_rate"] == j1["config"]["acceptance_rate"] and j2["config"]["chains_total"] == 8192


def test_bench_single_process_path_runs_the_collective_at_one_gpu():
    """`python bench.py --gpus 1` with no launcher: one process, bhip_comm_init_all + bhip_comm_allgather_group with a world of one"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--chains", "4096", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-other-modes"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and "bhip_comm_init_all" in j["config"]["launch"] and "unavailable" not in j["config"]["launch"]
    assert j["allgather_ms"] >= 0.0 and len(j["per_gpu_ms_per_step"]) == 1 and j["config"]["chains_total"] == 4096
    assert j["per_gpu_ms_per_step"][0] <= j["ms_per_step"] * 1.0001
    # the communicator in the record: RCCL's own count and rank, its version, the number of gathered blocks
    cm = j["comm"]
    assert cm["backend"] == "rccl" and cm["nranks"] == cm["rccl_nranks"] == 1 and cm["ranks_seen"] == [0] and cm["gathered_blocks"] == 1
    assert cm["consistent"] is True and cm["rccl_version"].count(".") == 2 and int(cm["rccl_version"].split(".")[0]) >= 2


def test_bench_under_the_launcher_at_one_rank_and_its_fallback():
    """the driver's N > 1 form (`python -m torch.distributed.run ... bench.py --gpus N`) at one rank (BENCH_PER_RANK=1 keeps it on the
    one-process-per-GPU code path; bare, a world of one takes the launcher-less path): RCCL through torch for the handshake,
    bhip_comm_init_rank + bhip_comm_allgather_stats for the data; and the fall-back -- should the product's communicator fail on any
    rank the statistics are gathered by torch.distributed and the record says so: same numbers"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29573",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--chains", "4096", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-modes"]
    out = []
    for force in ("0", "1"):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, BENCH_FORCE_COMM_FAILURE=force, BENCH_PER_RANK="1"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        out.append(json.loads(lines[0]))
    assert "bhip_comm_init_rank" in out[0]["config"]["launch"]
    assert out[0]["comm"]["backend"] == "rccl" and out[0]["comm"]["rccl_nranks"] == 1 and out[0]["comm"]["ranks_seen"] == [0] and out[0]["comm"]["consistent"]
    assert "fall-back" in out[1]["comm"]["backend"] and "rccl_nranks" not in out[1]["comm"] and out[1]["comm"]["gathered_blocks"] == 1
    assert "torch.distributed" in out[1]["config"]["launch"] and "forced by BENCH_FORCE_COMM_FAILURE" in out[1]["comm"]["note"]
    assert out[0]["config"]["acceptance_rate"] == out[1]["config"]["acceptance_rate"] and out[0]["config"]["chains_total"] == 4096
    assert len(out[0]["per_gpu_ms_per_step"]) == 1 and 0 < out[0]["per_gpu_ms_per_step"][0] <= out[0]["ms_per_step"] * 1.0001


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` beyond the visible devices: rc != 0, a clear message, no JSON line (and no launcher message)"""
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 2 and f"only {n - 1} device(s) visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_single_process_two_gpus_equal_one_gpu_with_twice_the_chains():
    common = ["--steps", "4", "--warmup", "0", "--no-cpu-baseline", "--no-other-modes"]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--chains", "4096"] + common,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--chains", "8192"] + common,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["config"]["acceptance_rate"] == j1["config"]["acceptance_rate"] and j2["config"]["chains_total"] == 8192


def test_ungrouped_gather_on_a_single_process_communicator(ctx):
    """advisor r2: the alias pair bhip_comm_init + bhip_allgather_stats must not deadlock on ndev > 1 -- a communicator from
    bhip_comm_init[_all] with more than one rank refuses the ungrouped call (BHIP_ESTATE); a world of one may use either"""
    ndev = torch.cuda.device_count()
    ctxs = [ctx] + [bh.Context(k) for k in range(1, ndev)]
    hs = (C.c_void_p * ndev)(*[c.h for c in ctxs])
    comms = (C.c_void_p * ndev)()
    ctx.check(ctx.lib.bhip_comm_init(ndev, hs, comms))
    send = torch.arange(8, dtype=torch.float64, device=ctx.device)
    recv = torch.zeros(8 * ndev, dtype=torch.float64, device=ctx.device)
    rc = ctx.lib.bhip_allgather_stats(comms[0], C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()))
    if ndev == 1:
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(recv, send)
    else:
        assert rc == -4 and b"bhip_comm_allgather_group" in ctx.lib.bhip_last_error(ctx.h)
    for k in range(ndev):
        ctx.lib.bhip_comm_destroy(comms[k])


def test_bench_single_process_n_device_code_path_on_one_gpu():
    """the N > 1 branch of bench.py's launcher-less path (round-robin launches on one context per "device", per-device events,
    grouped gather, max over devices, the SURVEY-C4 shard record) exercised on a one-GPU box: BENCH_SAME_DEVICE=1 puts the N contexts on
    device 0 and emulates the gather on the host (RCCL refuses two ranks on one device).  2 x 4096 chains == 1 x 8192 chains."""
    env = dict(os.environ, BENCH_SAME_DEVICE="1")
    common = ["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-other-modes"]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--chains", "4096"] + common,
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--chains", "8192"] + common,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and len(j2["per_gpu_ms_per_step"]) == 2 and j2["config"]["chains_total"] == 8192
    assert j2["config"]["acceptance_rate"] == j1["config"]["acceptance_rate"]
    assert abs(j2["config"]["mean_ll"] - j1["config"]["mean_ll"]) <= 1e-12 * abs(j1["config"]["mean_ll"])
    assert j2["config"]["path_steps_per_step"] == 8192 * 1000 and j2["scaling"] == "weak"
    # the default-size form adds the 32 768-chains-per-GPU record
    full = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                          capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert full.returncode == 0, full.stdout[-2000:] + full.stderr[-2000:]
    jf = json.loads([l for l in full.stdout.splitlines() if l.startswith("{")][-1])
    assert jf["n_gpus"] == 2 and jf["survey_c4"]["chains_per_gpu"] == 32768 and jf["value"] > 0 and jf["survey_c4"]["value"] > 0
