#!/usr/bin/env python
"""bench.py -- guided-bridge path-steps/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]/[3]): FitzHugh-Nagumo partial bridge
(project_partialbridge/partialbridge_fitzhugh.jl: eps=0.1, s=0, gamma=1.5, beta=0.8, sigma=0.3,
x0=(-0.5,-0.6), L=[1 0], Sigma=1e-10, v=1.1 "extreme", auxiliary "linearised_end", rho=0.9),
1001-point time-changed grid on T=2, fp64, 262 144 chains PER GPU (weak scaling).

A "step" is one pCN Metropolis-Hastings iteration of every chain of the rank = one launch of the
fused kernel = chains x 1000 path-steps, each path-step being: Philox normal -> Wiener increment ->
pCN mix -> guided Euler step -> log-likelihood increment (+ the accept at the end of the path).
All inputs are resident in HBM when the timed region starts.  The timed region ends with the
device-side reduction of the acceptance / log-weight statistics and (N > 1) ONE RCCL all-gather.

`--mode proposals` times independent fresh proposals instead (sample!+solve!+llikelihood, X stored).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

import bridgehip as bh
from bridgehip import dist as bdist

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
N_GRID = 1001
FHN = (0.1, 0.0, 1.5, 0.8, 0.3)
X0 = (-0.5, -0.6)
V_END = 1.1
RHO = 0.9


def tau_grid(T, N):
    s = np.linspace(0.0, T, N)
    return s * (2 - s / T)


def build_proposal(ctx):
    P = bh.FitzhughDiffusion(*FHN)
    Pt = bh.fitzhugh_aux_linearised_end(P, V_END)
    return bh.PartialBridge(tau_grid(2.0, N_GRID), P, Pt, [[1.0, 0.0]], [V_END], [[1e-10]], ctx=ctx)


def cpu_baseline(seconds_budget=20.0):
    """CPU restatement of Bridge.jl's path (oracle/bridge_oracle.c, the reference's four separate
    passes per proposal), timed on the host cores of this box on a bounded sample of the SAME workload."""
    import oracle as o          # checker / baseline leg only
    import problems
    tt = tau_grid(2.0, N_GRID)
    ap = problems.fhn_aux_end(*FHN, V_END)
    Lt, Mt, mut = o.partialbridge_ode(tt, 2, 1, 1, o.AUX_AFFINE, ap, [[1.0, 0.0]], [[1e-10]])
    Po = o.proposal_lmmu(tt, 2, 1, 1, o.MODEL_FHN, list(FHN), o.AUX_AFFINE, ap, Lt, Mt, mut, [V_END])
    ncpu = os.cpu_count() or 1
    iters = 9
    per_chain = (iters + 1) * (N_GRID - 1)

    def timed(nch, threads):
        t0 = time.perf_counter()
        n, _, _ = o.ensemble_mcmc(Po, X0, RHO, iters, nch, 0, 1, threads=threads)
        return n / (time.perf_counter() - t0)

    # 1 thread (Bridge.jl itself is single-threaded): calibrate, then ~seconds_budget/3 of work
    rate1 = timed(16, 1)
    nch1 = max(8, int(rate1 * seconds_budget / 3 / per_chain))
    rate1 = timed(nch1, 1)
    # OpenMP over chains: the best thread count is not always "all hardware threads" (SMT / NUMA); probe a
    # few counts with ~0.5 s each, then time the best one for ~seconds_budget/3
    best_t, best_r = 1, rate1
    for th in sorted({t for t in (8, 16, 32, 64, 96, 128, 192, ncpu) if 1 < t <= ncpu}):
        r = timed(max(th * 4, int(best_r * 0.5 / per_chain) // th * th), th)
        if r > best_r:
            best_t, best_r = th, r
    nchc = max(best_t, int(best_r * seconds_budget / 3 / per_chain) // best_t * best_t)
    ratec = timed(nchc, best_t)
    return {"value": ratec, "unit": "path-steps/s", "cores": best_t, "kind": "port",
            "value_1thread": rate1, "host_cpus": ncpu,
            "sample": f"{nchc} chains x {iters + 1} pCN iterations x {N_GRID - 1} steps of the bench workload "
                      f"(OpenMP over chains, {best_t} threads = the fastest of the probed counts on {ncpu} hardware threads); "
                      f"1-thread figure on {nch1} chains; "
                      "C restatement of Bridge.jl's four-pass loop (no Julia on this box), not Bridge.jl itself"}


def profiled_traffic(kernel_tag, tags=("r1_mcmc",)):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summaries
    (profiles/<tag>_fetch.txt, <tag>_write.txt; separate --pmc passes).  FETCH_SIZE / WRITE_SIZE are in
    KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM
    section), hence the factor 2.  Returns (bytes, source) or (None, None)."""
    import re
    for tag in tags:
        vals = {}
        for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            fn = os.path.join(ROOT, "profiles", f"{tag}_{kind}.txt")
            if not os.path.exists(fn):
                break
            for line in open(fn):
                if kernel_tag in line and counter in line:
                    m = re.search(counter + r"\s+\d+\s+([0-9.]+)", line)
                    if m:
                        vals[kind] = float(m.group(1))
        if len(vals) == 2:
            return (2.0 * vals["fetch"] + vals["write"]) * 1024.0, f"profiles/{tag}_fetch.txt (x2, gfx950 correction) + profiles/{tag}_write.txt"
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=262144, help="chains (paths) per GPU")
    ap.add_argument("--mode", choices=["mcmc", "proposals", "linpro32"], default="mcmc")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    ctx = bh.Context(local)
    P = args.chains
    path0 = rank * P                      # contiguous shard of the global chain ids; RNG keyed by global id
    steps_per_unit = N_GRID - 1
    stats = ctx.empty(bh.STATS_LEN)
    roof = None
    workload = None

    if args.mode == "linpro32":
        # config C5: LinPro d=32 GuidedBridge on the fp64 MFMA tile kernel; 65 536 paths by default
        d = 32
        if args.chains == 262144:
            P = 65536
            path0 = rank * P
        rng = np.random.default_rng(5)
        G, G2 = rng.standard_normal((d, d)) / np.sqrt(d), rng.standard_normal((d, d)) / np.sqrt(d)
        sig = 0.5 * np.eye(d) + 0.05 * G2
        Po = bh.GuidedBridge(np.linspace(0.0, 1.0, N_GRID), bh.LinPro(-np.eye(d) + 0.1 * G, np.zeros(d), sig),
                             bh.LinPro(-np.eye(d), np.zeros(d), sig), 0.5 * np.ones(d), ctx=ctx)
        X = bh.EnsemblePath(Po.tt, d, P, ctx)
        ll = ctx.empty(P)
        it = [0]
        x0 = np.zeros(d)

        def step():
            it[0] += 1
            ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, bh.api._dptr(x0), None, None, P, X.ptr(), P,
                                                bh.api.vp(ll.data_ptr()), 0, P, 5, it[0], path0))

        bytes_per_pathstep = 8 * d               # write X (8d)
        flops_per_pathstep = 5 * 2 * d * d       # five d x d mat-vecs
        kernel = "k_tile<32>"
        roof = ("mfma", flops_per_pathstep, 78.6, "TFLOP/s")
        workload = "LinPro d=32 GuidedBridge (dense sigma, pre-inverted Hdiamond), 1001-point grid T=1, independent fused proposals"
    elif args.mode == "mcmc":
        Po = build_proposal(ctx)
        ch = bh.Chains(Po, X0, P, seed=4, path0=path0, store_X=True)

        def step():
            ch.step(RHO, 1)

        bytes_per_pathstep = 8 * 2 + 16 * 1      # write Xo (8d) + read W, write Wo (16 m')   SURVEY 8(d) mode M
        kernel = "k_paths<MFHN, LMMU, 1, PCN>"
    else:
        Po = build_proposal(ctx)
        X = bh.EnsemblePath(Po.tt, 2, P, ctx)
        ll = ctx.empty(P)
        it = [0]
        x0 = np.array(X0)

        def step():
            it[0] += 1
            ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, bh.api._dptr(x0), None, None, P, X.ptr(), P,
                                                bh.api.vp(ll.data_ptr()), 0, P, 4, it[0], path0))

        bytes_per_pathstep = 8 * 2               # write X (8d)                               SURVEY 8(d) mode E
        kernel = "k_paths<MFHN, LMMU, 1, FRESH>"

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        evs[k].record()
        step()
    evs[args.steps].record()
    if args.mode == "mcmc":
        ch.stats(stats)
    else:
        stats.zero_()
    gathered = bdist.allgather_stats(stats, world)       # the ONE collective: acceptance / log-weight statistics
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kern_ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(args.steps)]
    kern_avg_s = float(np.mean(kern_ms)) * 1e-3

    if rank == 0:
        total_pathsteps = float(world) * P * steps_per_unit * args.steps
        value = total_pathsteps / elapsed
        alg_bytes_per_launch = float(P) * steps_per_unit * bytes_per_pathstep
        achieved = alg_bytes_per_launch / kern_avg_s / 1e9
        out = {
            "metric": "guided-bridge path-steps/sec (whole node)",
            "value": value,
            "unit": "path-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("FitzHugh-Nagumo PartialBridge (d=2, scalar noise, L=[1 0], Sigma=1e-10, v=1.1), "
                                    "1001-point tau-grid T=2, "
                                    + ("pCN-MCMC rho=0.9: one step = one MH iteration of every chain"
                                       if args.mode == "mcmc" else "independent fused proposals (sample!+solve!+llikelihood)")),
                       "mode": args.mode, "paths_per_gpu": P, "grid_points": N_GRID, "path_steps_per_step": P * steps_per_unit * world,
                       "parallelism": f"chains sharded over {world} GPU(s), one RCCL all-gather of the statistics block"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": kernel, "kernel_avg_ms": kern_avg_s * 1e3,
                         "kernel_min_ms": float(np.min(kern_ms)), "kernel_max_ms": float(np.max(kern_ms)),
                         "algorithmic_bytes_per_path_step": bytes_per_pathstep,
                         "path_steps_per_launch": P * steps_per_unit},
        }
        if args.mode == "mcmc" and P == 262144:
            # measured once per round with rocprofv3 PMC passes on this exact command (scripts/gpu_profile.sh)
            tr, src = profiled_traffic("MFHN, 2, 1, 2, 1")
            if tr:
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_source"] = src
        if workload:
            out["config"]["workload"] = workload
        if roof:   # compute-bound kernel: report against the fp64 matrix-core peak
            tf = float(P) * steps_per_unit * roof[1] / kern_avg_s / 1e12
            out["roofline"].update({"bound": roof[0], "achieved": tf, "peak": roof[2], "unit": roof[3], "frac": tf / roof[2],
                                    "algorithmic_flops_per_path_step": roof[1], "hbm_algorithmic_GBs": achieved})
        if args.mode == "mcmc":
            summary = bdist.combine_stats(gathered)
            out["config"]["acceptance_rate"] = summary["acceptance_rate"]
            out["config"]["mean_ll"] = summary["mean_ll"]
            out["config"]["chains_total"] = summary["chains"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
