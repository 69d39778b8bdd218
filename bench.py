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

`--mode proposals` times independent fresh proposals instead (sample!+solve!+llikelihood, X stored),
`--mode linpro32` config C5 (d = 32 on the fp64 matrix cores).  At N = 1 the default run appends the
kernel-level figures of those two modes as `other_modes` (measured after the timed region).
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
MFMA_F64_PEAK_TF = 78.6    # dense fp64 matrix peak, same guide
N_GRID = 1001
FHN = (0.1, 0.0, 1.5, 0.8, 0.3)
X0 = (-0.5, -0.6)
V_END = 1.1
RHO = 0.9
FHN_WORKLOAD = "FitzHugh-Nagumo PartialBridge (d=2, scalar noise, L=[1 0], Sigma=1e-10, v=1.1), 1001-point tau-grid T=2, "


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


class Workload:
    """one bench mode: owns its device buffers; step() = one launch of the dominant kernel"""

    def __init__(self, mode, ctx, chains, rank):
        self.mode, self.ctx = mode, ctx
        self.P = chains
        self.flops_per_pathstep = None
        self.chains = None
        if mode == "linpro32":
            # config C5: LinPro d=32 GuidedBridge on the fp64 MFMA tile kernel; 65 536 paths by default
            d = 32
            if chains == 262144:
                self.P = 65536
            rng = np.random.default_rng(5)
            G, G2 = rng.standard_normal((d, d)) / np.sqrt(d), rng.standard_normal((d, d)) / np.sqrt(d)
            sig = 0.5 * np.eye(d) + 0.05 * G2
            self.Po = bh.GuidedBridge(np.linspace(0.0, 1.0, N_GRID), bh.LinPro(-np.eye(d) + 0.1 * G, np.zeros(d), sig),
                                      bh.LinPro(-np.eye(d), np.zeros(d), sig), 0.5 * np.ones(d), ctx=ctx)
            self._fresh(d, np.zeros(d), 5)
            self.bytes_per_pathstep = 8 * d              # write X (8d)
            self.flops_per_pathstep = 5 * 2 * d * d      # five d x d mat-vecs
            self.kernel = "k_tile<32>"
            self.workload = "LinPro d=32 GuidedBridge (dense sigma, pre-inverted Hdiamond), 1001-point grid T=1, independent fused proposals"
        elif mode == "mcmc":
            self.Po = build_proposal(ctx)
            self.path0 = rank * self.P                   # contiguous shard of the global chain ids; RNG keyed by global id
            self.chains = bh.Chains(self.Po, X0, self.P, seed=4, path0=self.path0, store_X=True)
            self.step = lambda: self.chains.step(RHO, 1)
            self.bytes_per_pathstep = 8 * 2 + 16 * 1     # write Xo (8d) + read W, write Wo (16 m')   SURVEY 8(d) mode M
            self.kernel = "k_chain_lines<MFHN, LMMU, 1, store X>"
            self.workload = FHN_WORKLOAD + "pCN-MCMC rho=0.9: one step = one MH iteration of every chain"
        else:
            self.Po = build_proposal(ctx)
            self._fresh(2, np.array(X0), 4)
            self.bytes_per_pathstep = 8 * 2              # write X (8d)                               SURVEY 8(d) mode E
            self.kernel = "k_paths<MFHN, LMMU, 1, FRESH>"
            self.workload = FHN_WORKLOAD + "independent fused proposals (sample!+solve!+llikelihood)"
        self.path0 = rank * self.P

    def _fresh(self, d, x0, seed):
        ctx, P = self.ctx, self.P
        self.X = bh.EnsemblePath(self.Po.tt, d, P, ctx)
        self.ll = ctx.empty(P)
        self.it = 0
        self.x0 = np.ascontiguousarray(x0, dtype=np.float64)

        def step():
            self.it += 1
            ctx.check(ctx.lib.bhip_sample_solve(ctx.h, self.Po.h, bh.api._dptr(self.x0), None, None, P, self.X.ptr(), P,
                                                bh.api.vp(self.ll.data_ptr()), 0, P, seed, self.it, self.path0))
        self.step = step

    def roofline(self, kern_ms):
        """achieved = algorithmic bytes (or flops) per launch / mean launch duration (HIP events)"""
        avg_s = float(np.mean(kern_ms)) * 1e-3
        per_launch = float(self.P) * (N_GRID - 1)
        gbs = per_launch * self.bytes_per_pathstep / avg_s / 1e9
        r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
             "kernel": self.kernel, "kernel_avg_ms": avg_s * 1e3, "kernel_min_ms": float(np.min(kern_ms)),
             "kernel_max_ms": float(np.max(kern_ms)), "algorithmic_bytes_per_path_step": self.bytes_per_pathstep,
             "path_steps_per_launch": self.P * (N_GRID - 1)}
        if self.flops_per_pathstep:   # compute-bound kernel: report against the fp64 matrix-core peak
            tf = per_launch * self.flops_per_pathstep / avg_s / 1e12
            r.update({"bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F64_PEAK_TF,
                      "algorithmic_flops_per_path_step": self.flops_per_pathstep, "hbm_algorithmic_GBs": gbs})
        return r


def kernel_times(w, steps, warmup):
    """HIP-event duration of every launch (events on the stream the kernels go to: torch's current stream)"""
    for _ in range(warmup):
        w.step()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for k in range(steps):
        evs[k].record()
        w.step()
    evs[steps].record()
    torch.cuda.synchronize()
    return [evs[k].elapsed_time(evs[k + 1]) for k in range(steps)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=262144, help="chains (paths) per GPU")
    ap.add_argument("--mode", choices=["mcmc", "proposals", "linpro32"], default="mcmc")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # BENCH_SINGLE_DEVICE=1 (testing the N > 1 code path on a one-GPU box): every rank uses GPU 0 and the collectives
    # go over gloo, because RCCL refuses two ranks on one device.  Never set by the driver.
    single = os.environ.get("BENCH_SINGLE_DEVICE") == "1"
    if single:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    ctx = bh.Context(local)
    w = Workload(args.mode, ctx, args.chains, rank)
    P = w.P
    steps_per_unit = N_GRID - 1
    stats = ctx.empty(bh.STATS_LEN)

    for _ in range(args.warmup):
        w.step()
    if world > 1:   # untimed: the first collective of each kind sets up RCCL's channels over xGMI
        stats.zero_()
        bdist.allgather_stats(stats, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    evs =[torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        evs[k].record()
        w.step()
    evs[args.steps].record()
    if args.mode == "mcmc":
        w.chains.stats(stats)
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

    if rank == 0:
        total_pathsteps = float(world) * P * steps_per_unit * args.steps
        out = {
            "metric": "guided-bridge path-steps/sec (whole node)",
            "value": total_pathsteps / elapsed,
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
            "config": {"workload": w.workload, "mode": args.mode, "paths_per_gpu": P, "grid_points": N_GRID,
                       "path_steps_per_step": P * steps_per_unit * world,
                       "parallelism": f"chains sharded over {world} GPU(s), one RCCL all-gather of the statistics block"},
            "roofline": w.roofline(kern_ms),
        }
        if args.mode == "mcmc" and P == 262144:
            # measured once per round with rocprofv3 PMC passes on this exact command (scripts/gpu_profile.sh)
            tr, src = profiled_traffic("k_chain_lines<bhip::MFHN, 2, 1, 1>")
            if tr:
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_source"] = src
        if args.mode == "mcmc":
            summary = bdist.combine_stats(gathered)
            out["config"]["acceptance_rate"] = summary["acceptance_rate"]
            out["config"]["mean_ll"] = summary["mean_ll"]
            out["config"]["chains_total"] = summary["chains"]
        if world == 1 and args.mode == "mcmc" and args.chains == 262144 and not args.no_other_modes:
            # kernel-level figures of the other two workloads (outside the timed region above)
            others = []
            del w
            torch.cuda.empty_cache()
            for mode in ("proposals", "linpro32"):
                wo = Workload(mode, ctx, args.chains, rank)
                ms = kernel_times(wo, args.steps, args.warmup)
                roof = wo.roofline(ms)
                tr, src = profiled_traffic(*{"proposals": ("k_paths<bhip::MFHN, 2, 1, 1, 1>", ("r1_prop",)),
                                             "linpro32": ("k_tile<32, 1, false>", ("r1_lin32",))}[mode])
                if tr and wo.P == (262144 if mode == "proposals" else 65536):
                    roof["traffic"], roof["traffic_source"] = tr, src
                others.append({"mode": mode, "workload": wo.workload, "paths": wo.P,
                               "path_steps_per_s": wo.P * steps_per_unit / (float(np.mean(ms)) * 1e-3), "roofline": roof})
                del wo
                torch.cuda.empty_cache()
            out["other_modes"] = others
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
