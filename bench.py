#!/usr/bin/env python
"""bench.py -- guided-bridge path-steps/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W          (no launcher: ONE process drives the N devices, bhip_comm_init_all)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             (one process per GPU, bhip_comm_init_rank)

Workload (BASELINE.json configs[2]/[3]): FitzHugh-Nagumo partial bridge
(project_partialbridge/partialbridge_fitzhugh.jl: eps=0.1, s=0, gamma=1.5, beta=0.8, sigma=0.3,
x0=(-0.5,-0.6), L=[1 0], Sigma=1e-10, v=1.1 "extreme", auxiliary "linearised_end", rho=0.9),
1001-point time-changed grid on T=2, fp64, 262 144 chains PER GPU (weak scaling).

A "step" is one pCN Metropolis-Hastings iteration of every chain of the rank = one launch of the
fused kernel = chains x 1000 path-steps, each path-step being: Philox normal -> Wiener increment ->
pCN mix -> guided Euler step -> log-likelihood increment (+ the accept at the end of the path).
All inputs are resident in HBM when the timed region starts.  The timed region ends with the
device-side reduction of the acceptance / log-weight statistics and (N > 1) ONE RCCL all-gather.
It holds exactly its K steps, back to back between two HIP events (roofline.kernel_avg_ms = their distance / K);
the per-launch durations (kernel_min_ms / kernel_max_ms) come from an untimed pass of K more steps after it.

Other modes (`--mode`), each a named BASELINE / SURVEY 8(d) configuration:
  proposals   C3 as independent fresh proposals (sample!+solve!+llikelihood, X stored), 262 144 paths
  c2          C2: 1-d OU GuidedBridge, 65 536 independent paths (mode E, 8 B/path-step)
  c4shard     SURVEY C4's per-GPU shard: the default workload with 32 768 chains
  nclar       D1's "3-d": NCLAR partial bridge, 262 144 fresh proposals (24 B/path-step)
  nclar_mcmc  the same as pCN chains (40 B/path-step)
  linpro32    C5: LinPro d = 32 on the fp64 matrix cores, 65 536 paths;  linpro32_mcmc: its pCN chains
  linpro4     the same process family at d = 4: one path per lane (4 <= d <= 8), 262 144 paths
At N = 1 the default run appends the kernel-level figures of all of them as `other_modes` (measured after the timed
region), a `smoothing` record (the application loop of SURVEY 8(f) 1-2: joint MH over chained Lorenz segments with shared
and with per-chain device-built guides), a `sustained` record (>= 1 s of back-to-back launches) and the CPU baseline.  At N > 1 the SURVEY-C4 shard size
(32 768 chains per GPU) is timed after the headline region with the same barrier / max-over-ranks protocol and reported
as `survey_c4` next to the 262 144-chains-per-GPU headline.
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

import math

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
    # the threads this process may actually use: affinity mask and cgroup quota, not os.cpu_count()
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    usable = max(1, min(ncpu, int(quota)) if quota else ncpu)
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
    # OpenMP over chains: the best thread count is not always "all hardware threads" (SMT / NUMA); probe a few
    # counts with ~2 s of work each (sized from the running best rate), then time the best one for ~seconds_budget/3
    best_t, best_r = 1, rate1
    probed = {}
    for th in sorted({t for t in (8, 16, 32, 64, 128, usable // 2, usable) if 1 < t <= usable}):
        guess = max(best_r, rate1 * min(th, 16))
        r = timed(max(th * 4, int(guess * 2.0 / per_chain) // th * th), th)
        probed[th] = r
        if r > best_r:
            best_t, best_r = th, r
    nchc = max(best_t, int(best_r * seconds_budget / 3 / per_chain) // best_t * best_t)
    ratec = timed(nchc, best_t)
    return {"value": ratec, "unit": "path-steps/s", "cores": best_t, "kind": "port",
            "value_1thread": rate1, "host_cpus": os.cpu_count(), "affinity_cpus": ncpu, "cgroup_cpu_quota": quota,
            "probed_threads": {str(k): v for k, v in probed.items()},
            "sample_short": f"{nchc} chains x {iters + 1} pCN iterations x {N_GRID - 1} steps of the bench workload, {best_t} OpenMP threads; C port, not Bridge.jl",
            "sample": f"{nchc} chains x {iters + 1} pCN iterations x {N_GRID - 1} steps of the bench workload, OpenMP over chains, "
                      f"{best_t} threads (fastest probed; {usable} usable); 1-thread figure on {nch1} chains; C restatement of "
                      "Bridge.jl's four-pass loop, not Bridge.jl (no Julia here)"}


NOISE_SPEC = ("bhip-philox-v4: Philox4x32-10, four normals per call = one per 32-bit word through a piecewise polynomial inverse distribution "
              "function (256 segments of degree 4, within 3.7e-9 of the quantile, |z| <= 6.34; DESIGN 4); the reference's randn is a 52-bit ziggurat")
PROFILE_TAG = "r6"   # profiles/<PROFILE_TAG>_<mode>_{trace,fetch,write}.txt, written by scripts/gpu_profile_all.sh this round


def profiled_traffic(mode, kernel_name):
    """HBM bytes per launch of the dominant kernel from this round's committed rocprofv3 PMC summaries
    (profiles/r3_<mode>_fetch.txt, _write.txt; separate --pmc passes of `bench.py --mode <mode>`, scripts/gpu_profile.sh).
    FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
    (MI355X_MICROARCH.md, HBM section), hence the factor 2.  The kernel is looked up by the exact name the current
    build launches: a summary that does not contain it (kernel changed, profile stale) gives NO figure -- loudly."""
    import re
    vals = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        fn = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{mode}_{kind}.txt")
        if not os.path.exists(fn):
            return None, f"MISSING: {os.path.relpath(fn, ROOT)} (run scripts/gpu_profile.sh {PROFILE_TAG}_{mode} --mode {mode})"
        for line in open(fn):
            if kernel_name in line and counter in line:
                m = re.search(counter + r"\s+\d+\s+([0-9.]+)", line)
                if m:
                    vals[kind] = float(m.group(1))
        if kind not in vals:
            msg = f"STALE: {os.path.relpath(fn, ROOT)} has no {counter} row for '{kernel_name}' -- re-profile"
            print("bench.py: " + msg, file=sys.stderr)
            return None, msg
    return (2.0 * vals["fetch"] + vals["write"]) * 1024.0, (f"profiles/{PROFILE_TAG}_{mode}_fetch.txt (x2, gfx950 correction) + "
                                                           f"profiles/{PROFILE_TAG}_{mode}_write.txt")


CALIB_BYTES = 1 << 30


def calib_copy():
    """BENCH_CALIB_COPY=1 (set by live_traffic for its rocprofv3 child runs): three launches of a plain streaming kernel that reads exactly
    CALIB_BYTES and writes exactly CALIB_BYTES (torch's elementwise add of a 1-GiB tensor into another) -- the yardstick the FETCH_SIZE /
    WRITE_SIZE counters of the same pass are checked against."""
    x = torch.zeros(CALIB_BYTES // 8, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    for _ in range(3):
        torch.add(x, 1.0, out=y)
    torch.cuda.synchronize()
    del x, y
    torch.cuda.empty_cache()


def live_traffic(mode, kernel_name, steps=6):
    """HBM bytes per launch of the dominant kernel measured ON THIS BOX, now: two rocprofv3 runs of `bench.py --mode <mode>` (one
    --pmc pass each: FETCH_SIZE and WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md PMC slots; never combined with other trace
    domains), read back from the rocpd databases.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide
    coalesced read stream (same guide, HBM section), hence the factor 2.  Returns (bytes, source) or (None, reason)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="bhip_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", tmp, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--mode", mode,
                   "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-other-modes", "--no-live-traffic"]
            env = dict(os.environ, TMPDIR="/tmp", BENCH_CALIB_COPY="1")   # the child also runs a streaming kernel of KNOWN size (calib_copy)
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp) for f in fs if f.endswith(".db")]
            if not dbs:
                return None, f"rocprofv3 --pmc {counter}: no database (rc {r.returncode}): {r.stderr[-200:]}"
            cur = sqlite3.connect(dbs[0]).cursor()
            cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
            namecol = "kernel_name" if "kernel_name" in cols else "name"
            idcol = "dispatch_id" if "dispatch_id" in cols else None
            rows = list(cur.execute(f"select {idcol or namecol}, value from counters_collection where {namecol} like ? and counter_name = ?",
                                    ("%" + kernel_name + "%", counter)))   # (the database's names read `void bhip::k_pc<...>(bhip::KArgs)`)
            if not rows:
                return None, f"rocprofv3 --pmc {counter}: no row for '{kernel_name}'"
            if idcol:   # one value per dispatch = the sum over the counter's instances
                per = {}
                for did, v in rows:
                    per[did] = per.get(did, 0.0) + v
                vals[counter] = float(np.mean(list(per.values())))
            else:
                vals[counter] = float(np.mean([v for _, v in rows]))
            # the unit of the counter, validated IN THIS RUN (advisor r4): the child's calibration kernel reads and writes exactly
            # CALIB_BYTES; KiB reported for it -> the factor that turns this counter into bytes on this box (FETCH_SIZE: ~2 on gfx950 for
            # a wide coalesced read stream, the guide's correction; WRITE_SIZE: ~1)
            crow = list(cur.execute(f"select {idcol or namecol}, value from counters_collection where {namecol} like ? and counter_name = ?",
                                    ("%elementwise%add%", counter)))   # torch.add(x, 1.0, out=y): ...elementwise_kernel<.., CUDAFunctorOnSelf_add<double>, ..> -- not the fills
            if crow:
                per = {}
                for did, v in crow:
                    per[did] = per.get(did, 0.0) + v
                big = [v for v in per.values() if v > 0.25 * CALIB_BYTES / 1024.0]   # the 1-GiB launches, not torch's small fills
                if big:
                    vals[counter + "_factor"] = CALIB_BYTES / 1024.0 / float(np.mean(big))
        except Exception as e:   # noqa: BLE001 -- the record is context, never a reason to fail the bench
            return None, f"rocprofv3 --pmc {counter}: {type(e).__name__}: {e}"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    ff, wf = vals.get("FETCH_SIZE_factor"), vals.get("WRITE_SIZE_factor")
    note = (f"; factors validated in the same passes on a {CALIB_BYTES >> 20}-MiB streaming kernel of known size: FETCH_SIZE x {ff:.3f}, WRITE_SIZE x {wf:.3f} "
            "(applied: 2 and 1)" if ff and wf else "; calibration kernel not found in the passes (factors 2 and 1 as the guide prescribes)")
    if ff and wf and not (1.9 <= ff <= 2.1 and 0.95 <= wf <= 1.05):
        return None, f"counter units off on this box: FETCH_SIZE x {ff:.3f}, WRITE_SIZE x {wf:.3f} against the known-size kernel -- no traffic figure"
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, ("measured on this box in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE (x2, gfx950 "
                                                                       "correction) and --pmc WRITE_SIZE, separate passes of `bench.py --mode " + mode + "`" + note)


def profiled_valu(mode, kernel_name, paths):
    """Issue-side context for the roofline record, from this round's committed SQ counter summaries of the same command
    (profiles/r3_<mode>_sq.txt, _sq2.txt; summed over the 8 XCDs): VALU wave-instructions per path-step (SQ_INSTS_VALU x 64 lanes /
    path-steps) and the fraction of the kernel's duration the SIMDs' VALU was issuing (SQ_ACTIVE_INST_VALU, in units of 4 cycles,
    / (1024 SIMDs x GRBM_GUI_ACTIVE/8/4)).  ~0.7-0.85 for the kernels whose generator is the bottleneck, which is why their
    fraction of the HBM roofline is what it is.  None when a summary lacks the kernel (stale profile)."""
    import re

    def counter(kind, c):
        fn = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{mode}_{kind}.txt")
        if not os.path.exists(fn):
            return None
        for line in open(fn):
            if kernel_name in line and c in line:
                m = re.search(c + r"\s+\d+\s+([0-9.]+)", line)
                if m:
                    return float(m.group(1))
        return None
    iv, av, g = counter("sq", "SQ_INSTS_VALU"), counter("sq", "SQ_ACTIVE_INST_VALU"), counter("sq2", "GRBM_GUI_ACTIVE")
    if not (iv and av and g):
        return None
    return {"insts_per_path_step": iv * 64 / (paths * (N_GRID - 1)), "busy_frac": av / (1024 * g / 32),
            "source": f"profiles/{PROFILE_TAG}_{mode}_sq.txt, _sq2.txt (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE)"}


def _linpro32(ctx):
    return _linpro(ctx, 32)


def _linpro(ctx, d):
    rng = np.random.default_rng(5)
    G, G2 = rng.standard_normal((d, d)) / np.sqrt(d), rng.standard_normal((d, d)) / np.sqrt(d)
    sig = 0.5 * np.eye(d) + 0.05 * G2
    return bh.GuidedBridge(np.linspace(0.0, 1.0, N_GRID), bh.LinPro(-np.eye(d) + 0.1 * G, np.zeros(d), sig),
                           bh.LinPro(-np.eye(d), np.zeros(d), sig), 0.5 * np.ones(d), ctx=ctx)


def _ou(ctx):
    # C2: target LinPro(-0.8, 0, sqrt(.7)), auxiliary LinPro(-0.8, 0.2, sqrt(.7)), T = 2   test/guip.jl:117-120,248
    return bh.GuidedBridge(tau_grid(2.0, N_GRID), bh.LinPro([[-0.8]], [0.0], [[math.sqrt(0.7)]]),
                           bh.LinPro([[-0.8]], [0.2], [[math.sqrt(0.7)]]), [0.1], ctx=ctx)


def _nclar(ctx):
    # NCLAR(3) partial bridge (partialbridge_nclar.jl:13,43-45,52-86): alpha=6, omega=2pi, sigma=1, L=[1 0 0], v=5/128, T=0.5
    P = bh.NclarDiffusion(6.0, 2 * math.pi, 1.0)
    Pt = bh.AffineAux([[0, 1, 0], [0, 0, 1], [0, 0, 0]], [0, 0, 0], [[0.0], [0.0], [1.0]])
    return bh.PartialBridge(tau_grid(0.5, N_GRID), P, Pt, [[1.0, 0, 0]], [5 / 128], [[1e-10]], ctx=ctx)


def pc_shape(P):
    """workgroup shape the wave-specialised kernel is launched with under the default noise specification (bhip_pc_kernel.h launch_pc):
    template arguments `NPAIR, PPR, RLDS` -- 2 / 4 pairs with the coefficient rows through LDS for small ensembles, four pairs sharing
    one table of the generator without them beyond 65 536 chains"""
    g = (P + 63) // 64
    return "2, false, true" if g <= 512 else "4, false, true" if g <= 1024 else "4, false, false"


def fresh_small(margs, P):
    """fresh proposals of a small ensemble: the wave-specialised kernel"""
    return f"k_pc<{margs}, 6, 1, {pc_shape(P)}>"


# mode -> (proposal builder, d, m', x0, default paths, chains?, rho, workload text, kernel-name builder)
MODES = {
    "mcmc": (build_proposal, 2, 1, X0, 262144, True, RHO,
             FHN_WORKLOAD + "pCN-MCMC rho=0.9: one step = one MH iteration of every chain",
             lambda P: f"k_pc<bhip::MFHN, 2, 1, 7, 1, {pc_shape(P)}>"),
    "c4shard": (build_proposal, 2, 1, X0, 32768, True, RHO,
                FHN_WORKLOAD + "pCN-MCMC rho=0.9, SURVEY-C4 shard size (32 768 chains per GPU)",
                lambda P: f"k_pc<bhip::MFHN, 2, 1, 7, 1, {pc_shape(P)}>"),
    "proposals": (build_proposal, 2, 1, X0, 262144, False, None,
                  FHN_WORKLOAD + "independent fused proposals (sample!+solve!+llikelihood)",
                  lambda P: (fresh_small("bhip::MFHN, 2, 1", P) if P <= 98304 else "k_paths<bhip::MFHN, 2, 1, 1, 1, false>")),
    "c2": (_ou, 1, 1, (0.5,), 65536, False, None,
           "C2: 1-d OU target LinPro(-0.8,0,sqrt(.7)), GuidedBridge with auxiliary LinPro(-0.8,0.2,sqrt(.7)), 1001-point tau-grid T=2, "
           "independent fused proposals",
           lambda P: (fresh_small("bhip::MLinPro<1, double const*>, 1, 1", P) if P <= 98304 else "k_paths<bhip::MLinPro<1, double const*>, 1, 1, 1, 1, false>")),
    "nclar": (_nclar, 3, 1, (0.0, 0.0, 0.0), 262144, False, None,
              "NCLAR 3-d PartialBridge (scalar noise, L=[1 0 0], v=5/128), 1001-point tau-grid T=0.5, independent fused proposals",
              lambda P: (fresh_small("bhip::MNCLAR, 2, 1", P) if P <= 98304 else "k_paths<bhip::MNCLAR, 2, 1, 1, 1, false>")),
    "nclar_mcmc": (_nclar, 3, 1, (0.0, 0.0, 0.0), 262144, True, 0.95,
                   "NCLAR 3-d PartialBridge (scalar noise, L=[1 0 0], v=5/128), 1001-point tau-grid T=0.5, pCN-MCMC rho=0.95",
                   lambda P: f"k_pc<bhip::MNCLAR, 2, 1, 7, 1, {pc_shape(P)}>"),
    "linpro4": (lambda ctx: _linpro(ctx, 4), 4, 4, tuple([0.0] * 4), 262144, False, None,
                "LinPro d=4 GuidedBridge (dense sigma, pre-inverted Hdiamond), 1001-point grid T=1, independent fused proposals: one path per lane "
                "(dimensions 4..8 stay off the matrix cores)",
                lambda P: "k_paths<bhip::MLinPro<4, double const AS4*>, 5, 1, 1, 1, false>"),
    "linpro32": (_linpro32, 32, 32, tuple([0.0] * 32), 65536, False, None,
                 "LinPro d=32 GuidedBridge (dense sigma, pre-inverted Hdiamond), 1001-point grid T=1, independent fused proposals",
                 lambda P: "k_tile<32, 1, false, bhip::NoUserDrift, false>"),
    "linpro32_mcmc": (_linpro32, 32, 32, tuple([0.0] * 32), 65536, True, 0.95,
                      "LinPro d=32 GuidedBridge (dense sigma, pre-inverted Hdiamond), 1001-point grid T=1, pCN-MCMC rho=0.95",
                      lambda P: "k_tile<32, 2, false, bhip::NoUserDrift, false>"),
}


class Workload:
    """one bench mode: owns its device buffers; step() = one launch of the dominant kernel.
    `<mode>_fused`: the same workload under BHIP_OPT_FUSED_ARITHMETIC (the d <= 3 kernels built with a*b + c contracted: results to
    1e-9 / 1e-8 instead of bit for bit, tests/test_gpu_fused.py) on a context of its own."""

    def __init__(self, mode, ctx, chains, rank):
        self.parts = None   # fresh proposals: the container decides (EnsemblePath: two buffers in different pieces of the device memory from 1 GiB on)
        if mode.endswith("_parts"):   # ... X kept in two buffers whatever its size
            mode, self.parts = mode[:-len("_parts")], 2
        if mode.endswith("_1buf"):    # ... X in ONE buffer whatever its size (the default of rounds 1-5)
            mode, self.parts = mode[:-len("_1buf")], 1
        self.fused = mode.endswith("_fused")
        if self.fused:
            mode = mode[:-len("_fused")]
            ctx = bh.Context(ctx.device.index)
            ctx.set_option(bh.OPT_FUSED_ARITHMETIC, 1)
        self.v2noise = 0
        for spec in (2, 3):   # the same workload under an earlier noise specification, bhip-philox-v2 / -v3 (BHIP_OPT_NOISE_SPEC = 2 / 3)
            if mode.endswith(f"_v{spec}noise"):
                self.v2noise = spec
                mode = mode[:-len("_v2noise")]
                ctx = bh.Context(ctx.device.index)
                ctx.set_option(bh.OPT_NOISE_SPEC, spec)
        build, d, mp, x0, default_P, is_chains, rho, text, kname = MODES[mode]
        self.mode, self.ctx = mode, ctx
        self.P = chains if chains else default_P
        self.path0 = rank * self.P                   # contiguous shard of the global ids; the RNG is keyed by the global id
        self.Po = build(ctx)
        self.workload = text + (" [BHIP_OPT_FUSED_ARITHMETIC: tolerance parity 1e-9 / 1e-8]" if self.fused else "") + \
            (" [BHIP_OPT_NOISE_SPEC = 2: bhip-philox-v2, one Box-Muller pair of 53 + 53 bits per Philox call]" if self.v2noise == 2 else
             " [BHIP_OPT_NOISE_SPEC = 3: bhip-philox-v3, two Box-Muller pairs of 40 + 24 bits per Philox call]" if self.v2noise == 3 else "") + \
            (" [X kept in two buffers of half the paths each, in different 96-GiB pieces of the device memory: bhip_alloc_apart + bhip_sample_solve_parts, "
             "one launch, the same values]" if self.parts == 2 else " [X in ONE buffer]" if self.parts == 1 else "")
        self.kernel = kname(self.P).replace("bhip::", "bhip_fused::") if self.fused else kname(self.P)
        if self.fused:   # LinPro targets at d <= 3 run the regrouped step (GUIDE_QF = 5) under the option (bhip_path_kernel.h)
            self.kernel = self.kernel.replace("double const*>, 1, 1,", "double const*>, 5, 1,")
        if self.v2noise:   # large ensembles under v3 / v2: one pair per workgroup (bhip_pc_kernel.h launch_pc)
            self.kernel = self.kernel.replace("4, false, false>", "1, false, false>")
        self.flops_per_pathstep = 5 * 2 * d * d if d > 8 else None   # d = 32: five d x d mat-vecs per path-step on the matrix cores
        self.chains = None
        if is_chains:
            self.chains = bh.Chains(self.Po, np.array(x0), self.P, seed=4, path0=self.path0, store_X=True)
            self.rho = rho
            self.step = lambda: self.chains.step(rho, 1)
            self.bytes_per_pathstep = 8 * d + 16 * mp    # write Xo (8d) + read W, write Wo (16 m')   SURVEY 8(d) mode M
        else:
            self._fresh(d, np.array(x0), 4)
            self.bytes_per_pathstep = 8 * d              # write X (8d)                               SURVEY 8(d) mode E

    def _fresh(self, d, x0, seed):
        ctx, P = self.ctx, self.P
        self.ll = ctx.empty(P)
        self.it = 0
        self.x0 = np.ascontiguousarray(x0, dtype=np.float64)
        # ONE container type: EnsemblePath keeps an ensemble of 1 GiB and more in two buffers lying in different pieces of the device memory
        # (parts=None), and the fused proposal writes both in one launch
        self.X = X = bh.EnsemblePath(self.Po.tt, d, P, ctx, parts=self.parts)
        self.nparts, self.parts_apart = X.nparts, X.apart
        x0p, llp, lib, h, poh = bh.api._dptr(self.x0), bh.api.vp(self.ll.data_ptr()), ctx.lib, ctx.h, self.Po.h
        if X.nparts > 1:
            def step():
                self.it += 1
                ctx.check(lib.bhip_sample_solve_parts(h, poh, x0p, X.nparts, X._ptrs, X.part_paths, X.part_paths, llp, 0, P, seed, self.it, self.path0))
        else:
            xp = X.ptr()

            def step():
                self.it += 1
                ctx.check(lib.bhip_sample_solve(h, poh, x0p, None, None, P, xp, P, llp, 0, P, seed, self.it, self.path0))
        self.step = step

    def roofline(self, kern_ms):
        """achieved = algorithmic bytes (or flops) per launch / mean launch duration (HIP events: LaunchTimes.avg, the launches back to back
        between two events, where the caller measured it; else the mean of the per-launch durations)"""
        back_to_back = getattr(kern_ms, "avg", None)
        avg_s = float(back_to_back if back_to_back else np.mean(kern_ms)) * 1e-3
        per_launch = float(self.P) * (N_GRID - 1)
        gbs = per_launch * self.bytes_per_pathstep / avg_s / 1e9
        r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
             "kernel": self.kernel, "kernel_avg_ms": avg_s * 1e3, "kernel_min_ms": float(np.min(kern_ms)),
             "kernel_max_ms": float(np.max(kern_ms)), "algorithmic_bytes_per_path_step": self.bytes_per_pathstep,
             "path_steps_per_launch": self.P * (N_GRID - 1)}
        if back_to_back:   # (min / max above are of this second, untimed pass)
            r["per_launch_pass"] = {"launches": len(kern_ms), "avg_ms": float(np.mean(kern_ms)),
                                    "note": "as many launches again with a HIP event between consecutive ones (each record costs the stream 6-10 us); kernel_avg_ms is "
                                            "the average over the launches issued back to back between two events"}
        if self.flops_per_pathstep:   # compute-bound kernel: report against the fp64 matrix-core peak
            tf = per_launch * self.flops_per_pathstep / avg_s / 1e12
            # achieved / frac: the ALGORITHMIC flops (the reference's five d x d mat-vecs per path-step, SURVEY 8(d): 10 240 at d = 32).  Since
            # round 5 the kernel regroups the affine step into THREE products (bhip_tile_kernel.h: the log-likelihood's dot product as one
            # quadratic form, the update as one accumulation): executed flops and the share of the matrix pipe they occupy are stated
            # beside it -- the work left out is work not done, not work done faster
            ex = self.flops_per_pathstep * 3 // 5
            r.update({"bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F64_PEAK_TF,
                      "algorithmic_flops_per_path_step": self.flops_per_pathstep, "executed_flops_per_path_step": ex,
                      "mfma_pipe_frac": tf * 0.6 / MFMA_F64_PEAK_TF, "hbm_algorithmic_GBs": gbs})
        tr, src = (None, "not profiled (fused build / v2 noise)") if (self.fused or self.v2noise) else profiled_traffic(self.mode, self.kernel)
        if tr is not None and self.P != MODES[self.mode][4]:
            tr, src = None, "profiled at the mode's default size only"
        r["traffic"], r["traffic_source"] = tr, src
        r["traffic_box"] = "another box (committed rocprofv3 summaries under profiles/, looked up by kernel name)" if tr is not None else None
        r["valu"] = profiled_valu(self.mode, self.kernel, self.P) if (self.P == MODES[self.mode][4] and not self.fused and not self.v2noise) else None
        v = r["valu"]
        if v and r["bound"] == "hbm" and v["busy_frac"] > 0.6 and r["frac"] / v["busy_frac"] < 0.9:
            # the kernel's SIMDs spend most of its duration ISSUING vector instructions: what binds it is the instruction count
            # per byte, not the memory system.  achieved / peak / frac stay the HBM figures (the contract's yardstick);
            # valu_frac = share of the duration the VALU was issuing, hbm_frac_at_full_issue = the roofline fraction this
            # instruction stream would reach at 100 % issue -- the ceiling that actually applies (a kernel whose ceiling at full
            # issue lies at or above the HBM roofline keeps "hbm": its instructions are not what stops it)
            r["bound"] = "valu"
            r["valu_frac"] = v["busy_frac"]
            r["hbm_frac_at_full_issue"] = r["frac"] / v["busy_frac"]
        return r


class LaunchTimes(list):
    """Per-launch HIP-event durations (an event recorded BETWEEN consecutive launches: min / max / spread) plus `.avg`, the average launch
    duration of as many launches issued BACK TO BACK between two events.  An event record between two kernels costs the stream 6-10 us --
    0.5 % of the headline's launch, 6 % of C2's 150-us one -- that neither rocprofv3's kernel durations nor a caller's loop contain (same
    call, C2: 150.8 us per kernel in the trace, 160 between per-launch events): since the end of round 5 the timed region holds its K
    steps and TWO events, and the roofline's kernel_avg_ms is that region's average; the per-launch pass follows it, untimed."""
    avg = None


def kernel_times(w, steps, warmup, min_ms=0.0):
    """HIP-event durations of `steps` launches (events on the stream the kernels go to: torch's current stream): first the launches back
    to back between two events (-> .avg), then as many with an event between consecutive launches (the list: min / max).
    min_ms: sample at least that long (short launches measured for a few ms right after another workload see the clock
    state that workload left behind, not their own)"""
    for _ in range(warmup):
        w.step()
    if min_ms > 0.0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); w.step(); e1.record(); torch.cuda.synchronize()
        steps = max(steps, min(4000, int(min_ms / max(e0.elapsed_time(e1), 1e-3))))
        # settle the clocks on THIS kernel before sampling -- with a burst as long as the sampled one: the first long burst of launches in a
        # process costs the host up to the kernel's own duration per launch (HIP's command / kernel-argument pools growing; C2, same process:
        # 135 us per launch to issue the first 300, 4.5 us the next -- scripts/gpu_issue_probe2.py), and a 150-us kernel then waits for its host
        for _ in range(steps):
            w.step()
        torch.cuda.synchronize()
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record()
    for k in range(steps):
        w.step()
    b1.record()
    out = per_launch_pass(w, steps)
    out.avg = b0.elapsed_time(b1) / steps
    return out


def per_launch_pass(w, steps):
    """`steps` launches with a HIP event between consecutive ones: the duration of every launch (min / max of the record)"""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for k in range(steps):
        evs[k].record()
        w.step()
    evs[steps].record()
    torch.cuda.synchronize()
    return LaunchTimes(evs[k].elapsed_time(evs[k + 1]) for k in range(steps))


# what the implementation moves per path-step: padded W lines read + written 64, the proposal path into the other parity half 24, mcnext!
# reads the current half 24 and reads + writes its state 192 (rocprofv3 FETCH + WRITE of the loop's kernels: profiles/r4_smoothing_*.txt;
# round 3's plain SoA paths with the commit copy moved 64 + 24 + 66 + 192: every line of Xo AND of Xc read, the lines of Xc written back)
SMOOTH_MOVED_BYTES = 64 + 24 + 24 + 192   # with a statistics pass per iteration (K = 1); see smoothing_record for the deferred form


def smoothing_record(ctx, m=4, M=250, n=32768, reps=10):
    """The application loop around the hot path (supplements/smoothing/smoothing.jl:99-213): Lorenz d = 3,
    m GuidedBridge segments of M steps with LinearAppr auxiliaries, n chains, joint MH + pCN on the start + mcnext! per chain
    and iteration.  Times one iteration with guides shared by the ensemble, the per-chain adaptation on the device
    (bhip_segchains_adapt_device) and one iteration with per-chain guides; algorithmic bytes per path-step as DESIGN 10 states."""
    P = bh.Lorenz((10.0, 20.0, 8 / 3), (3.0, 3.0, 3.0))
    tgrid = np.linspace(0.0, 0.002 * m * M, m * M + 1)
    Y = np.zeros((m * M + 1, 3)); y = np.array([1.5, -1.5, 25.0])
    for i in range(m * M + 1):
        Y[i] = y
        if i < m * M:
            y = y + np.array([10 * (y[1] - y[0]), y[0] * (20 - y[2]) - y[1], y[0] * y[1] - 8 / 3 * y[2]]) * (tgrid[i + 1] - tgrid[i])
    L, Sig = np.eye(3), 0.25 * np.eye(3)
    obs = Y[::M] + 0.5 * np.random.default_rng(0).standard_normal((m + 1, 3))
    HT, vT = bh.gpupdate(1e3 * np.eye(3), np.zeros(3), L, Sig, obs[m])
    H, v, segs = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        segs[i] = bh.GuidedBridge(tgrid[i * M:(i + 1) * M + 1].copy(), P, bh.linearappr(Y[i * M:(i + 1) * M + 1]), v, H, ctx=ctx)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    sc = bh.SegChains(segs, v, bh.cholupper_t(H), n, seed=1, mcnext=True)
    wo, wn = 0.9, math.sqrt(1 - 0.81)

    def t(fn, k):
        fn(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        for j in range(k):
            ev[j].record(); fn()
        ev[k].record(); torch.cuda.synchronize()
        return float(np.mean([ev[j].elapsed_time(ev[j + 1]) for j in range(k)]))
    def t_iters(s, k, rounds=3):
        # `k` iterations in ONE call (the library runs the commit + mcnext! of an iteration on its own stream beside the next
        # iteration's proposals and joins the streams before the call returns: the join is inside the timed region)
        s.step(wo, wn, 2); torch.cuda.synchronize()
        out = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); s.step(wo, wn, k); e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / k)
        return float(np.median(out))
    ps = n * m * M
    K, nbuf = sc.statistics_info()
    ms_sh = t_iters(sc, 2 * reps)
    ms_ad = t(lambda: sc.adapt_device(L, Sig, obs[:m], HT, vT), max(2, reps // 2))
    ms_pc = t_iters(sc, 2 * reps)
    ok = bool(np.isfinite(sc.state()[0]).all())
    del sc
    torch.cuda.empty_cache()
    sm = bh.SegChains(segs, v, bh.cholupper_t(H), n, seed=1, mcnext_mean_only=True)   # the economy option: running means only
    ms_mo = t_iters(sm, 2 * reps)
    del sm
    torch.cuda.empty_cache()
    se = bh.SegChains(segs, v, bh.cholupper_t(H), n, seed=1, mcnext=True, stats_every_iteration=True)   # a statistics pass per iteration, two path buffers
    ms_se = t_iters(se, 2 * reps)
    del se
    torch.cuda.empty_cache()
    # ALGORITHMIC bytes per path-step = what the reference's loop must move (smoothing.jl:160-213 swaps references on accept, it
    # copies nothing): W read + Wo written 48 (m' = 3), Xo written 24, mcnext! reads X 24 and reads + writes its state
    # (mean 3 + m2 9 doubles) 192 = 288.  The implementation MOVES 16 more -- the W lines pad m' = 3 to 4 -- reported beside it as
    # `moved_bytes_per_path_step`; the fractions are computed on the algorithmic count.
    # Since round 4 the library applies mcnext! every K iterations to the K current paths a ring of path buffers has kept
    # (bhip_segchains_statistics_info: same bits, the state travels once per K iterations): what THAT loop must move per path-step is
    # 48 + 24 + 24 (a path read per iteration at most: none for an iteration in which the chain did not move) + 192 / K -- the count the
    # fractions below are computed on; the reference's count stays in the record beside it.
    b_ref = 48 + 24 + 24 + 192
    b_sh = 48 + 24 + 24 + 192 / K
    b_moved = b_sh + 16
    b_pc = b_sh + 120              # + the chain's compact guide row per step: Hd (9), V (3), linearisation point (3)
    return {"workload": f"Lorenz smoothing: {m} GuidedBridge(LinearAppr) segments x {M} steps, {n} chains, joint MH + pCN start + mcnext! per iteration",
            "path_steps_per_iteration": ps, "finite": ok, "statistics_every": K, "path_buffers_per_segment": nbuf,
            "iteration_shared_guides": {"ms": ms_sh, "path_steps_per_s": ps / ms_sh * 1e3, "algorithmic_bytes_per_path_step": b_sh,
                                        "moved_bytes_per_path_step": b_moved, "reference_loop_bytes_per_path_step": b_ref,
                                        "hbm_frac": ps * b_sh / ms_sh / 1e6 / HBM_PEAK_GBS,
                                        "note": f"mcnext! applied every {K} iterations to the {K} current paths kept in a ring of {nbuf} path buffers per segment (same bits as a pass per iteration, "
                                                "which this record timed at 2.15-2.28 ms in round 3 and 1.70-1.77 ms with the paths in parity halves); the iteration is now bound by its four "
                                                "dependent proposal launches (0.68 ms alone at one wave per SIMD), not by HBM"},
            "iteration_shared_guides_stats_every_iteration": {"ms": ms_se, "path_steps_per_s": ps / ms_se * 1e3, "algorithmic_bytes_per_path_step": b_ref,
                                                              "moved_bytes_per_path_step": SMOOTH_MOVED_BYTES, "hbm_frac": ps * b_ref / ms_se / 1e6 / HBM_PEAK_GBS,
                                                              "note": "BHIP_SEGCHAINS_STATS_EVERY_ITERATION: the same loop with mcnext! applied after every iteration (K = 1), the reference's count of bytes"},
            "iteration_shared_guides_means_only": {"ms": ms_mo, "path_steps_per_s": ps / ms_mo * 1e3, "algorithmic_bytes_per_path_step": b_sh - 144 / K,
                                                   "hbm_frac": ps * (b_sh - 144 / K) / ms_mo / 1e6 / HBM_PEAK_GBS,
                                                   "note": "BHIP_SEGCHAINS_MCNEXT_MEAN: the per-chain running means only (all the adaptation reads); mcnext! proper keeps the 3 x 3 second moments too"},
            # one call = the per-chain guide builder (k_seg_guide: 24 B of running mean read + the 120-byte compact row written per
            # chain and grid point) AND the re-evaluation of every chain's current log-likelihood under its new guide (the script
            # evaluates both llikelihoods afresh every iteration; here the current one is cached and re-done on adaptation:
            # 24 B of X + the 120-byte row read per path-step).  profiles/r3_smoothing_trace.txt has the two kernels apart.
            "adapt_device": {"ms": ms_ad, "guide_segments_per_s": n * m / ms_ad * 1e3,
                             "algorithmic_bytes": n * m * ((M + 1) * 144 + M * 144),
                             "algorithmic_bytes_guide_builder": n * m * (M + 1) * 144, "algorithmic_bytes_ll_reevaluation": n * m * M * 144,
                             "hbm_frac": n * m * ((M + 1) * 144 + M * 144) / ms_ad / 1e6 / HBM_PEAK_GBS},
            "iteration_per_chain_guides": {"ms": ms_pc, "path_steps_per_s": ps / ms_pc * 1e3, "algorithmic_bytes_per_path_step": b_pc,
                                           "hbm_frac": ps * b_pc / ms_pc / 1e6 / HBM_PEAK_GBS}}


def box_calibration(device):
    """What THIS box's memory system does on the plainest streams (boxes of the pool differ: the headline kernel moves exactly
    its algorithmic bytes and ran at 1.53-1.62 ms on some, 1.80-1.84 ms on others): device-to-device copy (1 read : 1 write)
    and fill (write only) of 1 GiB, GB/s of bytes moved; plus the partition modes and clocks rocm-smi reports."""
    import subprocess
    n = 1 << 27
    a = torch.empty(n, dtype=torch.float64, device=device)
    b = torch.empty(n, dtype=torch.float64, device=device)
    a.fill_(1.0); b.copy_(a); torch.cuda.synchronize()

    def t(fn, k=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k * 1e-3
    tc, tf = t(lambda: b.copy_(a)), t(lambda: a.fill_(2.0))
    out = {"copy_1r1w_GBs": 2 * 8 * n / tc / 1e9, "fill_write_GBs": 8 * n / tf / 1e9}
    del a, b
    torch.cuda.empty_cache()
    try:
        r = subprocess.run(["rocm-smi", "--showmemorypartition", "--showcomputepartition", "--showclocks"], capture_output=True, text=True, timeout=20)
        for key, pat in (("compute_partition", "Compute Partition:"), ("memory_partition", "Memory Partition:"), ("mclk", "mclk clock level"), ("fclk", "fclk clock level")):
            for line in r.stdout.splitlines():
                if pat in line:
                    out[key] = line.split(":", 2)[-1].strip() if "clock" not in pat else line.split(":")[-1].strip(" ()")
                    break
    except Exception as e:   # the record is context, never a reason to fail the bench
        out["rocm_smi"] = f"unavailable: {e}"
    return out


def timed_region(w, steps, world, ctx, stats, comm=None):
    """the contract's timed region (one process per GPU, launched by torch.distributed.run): K steps bracketed by barrier +
    synchronize on both sides, MAX over ranks; ends with the device-side statistics reduction and the ONE all-gather of the
    statistics block"""
    prewarm([w])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for k in range(steps):
        w.step()
    e1.record()
    if w.chains is not None:
        w.chains.stats(stats)
    else:
        stats.zero_()
    gathered = bdist.allgather_stats(stats, world, comm)   # N > 1: bhip_comm_allgather_stats (RCCL inside the library)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kern = per_launch_pass(w, steps)          # untimed: the same number of launches again, an event between consecutive ones
    kern.avg = e0.elapsed_time(e1) / steps
    return elapsed, kern, gathered


PREWARM_S = 0.5   # see prewarm(): what the step counts below amount to at the bench sizes
PREWARM_STEPS = {"linpro32": 40, "linpro32_mcmc": 32}   # every other mode: 300


def prewarm(ws):
    """Untimed, after the W warm-up steps and every other set-up, immediately before the synchronisation that opens the timed region:
    about half a second of the workload's own step QUEUED on the devices -- a fixed number of steps per mode, so that the chains' state
    at the start of the timed region does not depend on the machine or on how the chains are sharded -- so that the timed steps start
    on devices that have been busy up to that synchronisation.  The first process on an idle box otherwise measures the clocks' way up
    -- 1.55-1.57 ms per step of the headline where every later process of the same box reads 1.505-1.515 (thirty fresh processes on five
    boxes; it is the idle gap before the timed region that counts: a second of such steps FOLLOWED by the set-up of the collective left
    the first process at 1.56-1.57, 400 queued warm-up steps that were still running when the set-up ended put it at 1.511).  Not part
    of W, not part of the timed region; the record says so (config.prewarm_steps)."""
    for _ in range(PREWARM_STEPS.get(ws[0].mode, 300)):
        for x in ws:
            x.step()


def timed_region_local(ws, steps, stats, group):
    """the same region with ONE process driving len(ws) devices (no launcher): synchronize every device, issue the K steps
    round-robin (launches are asynchronous: one host thread keeps all devices busy), the per-device statistics reductions and
    the ONE grouped all-gather (bhip_comm_allgather_group), synchronize every device.  The wall clock around it is by
    construction the max over the devices.  Returns (elapsed s, per-launch ms of every device, gathered blocks,
    per-device ms for the K steps, ms of the gather on device 0, host microseconds per iteration inside the group step call)."""
    devs = [w.ctx.device for w in ws]
    streams = [torch.cuda.default_stream(d) for d in devs]
    n = len(ws)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] for _ in range(n)]
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # chain ensembles: ONE C-ABI call per iteration steps every device (bhip_chains_step_group), one reduces every device's
    # statistics (bhip_chains_stats_group); host_issue = host time inside those step calls, per iteration
    grp = bdist.ChainsGroup([w.chains for w in ws]) if all(w.chains is not None for w in ws) else None
    host_issue = 0.0
    prewarm(ws)
    for d in devs:
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    b0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    b1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for r in range(n):
        b0[r].record(streams[r])
    for k in range(steps):
        if grp is not None:
            ti = time.perf_counter()
            grp.step(ws[0].rho, 1)
            host_issue += time.perf_counter() - ti
        else:
            for r in range(n):
                ws[r].step()
    for r in range(n):
        b1[r].record(streams[r])
    if grp is not None:
        grp.stats(stats)
    else:
        for r in range(n):
            with torch.cuda.device(devs[r]):
                stats[r].zero_()
    g0.record(streams[0])
    gathered = group.allgather(stats) if group is not None else [stats[0].reshape(1, -1)]
    g1.record(streams[0])
    for d in devs:
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0
    # untimed: the same number of iterations again with an event between consecutive launches on every device (min / max per launch)
    for k in range(steps):
        for r in range(n):
            evs[r][k].record(streams[r])
        if grp is not None:
            grp.step(ws[0].rho, 1)
        else:
            for r in range(n):
                ws[r].step()
    for r in range(n):
        evs[r][steps].record(streams[r])
    for d in devs:
        torch.cuda.synchronize(d)
    kern = [LaunchTimes(evs[r][k].elapsed_time(evs[r][k + 1]) for k in range(steps)) for r in range(n)]
    per_dev = [b0[r].elapsed_time(b1[r]) for r in range(n)]
    for r in range(n):
        kern[r].avg = per_dev[r] / steps
    return (elapsed, kern, gathered[0], per_dev, g0.elapsed_time(g1),
            host_issue / steps * 1e6 if grp is not None else None)


class _HostGather:
    """BENCH_SAME_DEVICE test double of dist.CommGroup: the blocks of all "ranks" stacked (they live on one device)"""

    def __init__(self, n):
        self.nranks = n

    def allgather(self, sends):
        g = torch.stack([s.reshape(-1) for s in sends])
        return [g.clone() for _ in sends]

    def destroy(self):
        pass


def base_record(args, world, w, elapsed, kern_ms, launch):
    steps_per_unit = N_GRID - 1
    total_pathsteps = float(world) * w.P * steps_per_unit * args.steps
    return {
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
        "config": {"workload": w.workload, "mode": args.mode, "paths_per_gpu": w.P, "grid_points": N_GRID,
                   "path_steps_per_step": w.P * steps_per_unit * world,
                   "noise_spec": NOISE_SPEC, "noise_spec_short": SHORT_NOISE[4],
                   "workload_short": f"FitzHugh-Nagumo PartialBridge d=2 m'=1 (v=1.1, Sigma=1e-10), 1001-pt tau-grid T=2, {'pCN-MCMC rho=0.9' if w.chains is not None else 'fresh proposals'}"
                                     if w.mode in ("mcmc", "c4shard", "proposals") else w.workload,
                   "parallelism_short": f"chains sharded over {world} GPU(s) by contiguous global id; one RCCL all-gather of a 64-B statistics block",
                   "prewarm_steps": PREWARM_STEPS.get(args.mode, 300),   # untimed, queued right before the timed region opens: the same step (clocks of an idle box)
                   "parallelism": f"chains sharded over {world} GPU(s) by contiguous global id, no data-path collective, one RCCL all-gather "
                                  "of the 64-byte statistics block inside libbridgehip.so",
                   "launch": launch},
        "roofline": w.roofline(kern_ms),
    }


def rccl_version_string(v):
    """ncclGetVersion's integer (major*10000 + minor*100 + patch since 2.9; major*1000 + minor*100 + patch before)"""
    v = int(v)
    return f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v >= 10000 else f"{v // 1000}.{v // 100 % 10}.{v % 100}"


def comm_record(infos, gathered, backend="rccl", note=None):
    """The communicator in the record: what the library and RCCL ITSELF (ncclCommCount / ncclCommUserRank / ncclGetVersion through
    bhip_comm_query) report, rank by rank, and how many statistics blocks the gather delivered -- a SCALE line answers "did RCCL see N
    ranks" by itself.  infos: one dict per rank this process knows about (all of them in the one-process form; in the per-rank form the
    ranks' own reports, gathered)."""
    rec = {"backend": backend, "gathered_blocks": int(gathered.shape[0]) if gathered is not None else None}
    if infos:
        rec.update({"nranks": infos[0]["nranks"], "rccl_nranks": infos[0]["rccl_nranks"],
                    "ranks_seen": sorted(i["rccl_rank"] for i in infos), "rccl_version": rccl_version_string(infos[0]["rccl_version"]),
                    "consistent": all(i["nranks"] == i["rccl_nranks"] == len(infos) and i["rank"] == i["rccl_rank"] for i in infos)
                                  and sorted(i["rccl_rank"] for i in infos) == list(range(len(infos)))})
    if note:
        rec["note"] = note
    return rec


def add_chain_summary(out, gathered, w=None):
    if w is not None and w.chains is not None:
        # BHIP_OPT_TUNE_PLACEMENT (setup, before any timing): Xo allocations timed, ms per iteration of the same-piece reference and of the kept pair
        out["config"]["placement"] = w.chains.placement()
    summary = bdist.combine_stats(gathered)
    out["config"]["acceptance_rate"] = summary["acceptance_rate"]
    out["config"]["mean_ll"] = summary["mean_ll"]
    out["config"]["chains_total"] = summary["chains"]


def main_per_rank(args, world):
    """one process per GPU under torch.distributed.run (the driver's N > 1 form)"""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # BENCH_SINGLE_DEVICE=1 (testing the N > 1 code path on a one-GPU box): every rank uses GPU 0 and the collectives
    # go over gloo, because RCCL refuses two ranks on one device.  Never set by the driver.
    single = os.environ.get("BENCH_SINGLE_DEVICE") == "1"
    if single:
        local = 0
    elif local >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} needs device {local} but only {torch.cuda.device_count()} device(s) are visible")
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if single:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    ctx = bh.Context(local)
    # the data-path collective lives in the product: an RCCL communicator over the ranks (torch.distributed only carries
    # the id handshake, the barriers and the max-over-ranks of the clock)
    comm, comm_note = None, None
    stats = ctx.empty(bh.STATS_LEN)
    stats.zero_()
    if not single:
        # set the communicator up and run its first collective (untimed: it builds RCCL's channels over xGMI).  Should the
        # product's communicator fail on ANY rank, every rank falls back to torch.distributed's all-gather (RCCL as well) and the
        # record says so: a scaling run is never lost to the handshake.
        ok = 1
        try:
            if os.environ.get("BENCH_FORCE_COMM_FAILURE") == "1":   # test hook: exercise the fall-back below
                raise RuntimeError("forced by BENCH_FORCE_COMM_FAILURE")
            comm = bdist.Comm.from_torch_dist(ctx)
            bdist.allgather_stats(stats, world, comm)
        except Exception as e:   # noqa: BLE001 -- anything here is reported, not fatal
            ok, comm_note = 0, f"{type(e).__name__}: {e}"
        flag = torch.tensor([ok], dtype=torch.int32, device=ctx.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm is not None:
                try:
                    comm.destroy()
                except Exception:   # noqa: BLE001
                    pass
            comm = None
            comm_note = comm_note or "the product communicator failed on another rank"
            print(f"bench.py[{rank}]: bhip_comm_init_rank path unavailable ({comm_note}); statistics gathered by torch.distributed", file=sys.stderr)
    w = Workload(args.mode, ctx, args.chains, rank)
    steps_per_unit = N_GRID - 1

    for _ in range(args.warmup):
        w.step()
    # the protocol of rounds 1-3 beside the headline (same steps in both launch forms: two ranks == one rank with twice the chains)
    torch.cuda.synchronize()
    t_np = time.perf_counter()
    for _ in range(args.steps):
        w.step()
    torch.cuda.synchronize()
    no_prewarm_ms = (time.perf_counter() - t_np) / args.steps * 1e3
    stats.zero_()
    bdist.allgather_stats(stats, world, comm)
    elapsed, kern_ms, gathered = timed_region(w, args.steps, world, ctx, stats, comm)
    # every rank's own device time over the timed steps (HIP events), so that a straggler shows in the record
    mine = torch.tensor([float(kern_ms.avg * args.steps)], dtype=torch.float64, device=ctx.device)
    allms = torch.empty(world, dtype=torch.float64, device=ctx.device)
    if world > 1:
        dist.all_gather_into_tensor(allms, mine)
    else:
        allms.copy_(mine)
    per_gpu_ms = [float(x) for x in allms.cpu()]

    # the communicator's own report, from every rank (gathered as python objects: outside every timed region)
    my_info = comm.info() if comm is not None else None
    infos = [None] * world
    if world > 1:
        dist.all_gather_object(infos, my_info)
    else:
        infos = [my_info]
    out = None
    if rank == 0:
        how = ("bhip_comm_init_rank + bhip_comm_allgather_stats" if comm is not None else
               "statistics all-gather by torch.distributed" + (f" (product communicator: {comm_note})" if comm_note else " (gloo test double)"))
        out = base_record(args, world, w, elapsed, kern_ms, "one process per GPU (torch.distributed.run); " + how)
        out["per_gpu_ms_per_step"] = [t / args.steps for t in per_gpu_ms]
        out["no_prewarm"] = {"ms_per_step": no_prewarm_ms, "note": f"rank 0's own clock over the same {args.steps} steps right after the warm-up steps, before the pre-warm; not the headline"}
        if comm is not None and all(i is not None for i in infos):
            out["comm"] = comm_record(infos, gathered, "rccl", "bhip_comm_init_rank, one process per GPU")
        else:
            out["comm"] = comm_record(None, gathered, "gloo (test double)" if single else "torch.distributed nccl (fall-back)", comm_note)
        if w.chains is not None:
            add_chain_summary(out, gathered, w)
    if args.mode == "mcmc" and args.chains == 0:
        # SURVEY 8(d) C4 quotes 32 768 chains per GPU: the same protocol at that shard size, next to the headline
        del w
        torch.cuda.empty_cache()
        wc = Workload("c4shard", ctx, 0, rank)
        for _ in range(args.warmup):
            wc.step()
        el_c, ms_c, _ = timed_region(wc, args.steps, world, ctx, stats, comm)
        if rank == 0:
            tp = float(world) * wc.P * steps_per_unit * args.steps
            out["survey_c4"] = {"chains_per_gpu": wc.P, "value": tp / el_c, "unit": "path-steps/s", "ms_per_step": el_c / args.steps * 1e3,
                                "scaling": "weak", "roofline": wc.roofline(ms_c)}
    if rank == 0:
        emit_json(out)
    if comm is not None:
        comm.destroy()
    dist.destroy_process_group()


def main_local(args):
    """`python bench.py --gpus N` without a launcher: ONE process drives the N devices through one context each, the
    communicator comes from bhip_comm_init_all and the gather is bhip_comm_allgather_group -- at N = 1 a world of one through
    the very same calls, so the single-GPU line exercises the collective path too."""
    n = args.gpus
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    # BENCH_SAME_DEVICE=1 (testing the N > 1 code of this function on a one-GPU box): N contexts on device 0, the gather emulated
    # on the host, because RCCL refuses two ranks on one device.  Never set by the driver.
    same = os.environ.get("BENCH_SAME_DEVICE") == "1"
    if n < 1 or (n > ndev and not (same and ndev >= 1)):
        print(f"bench.py: --gpus {n} but only {ndev} device(s) visible to this process", file=sys.stderr)
        sys.exit(2)
    ctxs = [bh.Context(0 if same else k) for k in range(n)]
    if os.environ.get("BENCH_CALIB_COPY") == "1":
        calib_copy()
    comm_note = None
    try:
        group = _HostGather(n) if same and n > 1 else bdist.CommGroup(ctxs)
    except Exception as e:
        if n > 1:
            raise          # no collective, no multi-GPU line
        group, comm_note = None, f"RCCL communicator unavailable on this box ({e}); a world of one needs no exchange"
        print("bench.py: " + comm_note, file=sys.stderr)
    ws = [Workload(args.mode, ctxs[k], args.chains, k) for k in range(n)]
    w = ws[0]
    P = w.P
    steps_per_unit = N_GRID - 1
    stats = [c.empty(bh.STATS_LEN) for c in ctxs]
    for _ in range(args.warmup):
        for x in ws:
            x.step()
    # the protocol of rounds 1-3 beside the headline (advisor r4: round-to-round comparability): K steps right after the W warm-up
    # steps, NO pre-warm -- on an idle box this reads the clocks' way up.  Wall clock around synchronised devices, like the timed region.
    for c in ctxs:
        torch.cuda.synchronize(c.device)
    t_np = time.perf_counter()
    for _ in range(args.steps):
        for x in ws:
            x.step()
    for c in ctxs:
        torch.cuda.synchronize(c.device)
    no_prewarm_ms = (time.perf_counter() - t_np) / args.steps * 1e3
    if group is not None:   # untimed: the first collective sets up RCCL's channels
        for k, c in enumerate(ctxs):
            with torch.cuda.device(c.device):
                stats[k].zero_()
        group.allgather(stats)
    elapsed, kern, gathered, per_gpu_ms, gather_ms, issue_us = timed_region_local(ws, args.steps, stats, group)
    kern_ms = kern[0]
    if n > 1:
        kern_ms = LaunchTimes(float(np.mean([kern[r][k] for r in range(n)])) for k in range(args.steps))
        kern_ms.avg = float(np.mean([kern[r].avg for r in range(n)]))
    out = base_record(args, n, w, elapsed, kern_ms,
                      "one process, one context per device; bhip_comm_init_all + bhip_comm_allgather_group" + (f" [{comm_note}]" if comm_note else ""))
    out["per_gpu_ms_per_step"] = [t / args.steps for t in per_gpu_ms]
    out["no_prewarm"] = {"ms_per_step": no_prewarm_ms, "value": float(n) * P * steps_per_unit / (no_prewarm_ms * 1e-3),
                         "note": f"the same {args.steps} steps right after the {args.warmup} warm-up steps, before the pre-warm (the protocol of rounds 1-3): "
                                 "first process on an idle box = clocks on their way up; not the headline"}
    out["allgather_ms"] = gather_ms
    if isinstance(group, bdist.CommGroup):
        out["comm"] = comm_record(group.info(), gathered, "rccl", "bhip_comm_init_all, one process")
    else:
        out["comm"] = comm_record(None, gathered, "host emulation (BENCH_SAME_DEVICE test double)" if group is not None else "none", comm_note)
    if issue_us is not None:   # host time inside bhip_chains_step_group per iteration (all n devices' launches: one FFI crossing)
        out["host_issue_us_per_step"] = issue_us
    if w.chains is not None:
        add_chain_summary(out, gathered, w)
    ctx = ctxs[0]
    default_run = args.mode == "mcmc" and args.chains == 0
    if n == 1 and default_run and not args.no_other_modes:
        # >= 1 s of back-to-back launches of the headline kernel (the timed region above is K = 20 launches = 30-40 ms)
        n_sus = max(50, int(1.2e3 / max(float(kern_ms.avg), 1e-3)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            w.step()
        torch.cuda.synchronize()
        t_sus = time.perf_counter() - t0
        out["sustained"] = {"launches": n_sus, "seconds": t_sus, "ms_per_step": t_sus / n_sus * 1e3,
                            "path_steps_per_s": P * steps_per_unit * n_sus / t_sus,
                            "hbm_frac": P * steps_per_unit * n_sus * w.bytes_per_pathstep / t_sus / 1e9 / HBM_PEAK_GBS}
        # kernel-level figures of the other named configurations (outside the timed region above)
        others = []
        del w, ws
        torch.cuda.empty_cache()
        for mode in ("c4shard", "c2", "proposals", "nclar", "nclar_mcmc", "linpro4", "linpro32", "linpro32_mcmc", "c2_fused", "proposals_fused", "nclar_fused", "c4shard_fused", "proposals_1buf", "c2_parts", "c2_fused_parts",
                     "mcmc_v3noise", "proposals_v3noise", "c2_v3noise", "mcmc_v2noise", "proposals_v2noise", "c2_v2noise"):
            wo = Workload(mode, ctx, 0, 0)
            ms = kernel_times(wo, args.steps, args.warmup, min_ms=100.0)
            others.append({"mode": mode, "workload": wo.workload, "paths": wo.P,
                           "path_steps_per_s": wo.P * steps_per_unit / (float(ms.avg) * 1e-3), "roofline": wo.roofline(ms)})
            if getattr(wo, "nparts", 1) > 1:
                others[-1]["parts"] = {"n": wo.nparts, "pairwise_apart": wo.parts_apart}
                wo.X.free()
            del wo
            torch.cuda.empty_cache()
            if mode in ("c4shard", "c2", "proposals") and not args.no_live_traffic:
                # the named small configurations' HBM bytes from THIS box's counters as well (two rocprofv3 child runs each, outside
                # every timed region), like the headline's below; the committed look-up stays as the fall-back
                ro = others[-1]["roofline"]
                tr, src = live_traffic(mode, ro["kernel"])
                if tr is not None:
                    ro.update({"traffic": tr, "traffic_source": src, "traffic_box": "this box, this run",
                               "traffic_over_algorithmic": tr / (ro["algorithmic_bytes_per_path_step"] * ro["path_steps_per_launch"])})
                else:
                    ro["traffic_live_failed"] = src
        out["other_modes"] = others
        out["smoothing"] = smoothing_record(ctx)
        out["box"] = box_calibration(ctx.device)
        if not args.no_live_traffic:
            # HBM bytes of the headline kernel from THIS box's counters (two rocprofv3 child runs, outside every timed region);
            # the committed look-up stays as the fall-back and says which box it came from
            tr, src = live_traffic("mcmc", out["roofline"]["kernel"])
            if tr is not None:
                out["roofline"].update({"traffic": tr, "traffic_source": src, "traffic_box": "this box, this run",
                                        "traffic_over_algorithmic": tr / (out["roofline"]["algorithmic_bytes_per_path_step"] * out["roofline"]["path_steps_per_launch"])})
            else:
                out["roofline"]["traffic_live_failed"] = src
    elif n > 1 and default_run:
        # SURVEY 8(d) C4 quotes 32 768 chains per GPU: the same protocol at that shard size, next to the headline
        del w, ws
        for c in ctxs:
            with torch.cuda.device(c.device):
                torch.cuda.empty_cache()
        wcs = [Workload("c4shard", ctxs[k], 0, k) for k in range(n)]
        for _ in range(args.warmup):
            for x in wcs:
                x.step()
        el_c, kern_c, _, pg_c, _, issue_c = timed_region_local(wcs, args.steps, stats, group)
        tp = float(n) * wcs[0].P * steps_per_unit * args.steps
        out["survey_c4"] = {"chains_per_gpu": wcs[0].P, "value": tp / el_c, "unit": "path-steps/s", "ms_per_step": el_c / args.steps * 1e3,
                            "scaling": "weak", "per_gpu_ms_per_step": [t / args.steps for t in pg_c],
                            "roofline": wcs[0].roofline([float(np.mean([kern_c[r][k] for r in range(n)])) for k in range(args.steps)]),
                            "host_issue_us_per_step": issue_c,
                            "note": "one host thread issues the launches of all devices through ONE bhip_chains_step_group call per iteration; "
                                    "host_issue_us_per_step against ms_per_step says how far from launch-bound it is"}
        del wcs
    if n == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if "other_modes" in out:
        # the whole record once more, compact and LAST on the line (a 2 000-character tail of it still holds every mode):
        # mode -> [ms per launch, fraction of its roofline (8 TB/s; fp64 matrix peak for linpro32*), what binds it]
        r = out["roofline"]
        def entry(ro):
            e = [round(ro["kernel_avg_ms"], 4), round(ro["frac"], 3), ro["bound"]]
            if "mfma_pipe_frac" in ro:   # d = 32: frac is on the reference's flop count; the share of the matrix pipe the executed flops occupy beside it
                e.append(round(ro["mfma_pipe_frac"], 3))
            return e
        modes = {"mcmc": entry(r),
                 "mcmc_sustained": [round(out["sustained"]["ms_per_step"], 4), round(out["sustained"]["hbm_frac"], 3), "hbm"]}
        for o in out["other_modes"]:
            modes[o["mode"]] = entry(o["roofline"])
            if o["mode"] == "mcmc_v2noise":   # the headline workload on the full-resolution stream (53 + 53-bit Box-Muller), in the head of the line
                out["headline_v2noise"] = {"ms_per_step": o["roofline"]["kernel_avg_ms"], "frac": o["roofline"]["frac"],
                                           "path_steps_per_s": o["path_steps_per_s"], "noise_spec": SHORT_NOISE[2]}
        sm = out.get("smoothing") or {}
        for key, short in (("iteration_shared_guides", "smooth_shared"), ("iteration_shared_guides_stats_every_iteration", "smooth_shared_k1"),
                           ("iteration_shared_guides_means_only", "smooth_means"),
                           ("adapt_device", "smooth_adapt"), ("iteration_per_chain_guides", "smooth_perchain")):
            if key in sm:
                modes[short] = [round(sm[key]["ms"], 4), round(sm[key]["hbm_frac"], 3), "hbm"]
        out["modes"] = modes
    emit_json(out)
    if group is not None:
        group.destroy()


_JSON_FD = None
CONTRACT_MAX_BYTES = 6144      # the ONE line on stdout; everything else goes to bench_full.json and stderr (a 25-30 KB line was not parsed)
CONTRACT_STR_MAX = 118         # the driver's record cuts strings at ~120 characters: say it shorter, not truncated
FULL_RECORD = "bench_full.json"
SHORT_NOISE = {4: "bhip-philox-v4: Philox4x32-10, one normal per 32-bit word (piecewise deg-4 inverse CDF, |z|<=6.34)",
               3: "bhip-philox-v3: Philox4x32-10, two Box-Muller pairs of 40+24 bits per call",
               2: "bhip-philox-v2: Philox4x32-10, one Box-Muller pair of 53+53 bits per call"}


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries below us write there too (RCCL prints a version banner at the first
    communicator, flushed from C buffers at exit): from here on file descriptor 1 points to stderr, and only emit_json() writes
    to the real stdout."""
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)


def _short(s, n=CONTRACT_STR_MAX):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _num(x, sig=6):
    """numbers of the contract line to `sig` significant digits (a 17-digit double says nothing a 6-digit one does not, at 3x the bytes)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _num(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def contract_record(out):
    """The record the driver parses, cut from the full one: the contract's keys, `config`, `roofline` and `cpu_baseline` with what a
    consistency check needs, `comm`, `sustained`, the headline under the full-resolution noise stream, and the compact `modes` map
    (mode -> [ms per launch, fraction of its roofline, what binds it (, share of the matrix pipe)]).  Strings are short by
    construction (<= CONTRACT_STR_MAX), numbers carry 6 significant digits, the whole line is <= CONTRACT_MAX_BYTES -- asserted."""
    rec = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = out.get("config", {})
    c = _pick(cfg, ("mode", "paths_per_gpu", "grid_points", "path_steps_per_step", "prewarm_steps", "acceptance_rate", "mean_ll", "chains_total"))
    c["workload"] = _short(cfg.get("workload_short") or cfg.get("workload", ""))
    c["noise_spec"] = _short(cfg.get("noise_spec_short") or cfg.get("noise_spec", ""))
    c["parallelism"] = _short(cfg.get("parallelism_short") or cfg.get("parallelism", ""))
    c["launch"] = _short(cfg.get("launch", ""))
    if "placement" in cfg:
        c["placement"] = _pick(cfg["placement"], ("tries", "gbs_same_piece", "gbs_kept", "piece_w", "piece_xo"))
    rec["config"] = c
    ro = out.get("roofline", {})
    r = _pick(ro, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "kernel_avg_ms", "kernel_min_ms",
                   "kernel_max_ms", "algorithmic_bytes_per_path_step", "path_steps_per_launch", "algorithmic_flops_per_path_step",
                   "executed_flops_per_path_step", "mfma_pipe_frac", "valu_frac", "hbm_frac_at_full_issue"))
    if "traffic_box" in ro:
        r["traffic_source"] = _short("rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, " + ("this box, this run" if ro.get("traffic_box") == "this box, this run"
                                                                                           else "committed profiles/ (another box)" if ro.get("traffic") else "none"))
    rec["roofline"] = r
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        rec["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "value_1thread", "host_cpus", "cgroup_cpu_quota"))
        rec["cpu_baseline"]["sample"] = _short(cb.get("sample_short") or cb.get("sample", ""))
    if "comm" in out:
        rec["comm"] = {k: (_short(v) if isinstance(v, str) else v) for k, v in out["comm"].items()}
    for k in ("per_gpu_ms_per_step", "allgather_ms", "host_issue_us_per_step"):
        if k in out:
            rec[k] = out[k]
    if "no_prewarm" in out:
        rec["no_prewarm_ms_per_step"] = out["no_prewarm"].get("ms_per_step")
    if "sustained" in out:
        rec["sustained"] = _pick(out["sustained"], ("launches", "seconds", "ms_per_step", "path_steps_per_s", "hbm_frac"))
    if "survey_c4" in out:
        sc = out["survey_c4"]
        rec["survey_c4"] = _pick(sc, ("chains_per_gpu", "value", "unit", "ms_per_step", "scaling", "host_issue_us_per_step"))
        rec["survey_c4"]["roofline"] = _pick(sc.get("roofline", {}), ("bound", "frac", "kernel_avg_ms", "achieved", "unit"))
    if "headline_v2noise" in out:
        rec["headline_v2noise"] = out["headline_v2noise"]
    if "modes" in out:
        rec["modes"] = out["modes"]
    rec["full_record"] = out.get("full_record")
    rec = _num(rec)
    for k in ("value", "ms_per_step"):   # the two figures the driver checks against its own clock: at full precision
        if k in out:
            rec[k] = out[k]
    line = json.dumps(rec, separators=(",", ":"))
    if len(line.encode()) > CONTRACT_MAX_BYTES and "modes" in rec:
        # never lose the head of the line to its tail: drop the compact map first (it is in the full record), loudly
        print(f"bench.py: contract line {len(line.encode())} B > {CONTRACT_MAX_BYTES}: `modes` left to {FULL_RECORD}", file=sys.stderr)
        rec["modes"] = {"dropped": f"see {FULL_RECORD}"}
        line = json.dumps(rec, separators=(",", ":"))
    assert len(line.encode()) <= CONTRACT_MAX_BYTES, f"contract line is {len(line.encode())} bytes, the cap is {CONTRACT_MAX_BYTES}"
    assert "\n" not in line
    return rec, line


def write_full_record(out):
    """the whole record (every mode's roofline dict, the smoothing record, the box, notes): bench_full.json next to bench.py, a copy under
    gpurun_out/ when that exists (it travels back from a GPU box), and stderr"""
    full = json.dumps(out, indent=1)
    where = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, FULL_RECORD), "w") as f:
                    f.write(full + "\n")
                where.append(os.path.relpath(os.path.join(d, FULL_RECORD), ROOT))
            except OSError as e:
                print(f"bench.py: could not write {d}/{FULL_RECORD}: {e}", file=sys.stderr)
    print("bench.py: full record follows (also: " + ", ".join(where) + ")\n" + json.dumps(out), file=sys.stderr)
    return where[0] if where else None


def emit_json(out):
    out["full_record"] = FULL_RECORD
    write_full_record(out)
    _, line = contract_record(out)
    data = (line + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=0, help="chains (paths) per GPU; 0 = the mode's named size")
    ap.add_argument("--mode", choices=sorted(MODES) + sorted(m + "_fused" for m in MODES if MODES[m][1] <= 3) + sorted(m + "_v2noise" for m in MODES) + sorted(m + "_v3noise" for m in MODES) + sorted(m + sfx for m in MODES if not MODES[m][5] and MODES[m][1] <= 12 for sfx in ("_parts", "_1buf", "_fused_parts", "_fused_1buf") if not (sfx.startswith("_fused") and MODES[m][1] > 3)),
                    default="mcmc")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 child runs that measure the headline kernel's HBM bytes on this box")
    args = ap.parse_args()
    claim_stdout()
    # started by a launcher (torch.distributed.run sets WORLD_SIZE / RANK / LOCAL_RANK): one process per GPU.
    # started bare (`python bench.py --gpus N`): this process drives all N devices itself.
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world > 1 or (world == 1 and os.environ.get("BENCH_PER_RANK") == "1"):   # (BENCH_PER_RANK: the launcher's code path at one rank, for tests)
        main_per_rank(args, world)
    else:
        main_local(args)


if __name__ == "__main__":
    main()
