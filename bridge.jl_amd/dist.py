"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm,
"gloo" in the CPU tests).

The path shards trivially: chains / proposals are independent units (test/guip.jl:255-262; one chain
= partialbridge_fitzhugh.jl:143-176).  Rank r owns the contiguous global ids
[r*P/world, (r+1)*P/world); the Philox stream is keyed by the GLOBAL id, so results do not depend on
the number of GPUs.  Grid, model and guide coefficients are replicated (each rank computes the same
guide on its host, <= a few hundred KB).  There is NO data-path collective; the only communication
is ONE all-gather of the fixed-size statistics block (acceptance counts, log-weight moments) --
latency-bound (64 B per rank), independent of the xGMI per-link bandwidth.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

STATS_LEN = 8   # {n, iterations, sum acc, sum ll, sum ll^2, min ll, max ll, sum acc^2}


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """rendezvous from the torchrun environment; 127.0.0.1 unless MASTER_ADDR says otherwise"""
    rank, local, world = env_rank()
    if world == 1:
        return rank, local, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = device if device is not None else torch.device("cuda", local)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard(total, rank, world):
    """contiguous shard [lo, hi) of `total` independent units for `rank`"""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def comm_info(ctx, h):
    import ctypes as C
    n, r, v, rn, rr = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ctx.check(ctx.lib.bhip_comm_info(h, C.byref(n), C.byref(r)))
    ctx.check(ctx.lib.bhip_comm_query(h, C.byref(v), C.byref(rn), C.byref(rr)))
    return {"nranks": n.value, "rank": r.value, "rccl_version": v.value, "rccl_nranks": rn.value, "rccl_rank": rr.value}


class Comm:
    """The product's own communicator (bhip_comm, RCCL over xGMI, include/bridgehip.h): the all-gather of the statistics
    block runs inside libbridgehip.so on the context's stream.  One process per GPU: rank 0 draws the RCCL unique id and
    the launcher's rendezvous (here: whatever torch.distributed group is up -- gloo or nccl) hands it to the other ranks."""

    def __init__(self, ctx, nranks, rank, uid):
        import ctypes as C
        self.ctx, self.nranks, self.rank = ctx, int(nranks), int(rank)
        h = C.c_void_p()
        buf = (C.c_ubyte * len(uid)).from_buffer_copy(bytes(uid))
        ctx.check(ctx.lib.bhip_comm_init_rank(ctx.h, self.nranks, self.rank, C.cast(buf, C.c_void_p), C.byref(h)))
        self.h = h

    @staticmethod
    def unique_id(ctx):
        import ctypes as C
        buf = (C.c_ubyte * 128)()
        rc = ctx.lib.bhip_comm_unique_id(C.cast(buf, C.c_void_p), 128)
        if rc != 0:
            raise RuntimeError(f"bhip_comm_unique_id failed ({rc}): RCCL not available")
        return bytes(buf)

    @classmethod
    def from_torch_dist(cls, ctx):
        """every rank of the initialised torch.distributed group calls this"""
        if not dist.is_initialized():
            return cls(ctx, 1, 0, cls.unique_id(ctx))
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id(ctx) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(ctx, world, rank, box[0])

    def allgather(self, send, recv=None):
        """send: contiguous float64 device tensor [count]; returns [nranks, count] (one RCCL all-gather, in the library)"""
        send = send.contiguous()
        if recv is None:
            recv = torch.empty(self.nranks * send.numel(), dtype=torch.float64, device=send.device)
        import ctypes as C
        self.ctx.check(self.ctx.lib.bhip_comm_allgather(self.h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel()))
        return recv.reshape(self.nranks, -1)

    def info(self):
        """what the library (bhip_comm_info) and RCCL itself (bhip_comm_query: ncclGetVersion, ncclCommCount, ncclCommUserRank) say"""
        return comm_info(self.ctx, self.h)

    def destroy(self):
        if getattr(self, "h", None):
            self.ctx.lib.bhip_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class CommGroup:
    """ONE process driving several devices: bhip_comm_init_all (ncclCommInitAll) over one context per device and the grouped
    all-gather bhip_comm_allgather_group (every rank's ncclAllGather inside one ncclGroupStart/End, each on its context's
    stream).  This is the form a single-threaded `ccall` host -- the reference is single-threaded Julia -- uses for the 8 GPUs
    of a node; launches are asynchronous, so one host thread keeps all devices busy."""

    def __init__(self, ctxs):
        import ctypes as C
        self.ctxs = list(ctxs)
        n = self.nranks = len(self.ctxs)
        hs = (C.c_void_p * n)(*[c.h for c in self.ctxs])
        self._comms = (C.c_void_p * n)()
        self.ctxs[0].check(self.ctxs[0].lib.bhip_comm_init_all(n, hs, self._comms))

    def allgather(self, sends, recvs=None):
        """sends[k]: contiguous float64 tensor [count] on device k; returns the list of [nranks, count] tensors (one per device)"""
        import ctypes as C
        n, count = self.nranks, sends[0].numel()
        if recvs is None:
            recvs = [torch.empty(n * count, dtype=torch.float64, device=s.device) for s in sends]
        sp = (C.c_void_p * n)(*[s.data_ptr() for s in sends])
        rp = (C.c_void_p * n)(*[r.data_ptr() for r in recvs])
        self.ctxs[0].check(self.ctxs[0].lib.bhip_comm_allgather_group(n, self._comms, sp, rp, count))
        return [r.reshape(n, -1) for r in recvs]

    def info(self):
        """one bhip_comm_info / bhip_comm_query record per rank of the group"""
        return [comm_info(self.ctxs[k], self._comms[k]) for k in range(self.nranks)]

    def destroy(self):
        if getattr(self, "_comms", None) is not None:
            for k in range(self.nranks):
                if self._comms[k]:
                    self.ctxs[0].lib.bhip_comm_destroy(self._comms[k])
            self._comms = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class ChainsGroup:
    """The chain ensembles of ONE process driving several devices (one `Chains` per device): `step` is ONE C-ABI call that
    issues every device's launches round-robin (bhip_chains_step_group), `stats` one call for all the per-device reductions
    (bhip_chains_stats_group).  A single-threaded ccall host then pays one FFI crossing per call, not one per device and iteration."""

    def __init__(self, chains):
        import ctypes as C
        self.chains = list(chains)
        self.n = len(self.chains)
        self._hs = (C.c_void_p * self.n)(*[c.h.value if hasattr(c.h, "value") else c.h for c in self.chains])
        self.lib = self.chains[0].ctx.lib

    def step(self, rho, iters=1, skip=-1):
        import ctypes as C
        rc = self.lib.bhip_chains_step_group(self.n, self._hs, float(rho), int(iters), int(skip))
        # the library is the authority on how far every ensemble got: after a failing launch they stand at different counts
        for k, c in enumerate(self.chains):
            it = C.c_uint32()
            if self.lib.bhip_chains_iterations(self._hs[k], C.byref(it)) == 0:
                c.iterations = int(it.value)
        if rc:
            for c in self.chains:   # the failing ensemble's context holds the message
                msg = c.ctx.lib.bhip_last_error(c.ctx.h)
                if msg:
                    raise RuntimeError(f"bhip_chains_step_group: rc {rc}: {msg.decode()}")
            raise RuntimeError(f"bhip_chains_step_group: rc {rc}")

    def stats(self, outs):
        """outs[k]: float64 tensor [STATS_LEN] on the device of chains[k]"""
        import ctypes as C
        ps = (C.c_void_p * self.n)(*[o.data_ptr() for o in outs])
        self.chains[0].ctx.check(self.lib.bhip_chains_stats_group(self.n, self._hs, ps))
        return outs


def allgather_stats(stats, world=None, comm=None):
    """ONE all-gather of the per-rank statistics block -> tensor [world, STATS_LEN] on every rank.
    comm: a Comm (the product's RCCL communicator); without it the torch.distributed group is used (gloo in the CPU tests)"""
    if comm is not None:
        return comm.allgather(stats)
    world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
    if world == 1:
        return stats.reshape(1, -1)
    out = torch.empty(world * stats.numel(), dtype=stats.dtype, device=stats.device)
    dist.all_gather_into_tensor(out, stats.contiguous())
    return out.reshape(world, -1)


def combine_stats(gathered):
    """fold the gathered blocks into ensemble-level numbers (host)"""
    g = np.asarray(gathered.detach().cpu() if isinstance(gathered, torch.Tensor) else gathered, dtype=np.float64).reshape(-1, STATS_LEN)
    n = g[:, 0].sum()
    iters = g[:, 1].max()
    mean_ll = g[:, 3].sum() / n
    var_ll = max(g[:, 4].sum() / n - mean_ll ** 2, 0.0) * (n / max(n - 1, 1))
    acc_mean = g[:, 2].sum() / n
    return dict(chains=int(n), iterations=int(iters), acceptance_rate=float(acc_mean / max(iters, 1)),
                mean_ll=float(mean_ll), var_ll=float(var_ll), min_ll=float(g[:, 5].min()), max_ll=float(g[:, 6].max()),
                acc_per_chain_mean=float(acc_mean),
                acc_per_chain_var=float(max(g[:, 7].sum() / n - acc_mean ** 2, 0.0)))


def allgather_pathstats(n, mean, m2, world=None):
    """all-gather the per-rank pointwise Welford states (n, mean [N,d], m2 [N,d,d]) and merge them with
    the parallel form of mcnext (src/mclog.jl:31-38)."""
    from .api import mcmerge
    world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
    if world == 1:
        return n, mean, m2
    payload = torch.from_numpy(np.concatenate([[float(n)], np.ravel(mean), np.ravel(m2)]))
    out = torch.empty(world * payload.numel(), dtype=payload.dtype)
    if dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        o = out.to(dev)
        dist.all_gather_into_tensor(o, payload.to(dev))
        out = o.cpu()
    else:
        dist.all_gather_into_tensor(out, payload)
    out = out.numpy().reshape(world, -1)
    N, d = mean.shape
    state = None
    for r in range(world):
        nr, mr, qr = int(out[r, 0]), out[r, 1:1 + N * d].reshape(N, d), out[r, 1 + N * d:].reshape(N, d, d)
        state = (mr, qr, nr) if state is None else mcmerge(state, (mr, qr, nr))
    return state[2], state[0], state[1]
