"""Host-side mirror of Bridge.jl's interface for the guided-proposal hot path.

Names, argument order and error behaviour follow the reference (Julia `!` is spelled `_`):

    ContinuousTimeProcess, SamplePath          src/types.jl:23,71-76
    Wiener, sample, sample_                    src/wiener.jl:1-58
    Euler / EulerMaruyama, solve, solve_       src/euler.jl:14-16,117-152,246-268
    bridge_ (deprecated alias of solve_)       src/deprecated.jl:16-17
    GuidedBridge                               src/guip.jl:165-206
    PartialBridge                              src/partialbridge.jl:33-58
    PartialBridgeNuH                           src/partialbridgenuH.jl:122-169
    PartialBridgeInplace  (`PartialBridge!`)   src/partialbridgen!.jl:32-97
    LeftRule, llikelihood, lptilde             src/ode.jl:8, src/guip.jl:206,429-438
    mcmc  (the script loop)                    project_partialbridge/partialbridge_fitzhugh.jl:125-176
    mcstart / mcnext / mcstats / mcband        src/mclog.jl:22-93

What differs from the reference, by construction: a path object holds an ENSEMBLE of paths
(`EnsemblePath`, struct-of-arrays on the GPU) instead of one path, user processes are chosen from
a registry of device functors (arbitrary Julia/Python closures cannot run inside a HIP kernel),
and the noise comes from a counter-based Philox stream instead of a global RNG.

All numerical work happens in libbridgehip.so; nothing here falls back to the CPU.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import dp, vp

# ----------------------------------------------------------------------------- ids (include/bridgehip.h)
MODEL_WIENER, MODEL_OU, MODEL_LINPRO, MODEL_FHN, MODEL_NCLAR, MODEL_INTDIFF, MODEL_LORENZ, MODEL_FHN2, MODEL_PENDULUM = range(9)
AUX_AFFINE, AUX_LINPRO, AUX_FHN_STARTEND, AUX_CALLBACK = range(4)
GUIDE_NONE, GUIDE_HV, GUIDE_LMMU, GUIDE_NUH, GUIDE_NUH_INPLACE = range(5)
CHAINS_STORE_X = 1
STATS_LEN = 8


OPT_WAVE_SPECIALISED = 1   # BHIP_OPT_WAVE_SPECIALISED
OPT_TUNE_PLACEMENT = 2     # BHIP_OPT_TUNE_PLACEMENT
OPT_MID_VALU = 3           # BHIP_OPT_MID_VALU
OPT_FUSED_ARITHMETIC = 4   # BHIP_OPT_FUSED_ARITHMETIC
OPT_NOISE_SPEC = 5         # BHIP_OPT_NOISE_SPEC: 4 (bhip-philox-v4, default) | 3 (bhip-philox-v3) | 2 (bhip-philox-v2, full resolution)
AUX_LINEARAPPR = 4         # BHIP_AUX_LINEARAPPR


class BridgeError(RuntimeError):
    """error(...) of the reference / non-zero BHIP_E* code of the library"""


def _cm(A):
    """column-major flattening (Julia memory order) of a matrix or a stack of matrices"""
    A = np.asarray(A, dtype=np.float64)
    if A.ndim <= 1:
        return np.ascontiguousarray(A)
    if A.ndim == 2:
        return np.ascontiguousarray(A.T).ravel()
    return np.ascontiguousarray(np.swapaxes(A, -1, -2)).reshape(A.shape[0], -1)


def _uncm(a, r, c):
    a = np.asarray(a)
    if a.ndim == 1:
        return a.reshape(c, r).T.copy()
    return np.swapaxes(a.reshape(a.shape[0], c, r), -1, -2).copy()


def _dptr(a):
    return a.ctypes.data_as(dp)


# ----------------------------------------------------------------------------- context
class Context:
    """One device + stream (bhip_ctx).  Launches go to torch's current stream of that device."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = vp()
        if device == -1:   # host-only: guide coefficients can be computed, nothing can be launched
            self.device, self.stream = None, None
            rc = self.lib.bhip_ctx_create(-1, None, C.byref(h))
            if rc != 0:
                raise BridgeError(f"bhip_ctx_create failed ({rc})")
            self.h = h
            return
        if not torch.cuda.is_available():
            raise BridgeError("no HIP device visible: bridgehip has no CPU path")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self.stream = stream
        rc = self.lib.bhip_ctx_create(device, vp(stream), C.byref(h))
        if rc != 0:
            raise BridgeError(f"bhip_ctx_create failed ({rc})")
        self.h = h

    def check(self, rc):
        if rc != 0:
            msg = self.lib.bhip_last_error(self.h)
            raise BridgeError((msg.decode() if msg else "") + f" [bhip {rc}]")

    def sync(self):
        self.check(self.lib.bhip_ctx_sync(self.h))

    def set_option(self, option, value):
        """bhip_ctx_set_option: OPT_WAVE_SPECIALISED 1 (default) / 0"""
        self.check(self.lib.bhip_ctx_set_option(self.h, int(option), int(value)))

    def get_option(self, option):
        """bhip_ctx_get_option, e.g. get_option(OPT_NOISE_SPEC) -> 4 | 3 | 2 (which stream of normals this context draws)"""
        v = C.c_int()
        self.check(self.lib.bhip_ctx_get_option(self.h, int(option), C.byref(v)))
        return v.value

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float64, device=self.device)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.bhip_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


# ----------------------------------------------------------------------------- processes
class ContinuousTimeProcess:
    """src/types.jl:23.  Subclasses name a device functor (model id) and its parameters."""
    model_id = None
    d = None
    mp = None

    def params(self):
        raise NotImplementedError


class Wiener(ContinuousTimeProcess):
    """Wiener{SVector{d}}  src/wiener.jl:1-3"""
    model_id = MODEL_WIENER

    def __init__(self, d=1):
        self.d = self.mp = d

    def params(self):
        return np.zeros(0)


class OrnsteinUhlenbeck(ContinuousTimeProcess):
    """b = -beta*x, sigma constant (test/guip.jl:8-26, README.md:69-77)"""
    model_id, d, mp = MODEL_OU, 1, 1

    def __init__(self, beta, sigma):
        if not (math.isnan(beta) or beta > 0.0):
            raise BridgeError("Parameter λ must be positive.")       # test/guip.jl:13
        if not (math.isnan(sigma) or sigma > 0.0):
            raise BridgeError("Parameter σ must be positive.")       # test/guip.jl:14
        self.beta, self.sigma = float(beta), float(sigma)

    def params(self):
        return np.array([self.beta, self.sigma])


class LinPro(ContinuousTimeProcess):
    """dX = B(X - mu)dt + sigma dW   src/linpro.jl:65-87.  Usable as target and as auxiliary."""
    model_id = MODEL_LINPRO
    aux_kind = AUX_LINPRO

    def __init__(self, B, mu, sigma):
        self.B = np.atleast_2d(np.asarray(B, dtype=np.float64))
        self.d = self.mp = self.B.shape[0]
        self.mu = np.atleast_1d(np.asarray(mu, dtype=np.float64)).reshape(self.d)
        self.sigma = np.atleast_2d(np.asarray(sigma, dtype=np.float64)).reshape(self.d, self.d)

    def params(self):
        return np.concatenate([_cm(self.B), self.mu, _cm(self.sigma)])

    aux_params = params


class FitzhughDiffusion(ContinuousTimeProcess):
    """project_partialbridge/partialbridge_fitzhugh.jl:36-46"""
    model_id, d, mp = MODEL_FHN, 2, 1

    def __init__(self, eps, s, gamma, beta, sigma):
        self.eps, self.s, self.gamma, self.beta, self.sigma = map(float, (eps, s, gamma, beta, sigma))

    def params(self):
        return np.array([self.eps, self.s, self.gamma, self.beta, self.sigma])


class NclarDiffusion(ContinuousTimeProcess):
    """project_partialbridge/partialbridge_nclar.jl:52-61"""
    model_id, d, mp = MODEL_NCLAR, 3, 1

    def __init__(self, alpha, omega, sigma):
        self.alpha, self.omega, self.sigma = map(float, (alpha, omega, sigma))

    def params(self):
        return np.array([self.alpha, self.omega, self.sigma])


class IntegratedDiffusion(ContinuousTimeProcess):
    """test/partialbridge.jl:7-15"""
    model_id, d, mp = MODEL_INTDIFF, 2, 1

    def __init__(self, gamma):
        self.gamma = float(gamma)

    def params(self):
        return np.array([self.gamma])


class Lorenz(ContinuousTimeProcess):
    """src/Models.jl:41-58"""
    model_id, d, mp = MODEL_LORENZ, 3, 3

    def __init__(self, theta=(10.0, 28.0, 8 / 3), sigma=(1.0, 1.0, 1.0)):
        self.theta, self.sigma = tuple(map(float, theta)), tuple(map(float, sigma))

    def params(self):
        return np.array(self.theta + self.sigma)


class FitzHughNagumo(ContinuousTimeProcess):
    """Bridge.Models.FitzHughNagumo  src/Models.jl:9-20"""
    model_id, d, mp = MODEL_FHN2, 2, 2

    def __init__(self, eps, s, gamma, beta, sigma1, sigma2):
        self.p = tuple(map(float, (eps, s, gamma, beta, sigma1, sigma2)))

    def params(self):
        return np.array(self.p)


class Pendulum(ContinuousTimeProcess):
    """src/Models.jl:69-88"""
    model_id, d, mp = MODEL_PENDULUM, 2, 1

    def __init__(self, theta2, gamma):
        self.theta2, self.gamma = float(theta2), float(gamma)

    def params(self):
        return np.array([self.theta2, self.gamma])


class UserProcess(ContinuousTimeProcess):
    """A user-defined target process: the body of `Bridge.b(t, x, P)` as HIP C++ text (compiled for
    gfx950 with hipRTC on first use) + a constant diffusion matrix sigma [d, mp],

        P = UserProcess(2, "o[0] = (x[0]-x[1]-x[0]*x[0]*x[0]+par[1])/par[0]; o[1] = par[2]*x[0]-x[1]+par[3];",
                        par=[0.1, 0.0, 1.5, 0.8], sigma=[[0.0], [0.3]])

    or, with `sigma_src`, the body of a state-dependent `Bridge.sigma(t, x, P)` that fills `s` (d x mp,
    column-major, zero-initialised) -- then `constdiff(P)` is false:

        P = UserProcess(2, drift_src, par, sigma_src="s[0] = par[5]*sqrt(1.0 + x[0]*x[0]); s[3] = par[6];", mp=2)

    The texts see `double t`, `const double* x`, `const double* par`; the drift writes `double* o`."""

    def __init__(self, d, drift_src, par, sigma=None, ctx=None, sigma_src=None, mp=None):
        self.ctx = ctx or default_context()
        self.d = int(d)
        self.par = np.atleast_1d(np.asarray(par, dtype=np.float64)).ravel()
        self.drift_src, self.sigma_src = drift_src, sigma_src
        mid = C.c_int()
        if sigma_src is None:
            if sigma is None:
                raise BridgeError("UserProcess: give a constant sigma matrix or a sigma_src text")
            self.sigma = np.asarray(sigma, dtype=np.float64).reshape(self.d, -1)
            self.mp = self.sigma.shape[1]
            self.ctx.check(self.ctx.lib.bhip_model_define(self.ctx.h, self.d, self.mp, len(self.par), drift_src.encode(), C.byref(mid)))
        else:
            if sigma is not None:
                raise BridgeError("UserProcess: sigma and sigma_src are exclusive")
            self.sigma = None
            self.mp = int(mp if mp is not None else d)
            self.ctx.check(self.ctx.lib.bhip_model_define_sigma(self.ctx.h, self.d, self.mp, len(self.par), drift_src.encode(),
                                                                sigma_src.encode(), C.byref(mid)))
        self.model_id = mid.value

    def params(self):
        return self.par if self.sigma is None else np.concatenate([self.par, _cm(self.sigma)])


class UserProcessComponents(ContinuousTimeProcess):
    """A user-defined target at LARGE state dimension (4 <= d <= 32, dense constant sigma [d, d]): the body of
    `Bridge.b(t, x, P)` given COMPONENT-WISE as HIP C++ text -- it sees `int k` (component), `int d`, `double t`,
    `const double* x`, `const double* par` and writes `double o` -- compiled with hipRTC into the fp64-MFMA tile kernel (d >= 9) or the
    path-per-lane kernels (d = 4..8: `k` is then a constant per unrolled component):

        P = UserProcessComponents(16, "o = (x[(k+1)%d] - x[(k+d-2)%d])*x[(k+d-1)%d] - x[k] + par[0];", par=[8.0], sigma=0.5*np.eye(16))"""

    def __init__(self, d, component_src, par, sigma, ctx=None):
        self.ctx = ctx or default_context()
        self.d = self.mp = int(d)
        self.par = np.atleast_1d(np.asarray(par, dtype=np.float64)).ravel()
        self.sigma = np.asarray(sigma, dtype=np.float64).reshape(self.d, self.d)
        self.component_src = component_src
        mid = C.c_int()
        self.ctx.check(self.ctx.lib.bhip_model_define_components(self.ctx.h, self.d, len(self.par), component_src.encode(), C.byref(mid)))
        self.model_id = mid.value

    def params(self):
        return np.concatenate([self.par, _cm(self.sigma)])


# ---- auxiliary processes (Bridge.B / Bridge.beta / Bridge.sigma / Bridge.a 2-arg methods)
class AffineAux:
    """constant B, beta, sigma;  b~(t,x) = B*x + beta  (e.g. FitzhughDiffusionAux "linearised_end",
    partialbridge_fitzhugh.jl:99-101,112-116; NclarDiffusionAux; IntegratedDiffusionAux)"""
    aux_kind = AUX_AFFINE

    def __init__(self, B, beta, sigma):
        self.B = np.atleast_2d(np.asarray(B, dtype=np.float64))
        self.d = self.B.shape[0]
        self.beta = np.atleast_1d(np.asarray(beta, dtype=np.float64)).reshape(self.d)
        self.sigma = np.asarray(sigma, dtype=np.float64).reshape(self.d, -1)

    def aux_params(self):
        return np.concatenate([_cm(self.B), self.beta, _cm(self.sigma)])


class FitzhughDiffusionAuxStartEnd:
    """time-dependent "linearised_startend" auxiliary  partialbridge_fitzhugh.jl:58-73,102-105"""
    aux_kind = AUX_FHN_STARTEND
    d = 2

    def __init__(self, eps, s, gamma, beta, sigma, t, u, T, v):
        self.p = tuple(map(float, (eps, s, gamma, beta, sigma, t, u, T, v)))

    def aux_params(self):
        return np.array(self.p)


def fitzhugh_aux_linearised_end(P, v):
    """Bridge.B/beta of FitzhughDiffusionAux, aux_choice == "linearised_end" (partialbridge_fitzhugh.jl:99-100)"""
    B = [[1 / P.eps - 3 * (v * v) / P.eps, -1 / P.eps], [P.gamma, -1.0]]       # Julia lowers v^2, v^3 to products
    beta = [P.s / P.eps + 2 * (v * v * v) / P.eps, P.beta]
    return AffineAux(B, beta, [[0.0], [P.sigma]])


class LinearAppr:
    """LinearAppr (src/linpro.jl:181-204): the linearisation of a target along a path on the proposal's grid.
    LinearAppr(xx, B, b, Sigma) with arrays [N, d], [N, d, d], [N, d], [N, d, m'] -- or LinearAppr.along(Y) = linearappr(Y, P):
    B_i = bderiv(t_i, y_i, P), b_i = b(t_i, y_i, P), Sigma_i = sigma(t_i, y_i, P), filled by the library for the proposal's target."""
    aux_kind = 4

    def __init__(self, xx, B=None, b=None, Sigma=None):
        self.xx = np.ascontiguousarray(np.atleast_2d(xx), dtype=np.float64)
        self.B, self.b, self.Sigma = B, b, Sigma

    @classmethod
    def along(cls, Y):
        return cls(Y)

    def _fill(self, po):
        N, d, mp = len(po.tt), po.d, po.mp
        B, b, S = np.empty((N, d * d)), np.empty((N, d)), np.empty((N, d * mp))
        po.ctx.check(po.ctx.lib.bhip_linearappr(po.h, _dptr(self.xx), _dptr(B), _dptr(b), _dptr(S)))
        self.B = np.swapaxes(B.reshape(N, d, d), -1, -2).copy()
        self.b = b
        self.Sigma = np.swapaxes(S.reshape(N, mp, d), -1, -2).copy()


class LinearNoiseAppr:
    """LinearNoiseAppr(tt, P, x, a, direction)  src/guip.jl:114-146: B = 0, beta_i = the slope of the deterministic path Y
    (R3 of the target's drift, direction "forward" from x at tt[1], "backward" from x at tt[end], "nothing": Y = 0), a = the
    target's a.  LinearNoiseAppr.with_path(Y) takes Y as given (the adaptation's Pt.Y.yy[:] = xx, smoothing.jl:136-139).
    The grid and the target are the proposal's; `a` must be the target's a (checked by the library)."""
    aux_kind = 4
    noise = True

    def __init__(self, tt=None, P=None, x=None, a=None, direction="forward", Y=None):
        self.x = None if x is None else np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
        self.direction = {"forward": 1, "backward": -1, "nothing": 0, 1: 1, -1: -1, 0: 0}[direction]
        self.Y = None if Y is None else np.ascontiguousarray(Y, dtype=np.float64)

    @classmethod
    def with_path(cls, Y):
        return cls(Y=Y)

    def _install(self, po):
        if self.Y is None:
            self.Y = np.empty((len(po.tt), po.d))
            po.ctx.check(po.ctx.lib.bhip_linearnoiseappr_path(po.h, None if self.x is None else _dptr(self.x), self.direction, _dptr(self.Y)))
        po.ctx.check(po.ctx.lib.bhip_proposal_set_aux_linearnoiseappr(po.h, _dptr(self.Y)))


def linearappr(Y, P=None):
    """linearappr(Y, P)  src/linpro.jl:196 -- the target P is the proposal's (filled when the GuidedBridge is built)"""
    return LinearAppr.along(Y.yy if hasattr(Y, "yy") else Y)


class CallbackAux:
    """user-defined auxiliary: fn(t) -> (B, beta, a) evaluated on the host while the guide ODE is
    integrated (the Python twin of a Julia @cfunction)"""
    aux_kind = AUX_CALLBACK

    def __init__(self, d, fn, mu=None):
        self.d, self.fn, self.mu = d, fn, None if mu is None else np.asarray(mu, dtype=np.float64)

        def _cb(t, Bp, bp, ap, _user):
            B, beta, a = self.fn(t)
            Bc, ac = _cm(np.atleast_2d(B)), _cm(np.atleast_2d(a))
            for k in range(d * d):
                Bp[k] = Bc[k]
                ap[k] = ac[k]
            bb = np.atleast_1d(np.asarray(beta, dtype=np.float64))
            for k in range(d):
                bp[k] = bb[k]

        self._cfn = _lib.AUX_FN(_cb)


# ----------------------------------------------------------------------------- paths
class SamplePath:
    """host SamplePath{T}: tt [N], yy [N, dim]   src/types.jl:71-76"""

    def __init__(self, tt, yy):
        self.tt = np.array(tt, dtype=np.float64)
        self.yy = np.array(yy, dtype=np.float64).reshape(len(self.tt), -1)

    def __len__(self):
        return len(self.tt)

    def copy(self):
        return SamplePath(self.tt.copy(), self.yy.copy())


PARTS_MIN_BYTES = 1 << 30   # ensembles of this size and more are kept in two parts by default (EnsemblePath(parts=None))


class _DevView:
    """library-owned device memory as something torch.as_tensor accepts (__cuda_array_interface__, version 2)"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2, "strides": None}


class EnsemblePath:
    """An ensemble of sample paths on one GPU, struct-of-arrays fp64:
    data[i, k, p] = component k of path p at grid index i (contiguous in p).

    ONE container, kept in `nparts` device buffers (src/types.jl:71-81 has one SamplePath type; an ensemble is a loop there):
      nparts = 1   a torch tensor [N, dim, npaths] (`data`, `ptr()`), leading dimension npaths;
      nparts = 2/3 paths [j*part_paths, (j+1)*part_paths) live in buffer j, [N, dim, part_paths] each, the buffers pairwise in DIFFERENT
                   96-GiB pieces of the device memory (bhip_alloc_apart): an ensemble written by one kernel is ONE write stream, and a
                   write stream inside one piece moves 4.3-4.4 TB/s where two streams in two pieces move 5.8-6.0 (profiles/r5_three_pieces.txt).
    parts = None (default): 2 from PARTS_MIN_BYTES (1 GiB) on, else 1.  Every function of this module takes either form: it walks the
    column ranges that lie inside one buffer of every ensemble involved (`segments`) and hands the library (pointer, leading dimension,
    paths) per range -- the fused proposal kernel writes all parts in one launch (bhip_sample_solve_parts)."""

    def __init__(self, tt, dim, npaths, ctx=None, data=None, parts=None):
        self.ctx = ctx or default_context()
        self.tt = np.array(tt, dtype=np.float64)
        self.dim, self.npaths = int(dim), int(npaths)
        N = len(self.tt)
        if data is not None:
            parts = 1
        if parts is None:
            # (one path per lane writes all parts in one launch; above d = 12 the MFMA tile kernel writes ONE buffer and is not bound by its store stream)
            parts = 2 if (8 * N * self.dim * self.npaths >= PARTS_MIN_BYTES and self.dim <= 12 and getattr(self.ctx, "device", None) is not None
                          and getattr(self.ctx.device, "type", "cpu") != "cpu") else 1
        self.nparts = int(parts)
        if not 1 <= self.nparts <= 3:
            raise BridgeError("EnsemblePath: 1, 2 or 3 parts")
        self._ptrs, self.parts, self.apart = None, [self], 0
        if self.nparts == 1:
            self.part_paths = self.npaths
            if data is None:
                data = torch.zeros((N, self.dim, self.npaths), dtype=torch.float64, device=self.ctx.device)
            if tuple(data.shape) != (N, self.dim, self.npaths) or data.dtype != torch.float64 or not data.is_contiguous():
                raise BridgeError("EnsemblePath: data must be a contiguous float64 tensor [N, dim, npaths]")
            self._data = data
            return
        self._data = None
        self.part_paths = ((self.npaths + self.nparts - 1) // self.nparts + 63) // 64 * 64
        self._ptrs = (vp * self.nparts)()
        apart = C.c_int()
        self.ctx.check(self.ctx.lib.bhip_alloc_apart(self.ctx.h, self.nparts, N * self.dim * self.part_paths * 8, self._ptrs, C.byref(apart)))
        self.apart = apart.value
        self.parts = [_PartPath(self.tt, self.dim, max(0, min(self.part_paths, self.npaths - j * self.part_paths)), self.part_paths, self._ptrs[j], self.ctx)
                      for j in range(self.nparts)]

    def __len__(self):
        return len(self.tt)

    @property
    def ld(self):
        """leading dimension of a buffer (of THE buffer when there is one)"""
        return self.part_paths

    @property
    def data(self):
        """the ensemble as one tensor [N, dim, npaths]: the tensor itself (one buffer) or a gathered COPY (parts)"""
        if self.nparts == 1:
            return self._data
        return torch.cat([q.data[:, :, :q.npaths] for q in self.parts if q.npaths > 0], dim=2)

    def ptr(self):
        if self.nparts != 1:
            raise BridgeError("this ensemble is kept in parts: address it by colptr(p) / segments()")
        return vp(self._data.data_ptr())

    def colptr(self, p):
        """device address of column p (its buffer's leading dimension is `ld`)"""
        if self.nparts == 1:
            return vp(self._data.data_ptr() + 8 * int(p))
        j, q = divmod(int(p), self.part_paths)
        return vp(int(self._ptrs[j]) + 8 * q)

    def _parts_args(self):
        """(nparts, pointer array, leading dimension, paths per part) as the *_parts entry points take an ensemble"""
        if self.nparts == 1:
            return 1, (vp * 1)(self.ptr()), self.ld, self.ld
        return self.nparts, self._ptrs, self.part_paths, self.part_paths

    def segments(self, *others):
        """column ranges (start, n) that lie inside ONE buffer of this ensemble and of every ensemble in `others` (None entries skipped)"""
        cuts = {0, self.npaths}
        for E in (self,) + tuple(o for o in others if o is not None):
            if E.npaths != self.npaths:
                raise BridgeError("ensembles differ in the number of paths")
            cuts.update(range(E.part_paths, E.npaths, E.part_paths))
        cuts = sorted(cuts)
        return [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]

    def copy(self):
        if self.nparts == 1:
            return EnsemblePath(self.tt.copy(), self.dim, self.npaths, self.ctx, self._data.clone())
        E = EnsemblePath(self.tt.copy(), self.dim, self.npaths, self.ctx, parts=self.nparts)
        for a, b in zip(E.parts, self.parts):
            a.data.copy_(b.data)
        return E

    @classmethod
    def from_paths(cls, tt, yy, ctx=None, parts=None):
        """upload host paths yy [npaths, N, dim] (Vector{SVector} per path) -> SoA ensemble"""
        yy = np.ascontiguousarray(yy, dtype=np.float64)
        if yy.ndim == 2:
            yy = yy[:, :, None]
        npaths, N, dim = yy.shape
        E = cls(tt, dim, npaths, ctx, parts=parts)
        for a, n in E.segments():
            E.ctx.check(E.ctx.lib.bhip_upload_aos(E.ctx.h, E.colptr(a), N, dim, E.ld, 0, n, _dptr(yy[a:a + n])))
        return E

    def paths(self, p0=0, n=None):
        """download paths p0..p0+n as host array [n, N, dim]"""
        n = self.npaths - p0 if n is None else n
        out = np.empty((n, len(self.tt), self.dim))
        for a, m in self.segments():
            lo, hi = max(a, p0), min(a + m, p0 + n)
            if hi > lo:
                seg = np.empty((hi - lo, len(self.tt), self.dim))
                self.ctx.check(self.ctx.lib.bhip_download_aos(self.ctx.h, self.colptr(a), len(self.tt), self.dim, self.ld, lo - a, hi - lo, _dptr(seg)))
                out[lo - p0:hi - p0] = seg
        return out

    def path(self, p):
        return SamplePath(self.tt, self.paths(p, 1)[0])

    def endpoints(self):
        """yy[N] of every path: tensor [dim, npaths]"""
        if self.nparts == 1:
            return self._data[-1]
        return torch.cat([q.data[-1][:, :q.npaths] for q in self.parts if q.npaths > 0], dim=1)

    def free(self):
        """give the buffers of an ensemble in parts back now (otherwise: when the object goes)"""
        if getattr(self, "_ptrs", None) is not None:
            self.ctx.lib.bhip_free_apart(self.ctx.h, self.nparts, self._ptrs)
            self._ptrs, self.parts = None, []

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PartPath(EnsemblePath):
    """one buffer of an ensemble in parts: an ensemble [N, dim, ld] in a library allocation, seen as a tensor through
    __cuda_array_interface__"""

    def __init__(self, tt, dim, npaths, ld, ptr, ctx):
        self.ctx, self.tt, self.dim, self.npaths = ctx, tt, int(dim), int(npaths)
        self.nparts, self.part_paths, self._ptr, self._ptrs, self.parts, self.apart = 1, int(ld), ptr, None, [self], 0
        self._view = None

    @property
    def data(self):
        """[N, dim, ld] -- columns npaths..ld-1 are padding"""
        if self._view is None:
            with torch.cuda.device(self.ctx.device):
                self._view = torch.as_tensor(_DevView(self._ptr, (len(self.tt), self.dim, self.part_paths)), device=self.ctx.device)
        return self._view

    def ptr(self):
        return vp(self._ptr)

    def colptr(self, p):
        return vp(int(self._ptr) + 8 * int(p))

    def copy(self):
        raise BridgeError("a part of an ensemble is a view of the library's allocation")

    def free(self):
        pass


def EnsembleParts(tt, dim, npaths, nparts=3, ctx=None):
    """an EnsemblePath kept in `nparts` buffers (the name of round 5's second container; there is one container now)"""
    return EnsemblePath(tt, dim, npaths, ctx, parts=nparts)


class SDESolver:
    pass


class EulerMaruyama(SDESolver):
    pass


Euler = EulerMaruyama


class LeftRule:
    pass


# ----------------------------------------------------------------------------- proposals
class _Proposal(ContinuousTimeProcess):
    """common part of GuidedBridge / PartialBridge*: grid, target, auxiliary, device rows"""
    kind = GUIDE_NONE

    def __init__(self, tt, P, Pt, ctx=None):
        self.ctx = ctx or default_context()
        self.tt = np.array(tt, dtype=np.float64)
        self.Target, self.Pt = P, Pt
        self.d, self.mp = P.d, P.mp
        par = np.ascontiguousarray(P.params(), dtype=np.float64)
        h = vp()
        lib = self.ctx.lib
        self.ctx.check(lib.bhip_proposal_create(self.ctx.h, _dptr(self.tt), len(self.tt), P.model_id, P.d,
                                                _dptr(par), len(par), C.byref(h)))
        self.h = h
        if Pt is not None:
            if getattr(Pt, "noise", False):
                Pt._install(self)
            elif Pt.aux_kind == AUX_LINEARAPPR:
                if Pt.B is None:      # LinearAppr given as (path Y, target): linearappr(Y, P) on the host  src/linpro.jl:196
                    Pt._fill(self)
                cmN = lambda A: np.ascontiguousarray(np.swapaxes(np.asarray(A, dtype=np.float64), -1, -2))   # [N] column-major matrices
                self.ctx.check(lib.bhip_proposal_set_aux_linearappr(h, _dptr(np.ascontiguousarray(Pt.xx, dtype=np.float64)), _dptr(cmN(Pt.B)),
                                                                    _dptr(np.ascontiguousarray(Pt.b, dtype=np.float64)), _dptr(cmN(Pt.Sigma))))
            elif Pt.aux_kind == AUX_CALLBACK:
                mu = None if Pt.mu is None else _dptr(np.ascontiguousarray(Pt.mu))
                self.ctx.check(lib.bhip_proposal_set_aux_callback(h, C.cast(Pt._cfn, vp), None, 0 if Pt.mu is None else 1, mu))
            else:
                ap = np.ascontiguousarray(Pt.aux_params(), dtype=np.float64)
                self.ctx.check(lib.bhip_proposal_set_aux(h, Pt.aux_kind, _dptr(ap), len(ap)))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.ctx.lib.bhip_proposal_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _get(self, shapes):
        N = len(self.tt)
        bufs = [None if s is None else np.empty((N,) + s if s != "v" else (self.m,)) for s in shapes]
        ptrs = [None if b is None else _dptr(b) for b in bufs]
        self.ctx.check(self.ctx.lib.bhip_proposal_guide_get(self.h, *ptrs))
        return bufs


class PlainProcess(_Proposal):
    """wraps an unguided target so that solve(EulerMaruyama(), u, W, P) runs on the device"""

    def __init__(self, tt, P, ctx=None):
        super().__init__(tt, P, None, ctx)


class GuidedBridge(_Proposal):
    """GuidedBridge(tt, P, Pt, v, h=0)  src/guip.jl:165-180; fields Hd ("H♢") [N,d,d], V [N,d]"""
    kind = GUIDE_HV

    def __init__(self, tt, P, Pt, v, h=None, ctx=None):
        super().__init__(tt, P, Pt, ctx)
        v = np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64)
        hp = None if h is None else _dptr(_cm(np.atleast_2d(h)))
        self.ctx.check(self.ctx.lib.bhip_proposal_guide_hv(self.h, _dptr(v), hp))
        self.m = self.d
        Hd, V, _, _ = self._get([(self.d * self.d,), (self.d,), None, None])
        self.Hd, self.V = _uncm(Hd, self.d, self.d), V


class PartialBridge(_Proposal):
    """PartialBridge(tt, P, Pt, L, v, Sigma)  src/partialbridge.jl:33-51; fields L, M, mu, v"""
    kind = GUIDE_LMMU

    def __init__(self, tt, P, Pt, L, v, Sigma=None, ctx=None):
        super().__init__(tt, P, Pt, ctx)
        L = np.atleast_2d(np.asarray(L, dtype=np.float64))
        self.m = L.shape[0]
        v = np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64)
        Sp = None if Sigma is None else _dptr(_cm(np.atleast_2d(Sigma)))
        self.ctx.check(self.ctx.lib.bhip_proposal_guide_lmmu(self.h, self.m, _dptr(_cm(L)), _dptr(v), Sp))
        m, d = self.m, self.d
        Lt, Mt, mut, vv = self._get([(m * d,), (m * m,), (m,), "v"])
        self.L, self.M, self.mu, self.v = _uncm(Lt, m, d), _uncm(Mt, m, m), mut, vv


class PartialBridgeNuH(_Proposal):
    """PartialBridgeνH(tt, P, Pt, L, v, eps, Sigma)  src/partialbridgenuH.jl:122-146; fields nu, H, C"""
    kind = GUIDE_NUH
    _inplace = 0

    def __init__(self, tt, P, Pt, L, v, eps, Sigma=None, ctx=None):
        super().__init__(tt, P, Pt, ctx)
        L = np.atleast_2d(np.asarray(L, dtype=np.float64))
        self.m = L.shape[0]
        v = np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64)
        Sp = None if Sigma is None else _dptr(_cm(np.atleast_2d(Sigma)))
        self.ctx.check(self.ctx.lib.bhip_proposal_guide_nuh(self.h, self.m, _dptr(_cm(L)), _dptr(v), float(eps), Sp, self._inplace))
        d = self.d
        nut, Ht, Cc, _ = self._get([(d,), (d * d,), (1,), None])
        self.nu, self.H, self.C = nut, _uncm(Ht, d, d), float(Cc[0, 0])


class PartialBridgeInplace(PartialBridgeNuH):
    """`PartialBridge!`(tt, P, Pt, L, v, eps, Sigmanoise)  src/partialbridgen!.jl:32-56"""
    kind = GUIDE_NUH_INPLACE
    _inplace = 1


class ProposalFromArrays(_Proposal):
    """a proposal whose guide arrays were computed elsewhere (bhip_proposal_guide_arrays)"""

    def __init__(self, tt, P, Pt, kind, m, A1, A2, A3=None, A4=None, ctx=None):
        super().__init__(tt, P, Pt, ctx)
        self.kind, self.m = kind, m
        arrs = [None if A is None else np.ascontiguousarray(A, dtype=np.float64) for A in (A1, A2, A3, A4)]
        self.ctx.check(self.ctx.lib.bhip_proposal_guide_arrays(self.h, kind, m, *[None if A is None else _dptr(A) for A in arrs]))


def lptilde(Po, u):
    """lptilde(P::GuidedBridge, u)  src/guip.jl:206 ;  NuH: -0.5 (nu1-u)'H1(nu1-u) - C"""
    out = C.c_double()
    u = np.ascontiguousarray(np.atleast_1d(u), dtype=np.float64)
    Po.ctx.check(Po.ctx.lib.bhip_proposal_lptilde(Po.h, _dptr(u), C.byref(out)))
    return out.value


# ----------------------------------------------------------------------------- sample / solve / llikelihood
def sample(tt, P, npaths=1, seed=0, iter=0, path0=0, ctx=None):
    """sample(tt, Wiener{...}()) -> W  (src/wiener.jl:11-15), for an ensemble of `npaths` paths.
    Path p draws from the Philox stream (seed, path0+p, iter)."""
    if not isinstance(P, Wiener):
        raise BridgeError("sample: only Wiener processes are sampled exactly on the device")
    W = EnsemblePath(tt, P.mp, npaths, ctx)
    return sample_(W, P, seed=seed, iter=iter, path0=path0)


def sample_(W, P, seed=0, iter=0, path0=0):
    """sample!(W, Wiener())  src/wiener.jl:24-58"""
    if not isinstance(P, Wiener) or P.mp != W.dim:
        raise BridgeError("sample!: dimension of W and of the Wiener process differ")
    ctx = W.ctx
    if W.nparts > 1 and W.dim <= 12:   # ONE launch over all buffers
        nw, wp, ldw, wpart = W._parts_args()
        ctx.check(ctx.lib.bhip_wiener_sample_parts(ctx.h, _dptr(W.tt), len(W.tt), W.dim, nw, wp, ldw, wpart, W.npaths, seed, iter, path0))
        return W
    for a, n in W.segments():
        ctx.check(ctx.lib.bhip_wiener_sample(ctx.h, _dptr(W.tt), len(W.tt), W.dim, W.colptr(a), W.ld, n, seed, iter, path0 + a))
    return W


def _x0(u, d):
    u = np.ascontiguousarray(np.atleast_1d(u), dtype=np.float64)
    if u.shape != (d,):
        raise BridgeError("Starting point has wrong length.")          # src/sde!.jl:31
    return u


def solve_(method, Y, u, W, P, ll=None, skip=0):
    """solve!(::EulerMaruyama, Y, u, W, P)  src/euler.jl:135-152 and, for guided proposals,
    solve!(::Euler, Y, u, W, P::Union{GuidedBridge,PartialBridge,PartialBridgeνH})  :247-268.
    Overwrites Y.tt with the proposal's grid (:256) and returns the endpoints yy[N] (:267) as a
    tensor [d, npaths].  If `ll` (tensor [npaths]) is given, llikelihood(LeftRule(), Y, P; skip)
    is accumulated in the same kernel."""
    if not isinstance(method, EulerMaruyama):
        raise BridgeError("solve!: only the Euler-Maruyama scheme runs on the device")
    if not isinstance(P, _Proposal):
        raise BridgeError("solve!: wrap the target with PlainProcess(tt, P) or use a guided proposal")
    if len(W) != len(Y):
        raise BridgeError("Y and W differ in length.")                  # src/euler.jl:137,251
    if len(P.tt) != len(W):
        raise BridgeError("Time axis mismatch between bridge P and driving W.")   # :248 (intent)
    if W.dim != P.mp or Y.dim != P.d or W.npaths != Y.npaths:
        raise BridgeError("solve!: dimension mismatch between Y, W and P")
    ctx = Y.ctx
    Y.tt[:] = P.tt                                                       # :256 / :142
    per_path = isinstance(u, torch.Tensor)
    x0 = None if per_path else _dptr(_x0(u, P.d))
    x0d = vp(u.data_ptr()) if per_path else None
    if per_path and (tuple(u.shape) != (P.d, Y.npaths) or not u.is_contiguous()):
        raise BridgeError("per-path starting points must be a contiguous tensor [d, npaths]")
    if (Y.nparts > 1 or W.nparts > 1) and not per_path:   # ensembles in parts: ONE launch reads and writes all buffers
        nw, wp, ldw, wpart = W._parts_args()
        nx, xp, ldx, xpart = Y._parts_args()
        rc = ctx.lib.bhip_solve_parts(ctx.h, P.h, x0, nw, wp, ldw, wpart, nx, xp, ldx, xpart, None if ll is None else vp(ll.data_ptr()), skip, Y.npaths)
        if rc != -3:   # (BHIP_EUNSUPPORTED: the tile kernel takes one buffer per launch -- range by range below)
            ctx.check(rc)
            return Y.endpoints()
    for a, n in Y.segments(W):
        llp = None if ll is None else vp(ll.data_ptr() + 8 * a)
        xs = None
        if per_path:   # x0_dev is [d][ldX]: this range's starts in a block of the buffer's leading dimension
            xs = u[:, a:a + n] if (Y.nparts == 1 and a == 0) else torch.zeros((P.d, Y.ld), dtype=torch.float64, device=u.device)
            if xs.shape[1] == Y.ld and xs is not u:
                xs[:, :n] = u[:, a:a + n]
            x0d = vp(xs.data_ptr())
        ctx.check(ctx.lib.bhip_solve(ctx.h, P.h, x0, x0d, W.colptr(a), W.ld, Y.colptr(a), Y.ld, llp, skip, n))
    return Y.endpoints()


def solve(method, u, W, P, ll=None, skip=0):
    """solve(::SDESolver, u, W, P) -> X   src/euler.jl:117-118,246"""
    X = EnsemblePath(W.tt, P.d, W.npaths, W.ctx, parts=W.nparts if W.nparts > 1 else None)
    solve_(method, X, u, W, P, ll=ll, skip=skip)
    return X


def bridge_(Y, u, W, P):
    """bridge!(Y, u, W, P): deprecated alias of solve!(Euler(), Y, u, W, P)  src/deprecated.jl:16-17;
    the 4-argument form is the one the scripts call (project/partialbridge.jl:63)"""
    return solve_(Euler(), Y, u, W, P)


def llikelihood(rule, X, Po, skip=0):
    """llikelihood(::LeftRule, X, Po; skip=0) for every path of the ensemble -> tensor [npaths]
    src/guip.jl:429-438, src/partialbridge.jl:67-77, src/partialbridgenuH.jl:171-181,
    src/partialbridgen!.jl:81-97  (constant-diffusivity branch)"""
    if not isinstance(rule, LeftRule):
        raise BridgeError("llikelihood: only LeftRule is implemented on the device")
    ctx = X.ctx
    out = ctx.empty(X.npaths)
    if X.nparts > 1:   # ONE launch over all buffers
        nx, xp, ldx, xpart = X._parts_args()
        rc = ctx.lib.bhip_llikelihood_parts(ctx.h, Po.h, nx, xp, ldx, xpart, vp(out.data_ptr()), skip, X.npaths)
        if rc != -3:   # (BHIP_EUNSUPPORTED: the tile kernel -- range by range below)
            ctx.check(rc)
            return out
    for a, n in X.segments():
        ctx.check(ctx.lib.bhip_llikelihood(ctx.h, Po.h, X.colptr(a), X.ld, vp(out.data_ptr() + 8 * a), skip, n))
    return out


def innovations_(method, W, Y, P):
    """innovations!(::EulerMaruyama, W, Y, P)  src/euler.jl:358-376: recover the driving W from the paths Y
    (inverse of solve!); P is a PlainProcess or a guided proposal with a square, invertible sigma."""
    if not isinstance(method, EulerMaruyama):
        raise BridgeError("innovations!: only the Euler-Maruyama scheme runs on the device")
    if len(W) != len(Y):
        raise BridgeError("Y and W differ in length.")                  # src/euler.jl:361
    if P.d != P.mp or W.dim != P.mp or Y.dim != P.d or W.npaths != Y.npaths:
        raise BridgeError("innovations!: needs square sigma and matching ensembles")
    W.tt[:] = Y.tt                                                       # :366
    ctx = Y.ctx
    for a, n in Y.segments(W):
        ctx.check(ctx.lib.bhip_innovations(ctx.h, P.h, Y.colptr(a), Y.ld, W.colptr(a), W.ld, n))
    return W


def innovations(method, Y, P):
    """innovations(method, Y, P) = innovations!(method, copy(Y), Y, P)   src/euler.jl:357"""
    return innovations_(method, EnsemblePath(Y.tt, P.mp, Y.npaths, Y.ctx, parts=Y.nparts if Y.nparts > 1 else None), Y, P)


def girsanov(X, P, Pt):
    """girsanov(X::SamplePath, P, Pt)  src/diffusion.jl:109-123: the discretised log-likelihood ratio
    dP/dPt of every path of the ensemble -> tensor [npaths].  P and Pt are two parameter sets of the same
    process type (example/fitzhugh_nagumo_full.jl:313-321), or Pt = Wiener (test/guip.jl:72).
    P may also be a PlainProcess / guided proposal already living on X's grid (its target is used)."""
    ctx = X.ctx
    if isinstance(P, _Proposal):
        Po = P
        if len(Po.tt) != len(X) or not np.array_equal(Po.tt, X.tt):
            raise BridgeError("Time axis mismatch between X and P")
        P = Po.Target
    else:
        Po = PlainProcess(X.tt, P, ctx)
    if isinstance(Pt, _Proposal):
        Pt = Pt.Target
    if X.dim != P.d:
        raise BridgeError("girsanov: dimension of X does not match P")
    if isinstance(Pt, Wiener):
        if Pt.d != P.d:
            raise BridgeError("girsanov: dimension of Pt does not match P")
        par_t, npar_t = None, 0
    elif type(Pt) is type(P) and Pt.d == P.d:
        pt = np.ascontiguousarray(Pt.params(), dtype=np.float64)
        par_t, npar_t = _dptr(pt), len(pt)
    else:
        raise BridgeError("girsanov: Pt must be of the same process type as P, or Wiener")
    out = ctx.empty(X.npaths)
    if X.nparts > 1:   # ONE launch over all buffers
        nx, xp, ldx, xpart = X._parts_args()
        ctx.check(ctx.lib.bhip_girsanov_parts(ctx.h, Po.h, par_t, npar_t, nx, xp, ldx, xpart, vp(out.data_ptr()), X.npaths))
        return out
    for a, n in X.segments():
        ctx.check(ctx.lib.bhip_girsanov(ctx.h, Po.h, par_t, npar_t, X.colptr(a), X.ld, vp(out.data_ptr() + 8 * a), n))
    return out


def gpupdate(H, V, L=None, Sigma=None, v=None):
    """gpupdate(Hd, V, L, Sigma, v) or gpupdate(P::GuidedBridge, L, Sigma, v)  src/guip.jl:221-243:
    returns the updated (Hd, V) after observing v = L x + N(0, Sigma) at the left end of the segment."""
    if isinstance(H, GuidedBridge):
        P, L, Sigma, v = H, V, L, Sigma
        H, V = P.Hd[0], P.V[0]
    H = np.atleast_2d(np.asarray(H, dtype=np.float64))
    d = H.shape[0]
    L = np.asarray(L, dtype=np.float64).reshape(-1, d)
    m = L.shape[0]
    Sigma = np.asarray(Sigma, dtype=np.float64).reshape(m, m)
    Vv = np.ascontiguousarray(np.atleast_1d(V), dtype=np.float64)
    vv = np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64)
    Ho, Vo = np.empty(d * d), np.empty(d)
    rc = _lib.load().bhip_gpupdate(d, m, _dptr(_cm(H)), _dptr(Vv), _dptr(_cm(L)), _dptr(_cm(Sigma)), _dptr(vv), _dptr(Ho), _dptr(Vo))
    if rc != 0:
        raise BridgeError(f"bhip_gpupdate failed ({rc})")
    return _uncm(Ho, d, d), Vo


def write_iterates_csv(fn, XX, subsamples, tt, every=50):
    """the script's CSV of MCMC iterates (partialbridge_fitzhugh.jl:180-189) for ONE chain:
    header `iteration, time, component, value`, rows for d in 1:dim, j in 1:every:N, (i,s) in subsamples
    (component fastest, then time, then iteration; 1-based component index)."""
    with open(fn, "w") as f:
        f.write("iteration, time, component, value \n")
        for X, s in zip(XX, subsamples):
            X = np.asarray(X)
            for j in range(0, X.shape[0], every):
                for dcomp in range(X.shape[1]):
                    f.write(f"{s},{float(tt[j])!r},{dcomp + 1},{float(X[j, dcomp])!r}\n")


def write_info(fn, aux_choice, endpoint, iterations, skip_it, x0, T, v, Sigma, L, dt, rho, acc):
    """the script's info-*.txt (partialbridge_fitzhugh.jl:191-208)"""
    ave_acc_perc = 100 * round(acc / iterations, 2)
    with open(fn, "w") as f:
        f.write(f"Choice of auxiliary process: {aux_choice}\n")
        f.write(f"Choice of endpoint: {endpoint}\n\n")
        f.write(f"Number of iterations: {iterations}\n")
        f.write(f"Skip every {skip_it} iterations, when saving to csv\n\n")
        f.write(f"Starting point: {list(np.atleast_1d(x0))}\n")
        f.write(f"End time T: {T}\n")
        f.write(f"Endpoint v: {v}\n")
        f.write(f"Noise Sigma: {np.asarray(Sigma).tolist()}\n")
        f.write(f"L: {np.asarray(L).tolist()}\n\n")
        f.write(f"Mesh width: {dt}\n")
        f.write(f"rho (Crank-Nicholsen parameter: {rho}\n")
        f.write(f"Average acceptance percentage: {ave_acc_perc}\n")
    return ave_acc_perc


def sample_solve(u, Po, npaths, seed=0, iter=0, path0=0, store_W=False, store_X=True, skip=0, ctx=None, parts=None):
    """fused  W = sample(tt, Wiener()); X = solve(Euler(), u, W, Po); ll = llikelihood(LeftRule(), X, Po)
    with in-kernel Philox noise.  Returns (X or None, W or None, ll or None).  A large X (1 GiB and more; `parts`) is kept in two
    buffers lying in different pieces of the device memory and written by ONE launch (bhip_sample_solve_parts)."""
    ctx = ctx or Po.ctx
    X = EnsemblePath(Po.tt, Po.d, npaths, ctx, parts=parts) if store_X else None
    W = EnsemblePath(Po.tt, Po.mp, npaths, ctx, parts=(X.nparts if X is not None and X.nparts > 1 else parts)) if store_W else None
    ll = ctx.empty(npaths) if Po.kind != GUIDE_NONE else None
    sample_solve_(X, W, ll, u, Po, npaths, seed=seed, iter=iter, path0=path0, skip=skip)
    return X, W, ll


def sample_solve_(X, W, ll, u, Po, npaths=None, seed=0, iter=0, path0=0, skip=0):
    """the fused proposal into existing containers (any of X, W, ll may be None): what sample_solve runs, and what a loop over
    iterations calls (bench.py's `proposals` mode)"""
    E = X if X is not None else W
    ctx = Po.ctx if E is None else E.ctx
    npaths = E.npaths if npaths is None else npaths
    per_path = isinstance(u, torch.Tensor)
    if per_path and (tuple(u.shape) != (Po.d, npaths) or not u.is_contiguous() or u.dtype != torch.float64):
        raise BridgeError("per-path starting points must be a contiguous float64 tensor [d, npaths]")
    x0 = None if per_path else _dptr(_x0(u, Po.d))
    llp = lambda a: None if ll is None else vp(ll.data_ptr() + 8 * a)
    if X is not None and X.nparts > 1 and W is None and not per_path:   # X in parts alone: all parts by one launch
        rc = ctx.lib.bhip_sample_solve_parts(ctx.h, Po.h, x0, X.nparts, X._ptrs, X.part_paths, X.part_paths, llp(0), skip, npaths, seed, iter, path0)
        if rc != -3:   # (BHIP_EUNSUPPORTED: the tile kernel writes one buffer per launch -- range by range below)
            ctx.check(rc)
            return X, W, ll
    segs = E.segments(X, W) if E is not None else [(0, npaths)]
    for a, n in segs:
        x0d = None
        if per_path:   # x0_dev is [d][ldX]
            ldx = X.ld if X is not None else npaths
            xs = u if (len(segs) == 1 and ldx == u.shape[1]) else torch.zeros((Po.d, ldx), dtype=torch.float64, device=u.device)
            if xs is not u:
                xs[:, :n] = u[:, a:a + n]
            x0d = vp(xs.data_ptr())
        ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, x0, x0d,
                                            None if W is None else W.colptr(a), n if W is None else W.ld,
                                            None if X is None else X.colptr(a), X.ld if X is not None else (ldx if per_path else n),
                                            llp(a), skip, n, seed, iter, path0 + a))
    return X, W, ll


def sample_solve_parts(u, Po, npaths, nparts=3, seed=0, iter=0, path0=0, skip=0, ctx=None, X=None):
    """sample_solve with X kept in `nparts` buffers (bhip_sample_solve_parts): the same values, column p of the ensemble = column
    p - j*part_paths of part j.  Returns (X, ll).  (= sample_solve(..., parts=nparts) without W.)"""
    ctx = ctx or Po.ctx
    X = X if X is not None else EnsemblePath(Po.tt, Po.d, npaths, ctx, parts=nparts)
    ll = ctx.empty(npaths) if Po.kind != GUIDE_NONE else None
    ctx.check(ctx.lib.bhip_sample_solve_parts(ctx.h, Po.h, _dptr(_x0(u, Po.d)), X.nparts, X._ptrs if X.nparts > 1 else (vp * 1)(X.ptr()), X.part_paths, X.part_paths,
                                              None if ll is None else vp(ll.data_ptr()), skip, npaths, seed, iter, path0))
    return X, ll


# ----------------------------------------------------------------------------- MCMC
class Chains:
    """An ensemble of independent pCN Metropolis-Hastings chains, one per GPU lane
    (project_partialbridge/partialbridge_fitzhugh.jl:125-176; test/partialbridgenuH.jl:155-198)."""

    def __init__(self, Po, x0, nchains, seed=0, path0=0, store_X=True, skip=0):
        self.Po, self.ctx = Po, Po.ctx
        h = vp()
        self.ctx.check(self.ctx.lib.bhip_chains_create(self.ctx.h, Po.h, nchains, path0, seed,
                                                       CHAINS_STORE_X if store_X else 0, C.byref(h)))
        self.h, self.n, self.skip = h, nchains, skip
        self.iterations = 0
        self.ctx.check(self.ctx.lib.bhip_chains_init(h, _dptr(_x0(x0, Po.d)), skip))

    def step(self, rho, iters=1, skip=None):
        """`iters` pCN iterations.  skip = None: the skip the ensemble was initialised with, so that llo and ll always
        sum the same terms (partialbridge_nclar.jl:121 passes it to both)."""
        skip = self.skip if skip is None else int(skip)
        self.ctx.check(self.ctx.lib.bhip_chains_step(self.h, float(rho), int(iters), skip))
        self.iterations += iters

    def placement(self):
        """what the placement did (bhip_chains_placement_info / _pieces): allocations of Xo tested against W (0 = not placed, 1 = the
        first pair already lay in different pieces), GB/s of two write streams into one piece (the context's reference) and into the
        kept (W, Xo) pair, the pieces of the context's map W and Xo lie in"""
        n, a, b, pw, px = C.c_int(), C.c_float(), C.c_float(), C.c_int(), C.c_int()
        self.ctx.check(self.ctx.lib.bhip_chains_placement_info(self.h, C.byref(n), C.byref(a), C.byref(b)))
        self.ctx.check(self.ctx.lib.bhip_chains_placement_pieces(self.h, C.byref(pw), C.byref(px)))
        return {"tries": n.value, "gbs_same_piece": a.value, "gbs_kept": b.value, "piece_w": pw.value, "piece_xo": px.value}

    def stats(self, out=None):
        """device tensor [8]: {nchains, iterations, sum acc, sum ll, sum ll^2, min ll, max ll, sum acc^2}"""
        out = self.ctx.empty(STATS_LEN) if out is None else out
        self.ctx.check(self.ctx.lib.bhip_chains_stats(self.h, vp(out.data_ptr())))
        return out

    def ll(self):
        out = np.empty(self.n)
        self.ctx.check(self.ctx.lib.bhip_chains_get(self.h, _dptr(out), None))
        return out

    def acc(self):
        out = np.empty(self.n, dtype=np.int64)
        self.ctx.check(self.ctx.lib.bhip_chains_get(self.h, None, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def paths(self, p0=0, n=None, want_W=True):
        n = self.n - p0 if n is None else n
        N = len(self.Po.tt)
        X = np.empty((n, N, self.Po.d))
        W = np.empty((n, N, self.Po.mp)) if want_W else None
        self.ctx.check(self.ctx.lib.bhip_chains_get_paths(self.h, p0, n, _dptr(X), None if W is None else _dptr(W)))
        return X, W

    def current_X(self):
        """EnsemblePath of the chains' CURRENT paths X (re-materialised from the current W)"""
        X = EnsemblePath(self.Po.tt, self.Po.d, self.n, self.ctx, parts=1)   # (the library writes all chains into one array)
        self.ctx.check(self.ctx.lib.bhip_chains_current_X(self.h, X.ptr(), X.ld))
        return X

    def proposal_X(self):
        """view [N, d, ld] of the proposal paths Xo written by the last iteration (store_X=True)"""
        ptr, ld = vp(), C.c_long()
        self.ctx.check(self.ctx.lib.bhip_chains_proposal_X(self.h, C.byref(ptr), C.byref(ld)))
        N, d = len(self.Po.tt), self.Po.d
        nbytes = N * d * ld.value * 8

        class _Raw:
            __cuda_array_interface__ = {"shape": (N, d, ld.value), "typestr": "<f8", "data": (ptr.value, False),
                                        "version": 2, "strides": None}
        t = torch.as_tensor(_Raw(), device=self.ctx.device)
        assert t.numel() * 8 == nbytes
        return t[:, :, :self.n]

    def pathstats(self):
        """pointwise ensemble (n, mean [N,d], m2 [N,d,d]) of the current X -- mcnext! state"""
        N, d = len(self.Po.tt), self.Po.d
        mean, m2 = np.empty((N, d)), np.empty((N, d * d))
        self.ctx.check(self.ctx.lib.bhip_chains_pathstats(self.h, _dptr(mean), _dptr(m2)))
        return self.n, mean, _uncm(m2, d, d)

    def save(self):
        """checkpoint: the chain state (current W, ll, acceptance counts, iteration counter) as a numpy byte array"""
        nb = C.c_size_t()
        self.ctx.check(self.ctx.lib.bhip_chains_state_bytes(self.h, C.byref(nb)))
        buf = np.empty(nb.value, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.bhip_chains_save(self.h, buf.ctypes.data_as(vp)))
        return buf

    def load(self, buf):
        """resume from Chains.save() of an ensemble with the same proposal shape, chains, seed and path0: the following
        iterations are bit-identical to those the saved ensemble would have run (counter-based noise)"""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.bhip_chains_load(self.h, buf.ctypes.data_as(vp)))
        self.iterations = int(self.stats().cpu()[1])
        return self

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.ctx.lib.bhip_chains_destroy(self.h)
                self.h = None
        except Exception:
            pass


SEG_NEWBLOCK, SEG_DOACCEPT = 1, 2


def cholupper_t(H):
    """cholupper(Hermitian(H))': the lower Cholesky factor read from H's UPPER triangle (src/gaussian.jl:54 needs it for rand(pi0))"""
    H = np.atleast_2d(np.asarray(H, dtype=np.float64))
    return np.linalg.cholesky(np.triu(H) + np.triu(H, 1).T)


class SegChains:
    """An ensemble of chains over m chained guided segments with ONE Metropolis-Hastings decision per chain and iteration,
    a pCN move of the starting point and (optionally) mcnext! per iteration -- the loop of
    supplements/smoothing/smoothing.jl:99-213 (bhip_segchains_*).  pos: the m proposals; pi0 = N(mu, C C')."""

    def __init__(self, pos, mu, chol, nchains, seed=0, path0=0, skip=0, mcnext=False, pooled=False, mcnext_mean_only=False, stats_every_iteration=False):
        self.pos, self.ctx = list(pos), pos[0].ctx
        self.m, self.n, self.d, self.mp, self.N = len(self.pos), int(nchains), pos[0].d, pos[0].mp, len(pos[0].tt)
        self.mean_only = bool(mcnext_mean_only) and not mcnext
        hs = (vp * self.m)(*[P.h for P in self.pos])
        h = vp()
        self.ctx.check(self.ctx.lib.bhip_segchains_create(self.ctx.h, self.m, hs, self.n, path0, seed, (1 if mcnext else 0) | (2 if pooled else 0) | (4 if mcnext_mean_only else 0) | (8 if stats_every_iteration else 0), C.byref(h)))
        self.h = h
        mu = np.ascontiguousarray(np.atleast_1d(mu), dtype=np.float64)
        self.ctx.check(self.ctx.lib.bhip_segchains_init(h, _dptr(mu), _dptr(_cm(np.atleast_2d(chol))), skip))
        self.iterations = 0

    def step(self, w_old, w_new, iters=None):
        """iterations with the pCN weights Wo = w_new*W2 + w_old*W (scalars: the same for `iters` iterations, or arrays)"""
        if np.isscalar(w_old):
            w_old, w_new = np.full(iters or 1, float(w_old)), np.full(iters or 1, float(w_new))
        w_old, w_new = np.ascontiguousarray(w_old, dtype=np.float64), np.ascontiguousarray(w_new, dtype=np.float64)
        self.ctx.check(self.ctx.lib.bhip_segchains_step(self.h, _dptr(w_old), _dptr(w_new), len(w_old)))
        self.iterations += len(w_old)

    def state(self):
        """(ll [m, n], acc [n], y0 [n, d])"""
        ll, acc, y0 = np.empty((self.m, self.n)), np.empty(self.n, dtype=np.int64), np.empty((self.n, self.d))
        self.ctx.check(self.ctx.lib.bhip_segchains_get(self.h, _dptr(ll), acc.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(y0)))
        return ll, acc, y0

    def paths(self, segment, p0=0, n=None):
        n = self.n - p0 if n is None else n
        X, W = np.empty((n, self.N, self.d)), np.empty((n, self.N, self.mp))
        self.ctx.check(self.ctx.lib.bhip_segchains_get_paths(self.h, segment, p0, n, _dptr(X), _dptr(W)))
        return X, W

    def set_proposals(self, pos):
        """hand over re-built proposals (same shapes): the chains keep their state, ll of the current paths is re-evaluated"""
        pos = list(pos)
        hs = (vp * self.m)(*[P.h for P in pos])
        self.ctx.check(self.ctx.lib.bhip_segchains_set_proposals(self.h, hs))
        self.pos = pos            # keeps the new proposals alive (and lets the old ones go)

    def adapt(self, P, L, Sigma, obs, HT, vT, means=None, set_pi0=True, newblock=True, doaccept=False):
        """Adaptive smoothing step, supplements/smoothing/smoothing.jl:130-160: re-linearise every segment's LinearAppr
        around the running mean of its paths (pooled over the ensemble unless `means` gives them), rebuild the chain of
        GuidedBridge's backwards from (HT, vT) with gpupdate at the observations obs[i] (obs[m] belongs to the right end and
        is already folded into (HT, vT)), hand the new proposals over and install pi0 = Gaussian(mu, Hermitian(Hd)) (:153) with
        the script's newblock / doaccept switches.  Returns (mu, Hd)."""
        H, v = np.array(HT, dtype=np.float64), np.array(vT, dtype=np.float64)
        new = [None] * self.m
        for i in range(self.m - 1, -1, -1):
            Y = means[i] if means is not None else self.pooled_stats(i)[0]
            new[i] = GuidedBridge(self.pos[i].tt, P, linearappr(Y), v, H, ctx=self.ctx)
            H, v = gpupdate(new[i], L, Sigma, obs[i])
        self.set_proposals(new)
        if set_pi0:
            self.set_pi0(v, cholupper_t(H), newblock=newblock, doaccept=doaccept)
        return v, H

    def set_pi0(self, mu=None, chol=None, newblock=True, doaccept=False):
        """pi0 <- Gaussian(mu, chol chol') (smoothing.jl:153); newblock: the start stays put until the first accept (:154,166-167);
        doaccept: the next iteration accepts unconditionally (:156-158,193)"""
        fl = (SEG_NEWBLOCK if newblock else 0) | (SEG_DOACCEPT if doaccept else 0)
        if mu is None:
            self.ctx.check(self.ctx.lib.bhip_segchains_set_pi0(self.h, None, None, fl))
        else:
            mu = np.ascontiguousarray(np.atleast_1d(mu), dtype=np.float64)
            self.ctx.check(self.ctx.lib.bhip_segchains_set_pi0(self.h, _dptr(mu), _dptr(_cm(np.atleast_2d(chol))), fl))

    def adapt_device(self, L, Sigma, obs, HT, vT, hwindow=0, newblock=True, doaccept=False):
        """The adaptation block of smoothing.jl:130-160 for every chain at once, on the device: every chain re-linearises its
        LinearAppr auxiliaries around ITS OWN running means (needs mcnext=True), rebuilds its GuidedBridge's backwards from
        (HT, vT) with gpupdate at obs[i] (the observation at the left end of segment i) and gets its own pi0."""
        L = np.atleast_2d(np.asarray(L, dtype=np.float64))
        mo = L.shape[0]
        obs = np.ascontiguousarray(np.asarray(obs, dtype=np.float64)[:self.m].reshape(self.m, mo))
        vT = np.ascontiguousarray(np.atleast_1d(vT), dtype=np.float64)
        fl = (SEG_NEWBLOCK if newblock else 0) | (SEG_DOACCEPT if doaccept else 0)
        self.ctx.check(self.ctx.lib.bhip_segchains_adapt_device(self.h, mo, _dptr(_cm(L)), _dptr(_cm(np.atleast_2d(Sigma))), _dptr(obs),
                                                                _dptr(_cm(np.atleast_2d(HT))), _dptr(vT), int(hwindow), fl))

    def chain_guide(self, segment, chain):
        """one chain's device-built guide of one segment: dict(B [N-1,d,d], beta [N-1,d], G [N-1,g] (the kernels' guide
        entries: d = 3 -> cofactors of Hd (row-wise 3x3), det, V), mu [d], chol [d,d])"""
        d = self.d
        g = 2 if d == 1 else 7 if d == 2 else 13
        rows, mu, ch = np.empty((self.N - 1, d * d + d + g)), np.empty(d), np.empty(d * d)
        self.ctx.check(self.ctx.lib.bhip_segchains_chain_guide(self.h, segment, chain, _dptr(rows), _dptr(mu), _dptr(ch)))
        return dict(B=_uncm(rows[:, :d * d].copy(), d, d), beta=rows[:, d * d:d * d + d].copy(), G=rows[:, d * d + d:].copy(),
                    mu=mu, chol=ch.reshape(d, d).T.copy())

    def pooled_stats(self, segment):
        """(mean [N, d], m2 [N, d, d], count) pooled over chains x iterations"""
        mean, m2, cnt = np.empty((self.N, self.d)), np.empty((self.N, self.d * self.d)), C.c_double()
        self.ctx.check(self.ctx.lib.bhip_segchains_pooled_stats(self.h, segment, _dptr(mean), _dptr(m2), C.byref(cnt)))
        return mean, _uncm(m2, self.d, self.d), cnt.value

    def statistics_info(self):
        """(K, buffers): iterations per mcnext! pass and path buffers per segment (bhip_segchains_statistics_info)"""
        k, b = C.c_int(), C.c_int()
        self.ctx.check(self.ctx.lib.bhip_segchains_statistics_info(self.h, C.byref(k), C.byref(b)))
        return k.value, b.value

    def mcstats(self, segment, chain):
        """the mcnext! state of one chain: (mean [N, d], m2 [N, d, d], count)   src/mclog.jl:48-56"""
        mean, cnt = np.empty((self.N, self.d)), C.c_int64()
        m2 = None if self.mean_only else np.empty((self.N, self.d * self.d))       # BHIP_SEGCHAINS_MCNEXT_MEAN keeps the running means only
        self.ctx.check(self.ctx.lib.bhip_segchains_mcstats(self.h, segment, chain, _dptr(mean), None if m2 is None else _dptr(m2), C.byref(cnt)))
        return mean, (None if m2 is None else _uncm(m2, self.d, self.d)), cnt.value

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.ctx.lib.bhip_segchains_destroy(self.h)
                self.h = None
        except Exception:
            pass


def mcmc(Po, x0, iterations, rho, nchains=1, seed=0, path0=0, skip=0, subsamples=None, store_X=True):
    """The MH loop of partialbridge_fitzhugh.jl:125-176 for `nchains` independent chains.
    Returns dict(chains=Chains, acc=per-chain acceptance counts, ll=final ll,
                 XX=list of saved X ensembles [n, N, d] at the `subsamples` iterations)."""
    ch = Chains(Po, x0, nchains, seed=seed, path0=path0, store_X=store_X, skip=skip)
    XX = []
    subs = sorted(set(subsamples)) if subsamples is not None else []
    if 0 in subs and store_X:
        XX.append(ch.paths(want_W=False)[0])
    done = 0
    for s in [s for s in subs if 0 < s <= iterations] + ([iterations] if iterations not in subs else []):
        ch.step(rho, s - done)
        done = s
        if s in subs and store_X:
            XX.append(ch.paths(want_W=False)[0])
    return dict(chains=ch, acc=ch.acc(), ll=ch.ll(), XX=XX)


# ----------------------------------------------------------------------------- online statistics (src/mclog.jl)
def mcstart(x):
    """mcstart(yy) -> (mean, m2, n)   src/mclog.jl:22-23 ; x: [entries, d]"""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    E, d = x.shape
    return np.zeros((E, d)), np.zeros((E, d, d)), 0


def mcnext(mc, x):
    """mcnext!(mc, x)   src/mclog.jl:48-56 (one chain value x [entries, d])"""
    m, m2, n = mc
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    delta = x - m
    m = m + delta / (n + 1)
    m2 = m2 + delta[:, :, None] * (x - m)[:, None, :]
    return m, m2, n + 1


def mcmerge(mc_a, mc_b):
    """merge two states (Chan's parallel form of the same recurrence; used across GPUs)"""
    ma, qa, na = mc_a
    mb, qb, nb = mc_b
    lib = _lib.load()
    E, d = ma.shape
    ma = np.ascontiguousarray(ma, dtype=np.float64).copy()
    qa_c = np.ascontiguousarray(_cm(qa).reshape(E, d * d)).copy()
    n = C.c_double(float(na))
    lib.bhip_welford_merge(E, d, C.byref(n), _dptr(ma), _dptr(qa_c), float(nb),
                           _dptr(np.ascontiguousarray(mb, dtype=np.float64)),
                           _dptr(np.ascontiguousarray(_cm(qb).reshape(E, d * d))))
    return ma, _uncm(qa_c, d, d), int(n.value)


def mcstats(mc):
    """mean and covariance estimates   src/mclog.jl:89-93"""
    m, m2, k = mc
    return m, m2 / (k - 1)


def mcband(mc):
    """marginal 95% band  src/mclog.jl:76-82 ; Q = sqrt(2)*erfinv(0.95)"""
    from scipy.special import erfinv
    m, m2, k = mc
    Q = math.sqrt(2.0) * erfinv(0.95)
    std = np.sqrt(np.einsum("eii->ei", m2) * (1 / (k - 1)))
    return m - Q * std, m + Q * std
