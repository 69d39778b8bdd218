/*
 * bridgehip.h -- C ABI of the MI355X-native guided-proposal diffusion-bridge sampler.
 *
 * This is the drop-in boundary for ONE hot path of mschauer/Bridge.jl v0.11.7 (paths below are
 * relative to the reference checkout):
 *
 *     sample!(W, Wiener())                       src/wiener.jl:24-58
 *     Wo = rho*W + sqrt(1-rho^2)*W2  (pCN)       project_partialbridge/partialbridge_fitzhugh.jl:147
 *     solve!(Euler(), Xo, x0, Wo, Po)            src/euler.jl:135-152, 247-268
 *     llikelihood(LeftRule(), Xo, Po; skip)      src/guip.jl:429-438, src/partialbridge.jl:67-77,
 *                                                src/partialbridgenuH.jl:171-181, src/partialbridgen!.jl:81-97
 *     MH accept                                  project_partialbridge/partialbridge_fitzhugh.jl:160-167
 *
 * The reference has no FFI of its own (it is pure Julia); the entry points below are what a
 * `ccall` shim for that path binds -- see INTEGRATION.md and bridge.jl_amd/julia/BridgeHIP.jl.
 *
 * Conventions
 *   - every function returns 0 on success or a negative BHIP_E* code; bhip_last_error() gives text.
 *     No exceptions cross the ABI.
 *   - host buffers are caller-owned and only touched during the call. `*_dev` pointers are DEVICE
 *     pointers (hipMalloc'ed by anyone in the process: bhip_malloc, PyTorch-ROCm, AMDGPU.jl ...).
 *   - ensembles are fp64 struct-of-arrays:  element (grid index i, component k, path p) lives at
 *     dev[(i*dim + k)*ld + p]   (ld >= npaths).  One lane owns one path; a wave stores 512
 *     contiguous bytes per component per step.
 *   - host-side matrices are column-major like Julia; host AoS paths are [p][i][k]
 *     (Vector{SVector{dim}} per path, src/types.jl:71-76).
 *   - all launches go to the stream the context was created with; calls on one context are
 *     serialised by the caller, different contexts are independent.
 *   - grid indices in comments are 0-based; step i goes from tt[i] to tt[i+1].
 */
#ifndef BRIDGEHIP_H
#define BRIDGEHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BHIP_VERSION 200

/* error codes */
#define BHIP_OK 0
#define BHIP_EINVAL (-1)       /* bad argument (message names it)                                   */
#define BHIP_EHIP (-2)         /* a HIP runtime call failed                                         */
#define BHIP_EUNSUPPORTED (-3) /* no device kernel for this (model, guide, dimension) combination   */
#define BHIP_ESTATE (-4)       /* call order violated (e.g. guide not computed yet)                 */
#define BHIP_ELENGTH (-5)      /* "Y and W differ in length."  src/euler.jl:137,251                 */

/* target processes: device functors for Bridge.b / Bridge.sigma / Bridge.a methods.
 * par layouts (doubles):                                                   reference definition  */
#define BHIP_MODEL_WIENER 0   /* -                         b=0, sigma=I      src/wiener.jl:143-167  */
#define BHIP_MODEL_OU 1       /* beta, sigma               b=-beta*x         test/guip.jl:8-26      */
#define BHIP_MODEL_LINPRO 2   /* B(d*d), mu(d), sigma(d*d) b=B*(x-mu)        src/linpro.jl:65-87    */
#define BHIP_MODEL_FHN 3      /* eps,s,gamma,beta,sigma    partialbridge_fitzhugh.jl:36-46          */
#define BHIP_MODEL_NCLAR 4    /* alpha,omega,sigma         partialbridge_nclar.jl:52-61             */
#define BHIP_MODEL_INTDIFF 5  /* gamma                     test/partialbridge.jl:7-15               */
#define BHIP_MODEL_LORENZ 6   /* th1,th2,th3,s1,s2,s3      src/Models.jl:41-58                      */
#define BHIP_MODEL_FHN2 7     /* eps,s,gamma,beta,s1,s2    src/Models.jl:9-20                       */
#define BHIP_MODEL_PENDULUM 8 /* theta2, gamma             src/Models.jl:69-88                      */

/* auxiliary linear processes dX = (B(t)X + beta(t))dt + sigma(t)dW  (Bridge.B/beta/sigma/a 2-arg
 * methods, src/partialbridge.jl:13-15, src/gode.jl:2-3) */
#define BHIP_AUX_AFFINE 0        /* B(d*d), beta(d), sigma(d*mp); drift evaluated as B*x + beta      */
#define BHIP_AUX_LINPRO 1        /* B(d*d), mu(d), sigma(d*mp);  drift B*(x-mu), beta = -B*mu        */
#define BHIP_AUX_FHN_STARTEND 2  /* eps,s,gamma,beta,sigma,t0,u,T,v  partialbridge_fitzhugh.jl:58-73,102-105 */
#define BHIP_AUX_CALLBACK 3      /* user C callback (Julia @cfunction), see bhip_proposal_set_aux_callback */
#define BHIP_AUX_LINEARAPPR 4    /* LinearAppr: coefficients per grid index, see bhip_proposal_set_aux_linearappr    src/linpro.jl:181-204 */

/* guide parametrisations */
#define BHIP_GUIDE_NONE 0         /* plain Euler-Maruyama of the target            src/euler.jl:135-152 */
#define BHIP_GUIDE_HV 1           /* GuidedBridge  (Hdiamond, V)                   src/guip.jl:165-194  */
#define BHIP_GUIDE_LMMU 2         /* PartialBridge (L, M, mu, v)                   src/partialbridge.jl:33-58 */
#define BHIP_GUIDE_NUH 3          /* PartialBridgeNuH (nu, H)                      src/partialbridgenuH.jl:122-162 */
#define BHIP_GUIDE_NUH_INPLACE 4  /* PartialBridge! (nu, H), ll as two dots        src/partialbridgen!.jl:32-97 */

typedef struct bhip_ctx bhip_ctx;
typedef struct bhip_proposal bhip_proposal;
typedef struct bhip_chains bhip_chains;

/* user-defined auxiliary process: fill B (d*d col-major), beta (d), a (d*d col-major) at time t.
 * Called on the host only, O(N) times while the guide is computed. */
typedef void (*bhip_aux_fn)(double t, double *B, double *beta, double *a, void *user);

/* ------------------------------------------------------------------ context */
int bhip_version(void);
int bhip_device_count(void);
/* device: HIP ordinal; stream: a hipStream_t (NULL = the null stream).
 * device = -1 creates a HOST-ONLY context: proposals and their guide coefficients can be computed
 * and read back (bhip_proposal_guide_get), every call that needs the GPU returns BHIP_EHIP. */
int bhip_ctx_create(int device, void *stream, bhip_ctx **out);
/* Handles may be destroyed in ANY order (finalizers of a garbage collector run in no particular one): proposals, chain
 * ensembles and communicators hold a reference to their context; destroying a context that still has such children only
 * closes it -- no new work may be issued through it, the children remain destroyable -- and the last child frees it. */
void bhip_ctx_destroy(bhip_ctx *ctx);
int bhip_ctx_sync(bhip_ctx *ctx);
/* Options.  BHIP_OPT_WAVE_SPECIALISED (default 1): run fresh proposals and pCN iterations (noise dimension 1 or 2) on
 * the producer/consumer kernels -- one wave draws the Wiener noise and moves the chain state, its partner wave runs the
 * Euler recurrence and the log-likelihood; 0 selects the one-lane-does-everything kernels.  Results are bit-identical;
 * the switch exists for A/B measurements and for the test that proves the identity. */
#define BHIP_OPT_WAVE_SPECIALISED 1
/* BHIP_OPT_TUNE_PLACEMENT (default 1): placement of large chain ensembles.  On MI355X the pCN iteration (three streams: read W,
 * write Wo, write Xo) runs 14-16 % faster when the chain state W and the proposal paths Xo lie in DIFFERENT 96-GiB pieces of the
 * device's physical memory (each piece has its own DRAM banks; three streams inside one piece close each other's rows:
 * profiles/r4_placement_regions.txt).  HIP neither reports nor accepts physical addresses, but two plain write streams tell in half
 * a millisecond whether two buffers share a piece (~4.7 TB/s) or not (~6.1).  Ensembles of 1 GiB or more keep W and Xo in two
 * physically contiguous allocations; bhip_chains_init tests the pair with such streams and, only if it shares a piece, allocates
 * further candidates for Xo (8 to 24 by size, freed at once) until one lies elsewhere.  The CONTEXT keeps a piece map -- the large
 * buffers of its live ensembles with the piece each was found in -- against which new buffers are classified
 * (bhip_chains_placement_info / _pieces, bhip_ctx_piece_of).  No kernel-timed runs, no reference block: a few milliseconds per
 * ensemble.  0 keeps the pair the ensemble was created with.  Results do not depend on it. */
#define BHIP_OPT_TUNE_PLACEMENT 2
/* BHIP_OPT_MID_VALU (default 1): LinPro targets and component-wise user drifts of "middle" dimension run one path per lane like the
 * d <= 3 processes (the d x d products as scalar FMAs, coefficients through the scalar unit) in bhip_sample_solve, bhip_solve,
 * bhip_llikelihood, bhip_innovations and bhip_chains_* / bhip_segchains_* (read when the ensemble is created) instead of zero padded on
 * the 16-row MFMA tile kernel (no innovations there).  On MI355X the fp64 matrix cores have no rate advantage over fp64 FMAs, and a
 * 16x16x4 instruction cannot skip padding: d = 4 is 5.7x faster for proposals, 3.2x for chains; d = 9 1.9x for proposals.
 *   1        the default cuts: proposals / solve / llikelihood / innovations up to the dimension where the lanes stop winning
 *            (profiles/r4_mid_dims.txt), pCN chains (16-byte slots) up to d = 8;
 *   4 .. 12  one path per lane up to that dimension (chains: up to min(that, 8));
 *   0        off: every d > 3 on the tile kernel.
 * Same results to the tile kernel's tolerance (the guide solve is a product with the pre-inverted matrix in both), identical accept
 * decisions and Wiener states.  Under BHIP_OPT_NOISE_SPEC = 2 / 3 chains at d > 3 always run on the tile kernel. */
#define BHIP_OPT_MID_VALU 3
/* BHIP_OPT_FUSED_ARITHMETIC (default 0): 1 runs the d <= 3 path kernels (built-in processes; ensembles and chains with a guide
 * shared by the ensemble) from a second build of the same source in which the compiler may contract a*b + c into one fused
 * multiply-add.  The default build rounds every product like the reference (Julia never fuses) and is compared with the CPU
 * restatement bit for bit; the fused build agrees with it to 1e-9 on paths and 1e-8 on log-likelihoods (tests) -- the stated
 * fp64 tolerance -- and issues ~15 % fewer vector instructions in the instruction-bound modes.  Noise is unaffected up to the
 * last bit of the Wiener cumulation W[i] + sqrt(dt)*xi.  Per-chain device-built guides and hipRTC user processes ignore it. */
#define BHIP_OPT_FUSED_ARITHMETIC 4
/* BHIP_OPT_NOISE_SPEC (default 4): which stream of standard normals replaces the reference's randn (src/wiener.jl:31,44,55 -- a
 * ziggurat on Julia's global generator, not reproducible outside Julia).  Philox4x32-10, counter = (global path id, stream, iteration,
 * call), in all three:
 *   4  bhip-philox-v4: one call gives FOUR normals, one per 32-bit word, through a piecewise polynomial inverse of the normal
 *      distribution function (256 segments of degree 4: within 3.7e-9 of the exact quantile, Kolmogorov distance of the marginal
 *      <= 1.3e-9; |z| <= 6.34, 2^32 equiprobable values) -- the cheap normal (the reference's own randn costs a handful of
 *      instructions), the one every figure in BENCH / profiles is quoted on since round 5;
 *   3  bhip-philox-v3: one call gives FOUR normals as two Box-Muller pairs (40 bits of radius + 24 bits of angle each: |z| <= 7.45,
 *      the angle on a 2^24 grid) -- the default of rounds 3 and 4;
 *   2  bhip-philox-v2: one call gives TWO normals (one pair, 53 + 53 bits: |z| <= 8.57) -- the full-resolution stream, for callers
 *      who want nothing between them and the reference's 52-bit ziggurat but the Box-Muller map; twice the Philox calls.
 * All are bit-identical on host and device and keyed by (seed, global path id, iteration, normal index).  Takes effect for
 * everything drawn afterwards on the context (bhip_wiener_sample, bhip_sample_solve, chain and multi-segment ensembles); an
 * ensemble keeps the specification it was created under (its saved state carries it) and refuses to run under another
 * (BHIP_ESTATE).  Under 3 and 2 chains at d > 3 always run on the tile kernel and the pCN step on the 16-byte slots is refused
 * (BHIP_EUNSUPPORTED): the register-tight kernels hold the default stream only. */
#define BHIP_OPT_NOISE_SPEC 5
int bhip_ctx_set_option(bhip_ctx *ctx, int option, int value);
/* what an option stands at -- e.g. BHIP_OPT_NOISE_SPEC -> 4 | 3 | 2: a stored run can record (and a reader assert) which stream of normals
 * its paths were drawn under (the DEFAULT moved from 3 to 4 in round 5: the same (seed, path, iter) gives other normals under another
 * specification; a saved chain ensemble carries its own and refuses to resume under another).  BHIP_OPT_MID_VALU reads back as the largest
 * dimension that runs one path per lane (0: none). */
int bhip_ctx_get_option(const bhip_ctx *ctx, int option, int *value);
const char *bhip_last_error(const bhip_ctx *ctx);
/* device memory helpers for callers without their own allocator */
int bhip_malloc(bhip_ctx *ctx, size_t bytes, void **dev);
int bhip_free(bhip_ctx *ctx, void *dev);
int bhip_memcpy_h2d(bhip_ctx *ctx, void *dev, const void *host, size_t bytes);
int bhip_memcpy_d2h(bhip_ctx *ctx, void *host, const void *dev, size_t bytes);
int bhip_memset(bhip_ctx *ctx, void *dev, int byte, size_t bytes);
/* AoS host [np][N][dim] <-> SoA device (i,k,p) -> (i*dim+k)*ld + p0+p   (src/misc.jl:80 `mat`) */
int bhip_upload_aos(bhip_ctx *ctx, double *dev, int N, int dim, long ld, long p0, long np, const double *aos);
int bhip_download_aos(bhip_ctx *ctx, const double *dev, int N, int dim, long ld, long p0, long np, double *aos);

/* ------------------------------------------------------------------ user-defined target drift
 * The reference's extension point is "add a method Bridge.b(t, x, P::MyProcess)" (README.md:69-77).
 * Here the BODY of that method is given as HIP C++ text and compiled for gfx950 at run time (hipRTC):
 *     inputs  double t, const double* x (d), const double* par (npar);  output double* o (d), e.g.
 *     "o[0] = (x[0]-x[1]-x[0]*x[0]*x[0]+par[1])/par[0]; o[1] = par[2]*x[0]-x[1]+par[3];"
 * The (constant) diffusion coefficient is data: proposals on the returned model id take
 * par = [npar drift parameters, sigma (d x mp, column-major)].  d, mp <= 3 -- or 4 <= d <= 32 with mp = d (constant dense sigma, npar <= 16;
 * since round 6): the same full-form body is then carried as a component-wise model (bhip_model_define_components below: it runs everything
 * that form runs; component-wise text is the faster form on the tile kernel, where a lane holds only part of a path's state).  A syntax
 * error returns BHIP_EINVAL with the compiler log in bhip_last_error().  Kernels are compiled on first use. */
int bhip_model_define(bhip_ctx *ctx, int d, int mp, int npar, const char *drift_src, int *model_id);
/* The same with a STATE-DEPENDENT diffusion coefficient sigma(t,x,P) (the other half of the reference's
 * extension point, README.md:69-77; a = sigma*sigma' by src/types.jl:32; constdiff(P) = false):
 * sigma_src fills `double* s` (d x mp, column-major, zero-initialised) from (t, x, par), e.g. for d = mp = 2
 *     "s[0] = par[4]*sqrt(1.0 + x[0]*x[0]); s[2] = par[6]*x[1]; s[3] = par[5];"
 * Proposals on the returned id take par = the npar parameters only.  Runs: plain Euler-Maruyama, guided
 * solves of all three proposal kinds, and llikelihood / pCN chains for PartialBridge (L,M,mu), whose
 * non-constant-diffusivity terms are src/partialbridge.jl:79-84; the reference's other !constdiff
 * branches are undefined (they name unbound variables), those calls return BHIP_EUNSUPPORTED. */
int bhip_model_define_sigma(bhip_ctx *ctx, int d, int mp, int npar, const char *drift_src, const char *sigma_src, int *model_id);

/* The same extension point at LARGE state dimension (4 <= d <= 32, odd d too; m' = d, constant dense sigma): at d >= 9 the drift
 * runs on the fp64-MFMA tile kernel, where a lane holds only part of a path's state, so the method body is given COMPONENT-WISE
 * (dimensions 4..12 compile the same text into the path-per-lane kernels, as for LinPro targets: BHIP_OPT_MID_VALU):
 *     inputs  int k (component, 0-based), int d, double t, const double* x (d), const double* par (npar <= 16);  output double o, e.g.
 *     Lorenz-96:  "o = (x[(k+1)%d] - x[(k+d-2)%d])*x[(k+d-1)%d] - x[k] + par[0];"
 * Proposals on the returned model id take par = [npar drift parameters, sigma (d x d, column-major)].  Runs everything the
 * built-in LinPro target runs at large d: plain Euler-Maruyama, GuidedBridge / (nu,H) / PartialBridge guides, fused and
 * stand-alone llikelihood, pCN chains (and innovations! at d <= 12) -- with a time-constant or a time-dependent auxiliary (B~(t), beta~(t)
 * per grid point: a callback or LinearAppr coefficients; src/partialbridge.jl:13-15). */
int bhip_model_define_components(bhip_ctx *ctx, int d, int npar, const char *component_src, int *model_id);

/* ------------------------------------------------------------------ proposal  ("Po")
 * A proposal holds the grid tt (Po.tt), the target P, the auxiliary Pt and the guide coefficient
 * rows; the latter are computed on the host by backward Ralston-3 (src/ode.jl:44-49,88-97) exactly
 * as the reference constructors do, then packed per step and uploaded once. */
int bhip_proposal_create(bhip_ctx *ctx, const double *tt, int N, int model, int d, const double *par,
                         int npar, bhip_proposal **out);
void bhip_proposal_destroy(bhip_proposal *po);
int bhip_proposal_set_aux(bhip_proposal *po, int aux_kind, const double *apar, int napar);
/* drift_form: 0 -> b~ = B(t)x + beta(t);  1 -> LinPro form B(x - mu) with mu given (d doubles) */
int bhip_proposal_set_aux_callback(bhip_proposal *po, bhip_aux_fn fn, void *user, int drift_form, const double *mu);
/* Pt::LinearAppr (src/linpro.jl:181-192): the linearisation of a target along a path Y on the proposal's own grid --
 * xx[N][d] = Y, B[N][d*d] (column-major) = bderiv(t_i, y_i, P), b[N][d] = b(t_i, y_i, P), Sigma[N][d*mp] = sigma(t_i, y_i, P):
 *     _b((i,s), x, Pt) = B_i (x - xx_i) + b_i,   B((i,s), Pt) = B_i,   beta((i,s), Pt) = b_i - B_i xx_i,   a = Sigma_i Sigma_i'.
 * bhip_proposal_guide_hv then integrates (Hdiamond, V) with the index-based Heun scheme of src/guip.jl:181-189 /
 * src/ode.jl:98-113 (restated with the loop index it evidently means, see DESIGN.md: the reference constructor reads an
 * undefined variable).  The log-likelihood uses the constant-diffusivity form, i.e. Sigma_i must equal the target's sigma
 * (checked): the reference's own !constdiff branch for GuidedBridge is not defined.  d <= 3: every target; 4 <= d <= 32:
 * LinPro targets (the coefficients of grid index i enter step i's coefficient row / the tile kernel's per-step matrices) and
 * component-wise user drifts (bhip_model_define_components): beside a user drift ANY time-dependent auxiliary (this one, a callback) enters
 * the one-path-per-lane rows as B~_i, beta~_i per step and the tile kernel's step row as a third per-step matrix -B~_i with the vector
 * c_i = B~_i mu~ - beta~_i (since round 6; until then such a proposal ran one path per lane only, dimension 4..8). */
int bhip_proposal_set_aux_linearappr(bhip_proposal *po, const double *xx, const double *B, const double *b, const double *Sigma);
/* linearappr(Y, P) / linearappr!(Pt, Y, P)  src/linpro.jl:196-204 (host): fills B, b, Sigma (layouts as above) for the
 * target of `po` along Y [N][d]; for the processes the reference defines bderiv for: Lorenz, Pendulum, LinPro, Wiener. */
int bhip_linearappr(const bhip_proposal *po, const double *Y, double *B, double *b, double *Sigma);
/* LinearNoiseAppr(tt, P, x, a, direction)  src/guip.jl:114-146, the auxiliary supplements/smoothing/smoothing.jl:85 uses with
 * direction = :backward:   Y = the deterministic path y' = b(t,y,P) by Ralston-3 (forward from x at tt[1]: direction 1,
 * src/ode.jl:178-184; backward from x at tt[N]: -1, src/ode.jl:88-97; zeros: 0), B(t,P) = 0I,
 * beta((i,t)) = (Y[i] - Y[i-1])/(tt[i] - tt[i-1]), _b((i,t),x,P) = beta at max(i,2), a = the target's a.
 * bhip_linearnoiseappr_path computes Y [N][d]; bhip_proposal_set_aux_linearnoiseappr installs the auxiliary for a given Y
 * (the adaptation replaces Y by the running mean, smoothing.jl:136-139).  Targets with a host drift: Lorenz, Pendulum,
 * LinPro, Wiener.  As committed the reference type cannot reach a GuidedBridge (`_b` calls an undefined `beta_`, the
 * two-argument `a((i,t), P)` has no method): restated with the evident intention -- in the index-based Heun solver it is a
 * LinearAppr with B_i = 0, xx_i = 0, b_i = that slope, and is carried as one. */
int bhip_linearnoiseappr_path(const bhip_proposal *po, const double *x, int direction, double *Y);
int bhip_proposal_set_aux_linearnoiseappr(bhip_proposal *po, const double *Y);
/* GuidedBridge(tt, P, Pt, v, hT = 0)                                       src/guip.jl:172-180 */
int bhip_proposal_guide_hv(bhip_proposal *po, const double *v, const double *hT);
/* PartialBridge(tt, P, Pt, L, v, Sigma)                                    src/partialbridge.jl:42-50 */
int bhip_proposal_guide_lmmu(bhip_proposal *po, int m, const double *L, const double *v, const double *Sigma);
/* PartialBridgeNuH(tt, P, Pt, L, v, eps, Sigma)  (inplace=0)               src/partialbridgenuH.jl:134-145
 * PartialBridge!(tt, P, Pt, L, v, eps, Sigmanoise) (inplace=1)             src/partialbridgen!.jl:40-55 */
int bhip_proposal_guide_nuh(bhip_proposal *po, int m, const double *L, const double *v, double eps,
                            const double *Sigma, int inplace);
/* take guide arrays computed elsewhere (e.g. by Bridge.jl's own constructors):
 *   HV: A1=Hd[N][d*d] A2=V[N][d]; LMMU: A1=L[N][m*d] A2=M[N][m*m] A3=mu[N][m] A4=v[m];
 *   NUH: A1=nu[N][d] A2=H[N][d*d]   (matrices column-major) */
int bhip_proposal_guide_arrays(bhip_proposal *po, int kind, int m, const double *A1, const double *A2,
                               const double *A3, const double *A4);
/* read the guide arrays back (same layouts; NULL pointers are skipped) */
int bhip_proposal_guide_get(const bhip_proposal *po, double *A1, double *A2, double *A3, double *A4);
/* lptilde(Po, u): GuidedBridge src/guip.jl:206; NuH: -0.5*(nu1-u)'H1(nu1-u) - C (src/partialbridgenuH.jl:169,
 * with the reference's P.nu typo corrected, cf. test/partialbridgenuH.jl:124) */
int bhip_proposal_lptilde(const bhip_proposal *po, const double *u, double *out);
int bhip_proposal_info(const bhip_proposal *po, int *N, int *d, int *mp, int *m, int *kind);

/* ------------------------------------------------------------------ the hot path
 * x0: host pointer to d doubles (shared start) -- or, if x0_dev != NULL, a device array [d][ldX] of
 * per-path starting points (segment chaining, src/euler.jl:267 returns the endpoint).
 * Any output pointer may be NULL (that output is then not stored). */

/* sample!(W, Wiener{SVector{mp}}): W[0]=0, W[i+1] = W[i] + sqrt(tt[i+1]-tt[i])*xi, Philox stream
 * (seed, path0+p, iter), normals time-major/component-minor.                src/wiener.jl:24-58 */
int bhip_wiener_sample(bhip_ctx *ctx, const double *tt, int N, int mp, double *W_dev, long ld, long npaths,
                       uint64_t seed, uint32_t iter, uint32_t path0);

/* solve!(Euler(), X, x0, W, Po) on an ensemble, driven by the given W, with llikelihood fused:
 *   X_dev [N][d][ld] out, ll_dev [npaths] out = llikelihood(LeftRule(), X, Po; skip)
 * Forward EM (guide NONE): ll_dev must be NULL.                             src/euler.jl:135-152,247-268 */
int bhip_solve(bhip_ctx *ctx, const bhip_proposal *po, const double *x0, const double *x0_dev,
               const double *W_dev, long ldW, double *X_dev, long ldX, double *ll_dev, int skip, long npaths);

/* fused sample! + solve! + llikelihood with in-kernel Philox noise; W_dev (optional) receives the
 * cumulated Wiener paths so the ensemble can continue as an MCMC state. */
int bhip_sample_solve(bhip_ctx *ctx, const bhip_proposal *po, const double *x0, const double *x0_dev,
                      double *W_dev, long ldW, double *X_dev, long ldX, double *ll_dev, int skip,
                      long npaths, uint64_t seed, uint32_t iter, uint32_t path0);

/* The same with X kept in nparts (1..3) buffers, each [N][d][ldX]: paths [j*part_paths, (j+1)*part_paths) go to X_parts[j], column
 * p - j*part_paths (part_paths a multiple of 64), written by ONE launch.  Values are those of bhip_sample_solve; every reader of an
 * ensemble (bhip_llikelihood, bhip_innovations, bhip_download_aos ...) takes a part as the ensemble it is.  Why: X is this call's only
 * stream, and one write stream inside one 96-GiB piece of the device memory moves 4.3-4.4 TB/s where three streams in three pieces
 * move 6.8-6.9 (profiles/r5_three_pieces.txt); bhip_alloc_apart hands out buffers that lie in different pieces.  One path per lane
 * (d <= 3; LinPro and component-wise user drifts up to BHIP_OPT_MID_VALU); the reference has no counterpart (an ensemble is a loop). */
int bhip_sample_solve_parts(bhip_ctx *ctx, const bhip_proposal *po, const double *x0, int nparts, double *const *X_parts, long ldX,
                            long part_paths, double *ll_dev, int skip, long npaths, uint64_t seed, uint32_t iter, uint32_t path0);
/* bhip_wiener_sample / bhip_solve / bhip_llikelihood of ensembles kept in parts, by ONE launch each (two launches of half the paths leave every SIMD half of
 * the waves that hide the recurrence's latency: the stand-alone llikelihood of 262 144 stored paths 0.82 ms range by range, 0.69 in one
 * launch).  W_parts / X_parts: nparts pointers to buffers [N][.][ld], paths [j*part_paths, (j+1)*part_paths) in buffer j; the two
 * ensembles of bhip_solve_parts may be cut differently; X_parts NULL: no path store.  Shared start x0 only (per-path starts: range by range
 * through bhip_solve).  One path per lane like bhip_sample_solve_parts; the tile kernel (d > 12, or BHIP_OPT_MID_VALU = 0) returns
 * BHIP_EUNSUPPORTED and the caller walks the ranges.  The reference has no counterpart (an ensemble is a loop there). */
int bhip_wiener_sample_parts(bhip_ctx *ctx, const double *tt, int N, int mp, int nparts, double *const *W_parts, long ld, long part_paths, long npaths,
                             uint64_t seed, uint32_t iter, uint32_t path0);   /* sample!(W, Wiener()) into all buffers by one launch (mp <= 12) */
int bhip_solve_parts(bhip_ctx *ctx, const bhip_proposal *po, const double *x0, int nwparts, const double *const *W_parts, long ldW, long wpart_paths,
                     int nxparts, double *const *X_parts, long ldX, long xpart_paths, double *ll_dev, int skip, long npaths);
int bhip_llikelihood_parts(bhip_ctx *ctx, const bhip_proposal *po, int nparts, const double *const *X_parts, long ldX, long part_paths,
                           double *ll_dev, int skip, long npaths);
int bhip_girsanov_parts(bhip_ctx *ctx, const bhip_proposal *po, const double *par_t, int npar_t, int nparts, const double *const *X_parts, long ldX,
                        long part_paths, double *out_dev, long npaths);   /* bhip_girsanov (below) of an ensemble in parts, one launch */
/* nparts (1..3) physically contiguous device buffers of `bytes` each, pairwise in different pieces of the device memory (tested with
 * write streams like the placement of chain ensembles; candidates that fail stay held until the set is complete).  *apart (optional):
 * how many of them ended up pairwise apart -- nparts when all did, 0 when the buffers are too small to be tested (< 64 MiB).
 * bhip_free_apart gives them back; like bhip_malloc buffers the parts hold a reference to their context (they may be freed after
 * bhip_ctx_destroy, in any order). */
int bhip_alloc_apart(bhip_ctx *ctx, int nparts, size_t bytes, void **out, int *apart);
int bhip_free_apart(bhip_ctx *ctx, int nparts, void *const *ptrs);

/* stand-alone llikelihood(LeftRule(), X, Po; skip) of stored paths          src/guip.jl:429-438 ... */
int bhip_llikelihood(bhip_ctx *ctx, const bhip_proposal *po, const double *X_dev, long ldX, double *ll_dev,
                     int skip, long npaths);

/* innovations!(EulerMaruyama(), W, X, P): the inverse map X -> W of the Euler scheme,
 *   W[0] = 0,  W[i+1] = W[i] + inv(sigma)*(X[i+1] - X[i] - _b((i,t_i), X[i], P)*(t_{i+1}-t_i))
 * for every path of the ensemble (P: the plain target or a guided proposal); needs a square,
 * invertible sigma (d == m'); d <= 3, or a LinPro target of dimension 4..8 (inv(sigma) by LU on the host: 1e-9 against the
 * reference's sigma \ v).  src/euler.jl:358-376 -- used by the non-centred parameter updates
 * (example/fitzhugh_nagumo_full.jl:353). */
int bhip_innovations(bhip_ctx *ctx, const bhip_proposal *po, const double *X_dev, long ldX, double *W_dev, long ldW, long npaths);

/* girsanov(X, P, Pt): the discretised log-likelihood ratio dP/dPt of every stored path,
 *   out[p] = sum_i dot( Gamma(P)*(B - Bt),  X[i+1] - X[i] - 0.5*(B + Bt)*(t_{i+1}-t_i) ),
 *   B = b(t_i, X[i], P), Bt = b(t_i, X[i], Pt), Gamma(P) = inv(a(P))              src/diffusion.jl:109-123
 * P = the target of `po` (grid = po's grid); Pt = the same process type with parameters par_t
 * (the theta-update of example/fitzhugh_nagumo_full.jl:313-321), or Wiener (zero drift, test/guip.jl:72)
 * when par_t == NULL.  Needs an invertible a: OU, LinPro (d <= 3), Lorenz, Models.FitzHughNagumo. */
int bhip_girsanov(bhip_ctx *ctx, const bhip_proposal *po, const double *par_t, int npar_t, const double *X_dev, long ldX,
                  double *out_dev, long npaths);

/* gpupdate(Hd, V, L, Sigma, v) -> (Hd_out [d*d], V_out [d]): fold the observation v = L x + N(0,Sigma)
 * at the left end of a segment into (Hdiamond, V), the backward link between chained GuidedBridge
 * segments (host only).  src/guip.jl:221-231, test/smoothing.jl:73-83 */
int bhip_gpupdate(int d, int m, const double *Hd, const double *V, const double *L, const double *Sigma, const double *v,
                  double *Hd_out, double *V_out);

/* ------------------------------------------------------------------ pCN Metropolis-Hastings ensemble
 * One chain per lane.  partialbridge_fitzhugh.jl:125-176, test/partialbridgenuH.jl:155-198
 * Chain state = (W, ll, parity): the current W and the proposal Wo live in the two parity halves of the
 * chain's storage (128-byte lines per half and chunk of 16/m' grid points -- m' = 3 padded to 4 components --, 16-byte slots only for very long grids; DESIGN.md
 * section 2) and an accept flips the chain's parity bit -- the reference's `W, Wo = Wo, W` swap
 * (test/partialbridgenuH.jl:186-187) without its copies.  With BHIP_CHAINS_STORE_X the proposal path
 * Xo (solve!(Euler(), Xo, x0, Wo, Po)) of the last iteration of every bhip_chains_step call is kept in
 * the proposal buffer (earlier iterations of the same call would be overwritten unseen and skip the
 * store; call with iters = 1 to have every iteration's Xo); the CURRENT path X is a
 * deterministic function of the current W and is re-materialised on demand, bit-identical to the Xo
 * stored when that W was accepted (bhip_chains_current_X / bhip_chains_get_paths). */
#define BHIP_CHAINS_STORE_X 1   /* keep the proposal paths Xo (the SamplePath contract) */
/* A chains object borrows `po` (it must outlive the chains) and is bound to po's context. */
int bhip_chains_create(bhip_ctx *ctx, const bhip_proposal *po, long nchains, uint32_t path0, uint64_t seed,
                       int flags, bhip_chains **out);
void bhip_chains_destroy(bhip_chains *ch);
/* iteration 0: W = sample(tt, Wiener()); solve!(X, x0, W, Po); ll = llikelihood(X, Po; skip) */
int bhip_chains_init(bhip_chains *ch, const double *x0, int skip);
/* what BHIP_OPT_TUNE_PLACEMENT did for this ensemble: allocations of Xo that were tested against W (0: not placed; 1: the pair the
 * ensemble was created with already lay in different pieces), GB/s of two write streams into ONE piece (the context's reference,
 * measured once: a median of six runs) and into the (W, Xo) pair that was kept (the mean of four runs: head and tail of W against
 * head and tail of Xo; a kept pair counts as apart from 1.14 x the reference); and the pieces (ids 0..2 of the context's map, -1: astride a cut or not
 * placed) W and Xo were found in.  The reference has no counterpart (memory placement is not its concern). */
int bhip_chains_placement_info(const bhip_chains *ch, int *tries, float *gbs_same_piece, float *gbs_kept);
int bhip_chains_placement_pieces(const bhip_chains *ch, int *piece_w, int *piece_xo);
/* the piece of the device memory a buffer of >= 64 MiB lies in, by the context's map (built by the first placed ensemble, alive with
 * the ensembles): a buffer the map holds is looked up; any other is classified with write streams -- ITS CONTENTS ARE OVERWRITTEN --
 * against one representative per known piece: that piece's id, a new id when it lies apart from all of them, -1 when inconclusive.
 * The representatives are buffers of LIVE ensembles: the ranges of a representative the write streams go over (head and tail, up to
 * 512 MiB each) are copied to a scratch allocation before the test and copied back after it, on the context's stream -- the
 * ensembles' states are bit for bit what they were (up to 1 GiB of transient device memory; without it the answer is -1 and nothing
 * is written).  The same holds for the classification bhip_chains_init / bhip_segchains_init run for a new ensemble's own buffers. */
int bhip_ctx_piece_of(bhip_ctx *ctx, void *dev_ptr, size_t bytes, int *piece);
/* `iters` pCN iterations: sample!(W2); Wo = rho*W + sqrt(1-rho^2)*W2; solve!; llo; accept iff
 * log(U) <= llo - ll.  skip applies to llo like partialbridge_nclar.jl:121; pass BHIP_SKIP_OF_INIT to use the skip the
 * ensemble was initialised with, so that llo and ll sum the same terms (partialbridge_fitzhugh.jl:131,155 passes its
 * skip to the initial ll only -- with sk = 0 there; any other combination is the caller's explicit choice). */
#define BHIP_SKIP_OF_INIT (-1)
int bhip_chains_step(bhip_chains *ch, double rho, int iters, int skip);
/* The loop of partialbridge_fitzhugh.jl:143-176 for n ensembles -- one per device of a node, each created on its own context --
 * in ONE call: iteration by iteration the n launches go out round-robin, each on its context's stream (asynchronous), so a
 * single host thread (a Julia `ccall` host) keeps every device busy with one FFI crossing per call instead of one per device
 * and iteration.  Same results as bhip_chains_step on every ensemble by itself.  n <= 64, no ensemble twice.
 * Arguments, state and the context's noise specification are checked for ALL ensembles before the first launch; should a launch
 * itself fail after that (no kernel for the combination, a device error), the call returns at once and the ensembles are left at
 * DIFFERENT iteration counts -- bhip_chains_iterations says where each one stands (only completed iterations are counted; a caller
 * keeping its own counter re-reads it after an error). */
int bhip_chains_step_group(int n, bhip_chains *const *chs, double rho, int iters, int skip);
/* pCN iterations the ensemble has completed since bhip_chains_init (the `iter` word of its noise counter; what a saved state resumes from) */
int bhip_chains_iterations(const bhip_chains *ch, uint32_t *iterations);
/* device-side reduction of the ensemble statistics into stats_dev[8] =
 *   {nchains, iterations done, sum acc, sum ll, sum ll^2, min ll, max ll, sum acc^2}
 * (the block that is all-gathered over RCCL in the multi-GPU run) */
#define BHIP_STATS_LEN 8
int bhip_chains_stats(bhip_chains *ch, double *stats_dev);
/* bhip_chains_stats for the n ensembles of bhip_chains_step_group (stats_dev[k] on the device of chs[k]), one call */
int bhip_chains_stats_group(int n, bhip_chains *const *chs, double *const *stats_dev);
/* per-chain outputs (host pointers, any may be NULL): current ll, acceptance counts */
int bhip_chains_get(bhip_chains *ch, double *ll, int64_t *acc);
/* current state of chains p0..p0+np as AoS host arrays: X [np][N][d], W [np][N][mp] */
int bhip_chains_get_paths(bhip_chains *ch, long p0, long np, double *X_aos, double *W_aos);
/* current paths X of all chains into a device SoA array [N][d][ldX] */
int bhip_chains_current_X(bhip_chains *ch, double *X_dev, long ldX);
/* the proposal-path buffer Xo [N][d][*ld] written by the last iteration (needs BHIP_CHAINS_STORE_X) */
int bhip_chains_proposal_X(bhip_chains *ch, double **Xo_dev, long *ld);
/* pointwise online mean/covariance of the current X over the chain ensemble (mcstart/mcnext!
 * semantics, src/mclog.jl:22-56): mean [N][d], m2 [N][d*d] (column-major), host pointers */
int bhip_chains_pathstats(bhip_chains *ch, double *mean, double *m2);
/* Checkpoint / resume.  A chain ensemble is fully described by (current W, ll, acceptance counts, iteration counter)
 * plus the creation arguments (seed, path0): the noise is counter based, so a restored ensemble continues with
 * exactly the iterations the original would have run.  bhip_chains_save writes bhip_chains_state_bytes(ch) bytes to
 * a host buffer; bhip_chains_load restores them into an ensemble created with the same proposal shape, number of
 * chains, seed and path0 (checked against the header). */
int bhip_chains_state_bytes(const bhip_chains *ch, size_t *bytes);
int bhip_chains_save(bhip_chains *ch, void *host_buf);
int bhip_chains_load(bhip_chains *ch, const void *host_buf);
/* merge two (n, mean, m2) Welford states in place into a: parallel form of src/mclog.jl:31-38 */
int bhip_welford_merge(long entries, int d, double *na, double *mean_a, double *m2_a, double nb,
                       const double *mean_b, const double *m2_b);

/* ------------------------------------------------------------------ joint MH over chained segments (smoothing)
 * The application loop around the hot path: supplements/smoothing/smoothing.jl:99-213, test/smoothing.jl:73-92.
 * m guided proposals Po[0..m-1] (same target process, same number of grid points), linked backwards by gpupdate
 * (bhip_gpupdate) and forwards by the end point bridge! returns (src/euler.jl:267).  Per iteration and chain:
 *     y0o = mu + w_new*(rand(pi0) - mu) + w_old*(y0 - mu),  pi0 = N(mu, C C')            smoothing.jl:172
 *     y = y0o; for i: sample!(WWo[i]); WWo[i] = w_new*WWo[i] + w_old*WW[i]; y = bridge!(XXo[i], y, WWo[i], Po[i])
 *     ll = sum_i llikelihood(XXo[i], Po[i]) - llikelihood(XX[i], Po[i]);  ONE accept: log(U) <= ll  (the script's
 *          rand() < exp(ll) -- the same event; log is the reproducible form, partialbridge_fitzhugh.jl:161)
 *     accept: y0 = y0o, XX <-> XXo, WW <-> WWo;    mcstate[i] = mcnext!(mcstate[i], XX[i].yy)   (:211-213, on request)
 * The script draws rho_ = exp(-alpha*randexp()) per iteration and uses (w_new, w_old) = (sqrt(rho_), sqrt(1-rho_)): the
 * weights are given per iteration.  d <= 3.  Noise: segment i uses the Philox blocks offset by i*2^24 of stream 0,
 * the start's normals stream 2, U stream 1 (all keyed by the global chain id path0 + p). */
typedef struct bhip_segchains bhip_segchains;
#define BHIP_SEGCHAINS_MCNEXT 1   /* keep the per-chain mcnext! state (mean, m2 per grid point) of every segment on the device */
#define BHIP_SEGCHAINS_MCNEXT_MEAN 4 /* an economy for ensembles: keep the per-chain MEANS only (what the adaptation reads, smoothing.jl:133); mcnext! itself also keeps the second moments -- 3x the traffic of the commit pass */
#define BHIP_SEGCHAINS_STATS_EVERY_ITERATION 8 /* with BHIP_SEGCHAINS_MCNEXT[_MEAN] at d <= 3: one statistics pass per iteration and two path buffers per segment (the least memory) instead of one pass per twelve iterations on a ring of sixteen (bhip_segchains_statistics_info); same results */
#define BHIP_SEGCHAINS_POOLED 2   /* keep ONE state per segment pooled over chains x iterations (one extra read of the current paths per iteration) */
int bhip_segchains_create(bhip_ctx *ctx, int m, const bhip_proposal *const *pos, long nchains, uint32_t path0, uint64_t seed,
                          int flags, bhip_segchains **out);
void bhip_segchains_destroy(bhip_segchains *sc);
/* mu (d), chol = C (d*d, column-major, lower): pi0 = N(mu, C C').  Initial state: y0 = mu; per segment fresh W,
 * X = bridge!(...) chained, ll (skip applies to every llikelihood).                               smoothing.jl:99-106 */
int bhip_segchains_init(bhip_segchains *sc, const double *mu, const double *chol, int skip);
/* `iters` iterations; w_old[it], w_new[it] host arrays of length iters.  Within a call the mcnext! (and, where paths are copied on
 * accept, the commit) of an iteration runs on a second stream of the library beside the next iteration's proposals; the two
 * streams are joined before the call returns -- whatever follows on the context's stream sees the state after `iters` iterations.
 * Several iterations per call is the fast way to run the loop. */
int bhip_segchains_step(bhip_segchains *sc, const double *w_old, const double *w_new, int iters);
/* bhip_chains_placement_info of segment `segment` (large segments whose proposals go to plain path buffers -- d > 3 or pooled
 * statistics -- are placed at bhip_segchains_init: BHIP_OPT_TUNE_PLACEMENT; with d <= 3 and no pooled statistics the paths live
 * in a ring of time-blocked buffers and nothing is placed: tries = 0) */
int bhip_segchains_placement_info(const bhip_segchains *sc, int segment, int *tries, float *gbs_same_piece, float *gbs_kept);
/* how the ensemble keeps its mcnext! statistics (supplements/smoothing/smoothing.jl:211-213 updates them every iteration): *every = K,
 * the number of iterations one statistics pass covers (1: a pass per iteration) -- with d <= 3 and BHIP_SEGCHAINS_MCNEXT[_MEAN] the
 * segments' paths live in a ring of *buffers = K + L path buffers, the K current paths of a batch stay where they are until ONE pass has
 * applied the K updates of src/mclog.jl:48-56 in order (same bits; the state travels once per K iterations), running beside the first
 * L iterations of the next batch; every bhip_segchains_step call ends with a pass over what is pending.  K = 12, L = 4 (8 + 8 at the end of round 5, 4 + 4 before) unless the device
 * lacks the memory for the buffers.  Any pointer may be NULL. */
int bhip_segchains_statistics_info(const bhip_segchains *sc, int *every, int *buffers);
/* host outputs (any may be NULL): ll [m][nchains] (current, per segment), acc [nchains], y0 [nchains][d] */
int bhip_segchains_get(bhip_segchains *sc, double *ll, int64_t *acc, double *y0);
/* current paths of chains p0..p0+np of one segment as AoS host arrays: X [np][N][d], W [np][N][mp] */
int bhip_segchains_get_paths(bhip_segchains *sc, int segment, long p0, long np, double *X_aos, double *W_aos);
/* the device-resident current paths of one segment, SoA [N][d][*ld]; valid until the next bhip_segchains_step (d <= 3 without
 * pooled statistics: the ensemble keeps its paths in a ring of buffers, sixteen grid points of a chain per 128-byte line, and this
 * array is gathered from them on the context's stream when asked for) */
int bhip_segchains_current_X(bhip_segchains *sc, int segment, double **Xc_dev, long *ld);
/* the mcnext! state of ONE chain of one segment (src/mclog.jl:48-56): mean [N][d], m2 [N][d*d] (column-major), count */
int bhip_segchains_mcstats(bhip_segchains *sc, int segment, long chain, double *mean, double *m2, int64_t *count);
/* Adaptive smoothing (smoothing.jl:130-160): replace the m proposals by new ones of the same shape -- e.g. GuidedBridge's
 * whose LinearAppr auxiliaries were re-linearised around the running means (bhip_segchains_pooled_stats -> bhip_linearappr
 * -> bhip_proposal_guide_hv, linked backwards by bhip_gpupdate).  The chains keep W, X and y0; the log-likelihoods of the
 * current paths are re-evaluated under the new proposals.  The old proposals may be destroyed afterwards. */
int bhip_segchains_set_proposals(bhip_segchains *sc, const bhip_proposal *const *pos);
/* the pooled state of one segment: every chain's current path of every iteration as one sample (mcnext semantics,
 * batches merged with the parallel form, cf. bhip_welford_merge): mean [N][d], m2 [N][d*d], count = chains*iterations */
int bhip_segchains_pooled_stats(bhip_segchains *sc, int segment, double *mean, double *m2, double *count);
/* pi0 <- Gaussian(mu, chol*chol') after an adaptation (supplements/smoothing/smoothing.jl:153; mu, chol both NULL: unchanged),
 * with the script's two switches:
 *   BHIP_SEG_NEWBLOCK  the start is not moved (y0o = y0) until a proposal has been accepted          (:154,166-167,201)
 *   BHIP_SEG_DOACCEPT  the next iteration accepts unconditionally (the first adaptive proposal)       (:156-158,193) */
#define BHIP_SEG_NEWBLOCK 1
#define BHIP_SEG_DOACCEPT 2
int bhip_segchains_set_pi0(bhip_segchains *sc, const double *mu, const double *chol, int flags);
/* Guide pre-computation ON THE DEVICE, one guide per chain -- the adaptation block of supplements/smoothing/smoothing.jl:130-160
 * for every chain at once: linearappr!(Pt[i], Y, P) around the chain's OWN running mean Y = mcstate[i][1] (src/linpro.jl:196-204;
 * hwindow > 0: its moving average over j-hwindow..j+hwindow, smoothmean :136,142), GuidedBridge(tt_i, P, Pt[i], v, H) by the
 * index-based Heun solver (src/guip.jl:181-189, src/ode.jl:98-113), H, v = gpupdate(Po[i], L, Sigma, obs[i]) backwards
 * through the segments (src/guip.jl:221-231), pi0 = Gaussian(v, Hermitian(H)).
 *   requires BHIP_SEGCHAINS_MCNEXT, GuidedBridge segments with LinearAppr auxiliaries, a target whose bderiv exists on the
 *   device (Lorenz src/Models.jl:49-53, Pendulum :81-84, LinPro src/linpro.jl:82, Wiener src/wiener.jl:147; d <= 3)
 *   L [mo x d], Sigma [mo x mo] column-major; obs [m][mo]: obs[i] = the observation at the LEFT end of segment i (V.yy[i]);
 *   (HT [d x d], vT [d]) = gpupdate(prior, last observation): the right-end condition of segment m-1 (bhip_gpupdate)
 *   flags: BHIP_SEG_NEWBLOCK | BHIP_SEG_DOACCEPT as above
 * Afterwards every chain proposes with its own coefficient rows and pi0; the log-likelihoods of the current paths are
 * re-evaluated under them.  bhip_segchains_set_proposals returns the ensemble to shared proposals. */
int bhip_segchains_adapt_device(bhip_segchains *sc, int mo, const double *L, const double *Sigma, const double *obs,
                                const double *HT, const double *vT, int hwindow, int flags);
/* one chain's device-built guide: rows [N-1][d*d + d + g] = (B~_i, beta~_i, guide part: Hd and V as the kernels read them --
 * d = 1: Hd, V; d = 2: Hd (4), det, V; d = 3: cofactors of Hd (9), det, V), and its pi0: mu [d], chol [d*d] (lower, column-major) */
int bhip_segchains_chain_guide(bhip_segchains *sc, int segment, long chain, double *rows, double *mu, double *chol);

/* ------------------------------------------------------------------ multi-GPU: the one collective of the path
 * The reference has no parallelism (single-threaded Julia; the MH loops of partialbridge_fitzhugh.jl:143-176 run one
 * chain).  Here chains / proposals are sharded over GPUs by contiguous global path id (path0) with the noise keyed by
 * that id, so results do not depend on the GPU count and no path data ever crosses a link; what is exchanged is the
 * statistics block of bhip_chains_stats (and, optionally, Welford states: bhip_chains_pathstats + bhip_welford_merge)
 * with ONE RCCL all-gather over xGMI.  RCCL is loaded at the first bhip_comm_* call (dlopen librccl.so.1).
 *   one process per GPU : rank 0 calls bhip_comm_unique_id, the launcher hands the BHIP_COMM_ID_BYTES to every rank
 *                         (MPI, torch.distributed, a file ...), every rank calls bhip_comm_init_rank on its context;
 *   one process, n GPUs : bhip_comm_init_all over one context per device, collectives through bhip_comm_allgather_group.
 * The gather runs on the context's stream, after the kernels that produced the statistics; recv_dev is [nranks][count]. */
typedef struct bhip_comm bhip_comm;
#define BHIP_COMM_ID_BYTES 128
int bhip_comm_unique_id(void *id, size_t bytes);
int bhip_comm_init_rank(bhip_ctx *ctx, int nranks, int rank, const void *id, bhip_comm **out);
int bhip_comm_init_all(int ndev, bhip_ctx *const *ctxs, bhip_comm **comms_out);
int bhip_comm_info(const bhip_comm *comm, int *nranks, int *rank);
/* what RCCL ITSELF reports for the communicator -- ncclGetVersion (e.g. 22606), ncclCommCount, ncclCommUserRank -- so that a record
 * of a multi-GPU run can answer "did RCCL see N ranks" by itself (bench.py's "comm" object).  The reference has no counterpart
 * (single-threaded Julia; the unit being sharded is project_partialbridge/partialbridge_fitzhugh.jl:143-176). */
int bhip_comm_query(const bhip_comm *comm, int *rccl_version, int *rccl_nranks, int *rccl_rank);
int bhip_comm_allgather(bhip_comm *comm, const double *send_dev, double *recv_dev, size_t count);
/* = bhip_comm_allgather(comm, stats_dev, all_dev, BHIP_STATS_LEN): all_dev [nranks][BHIP_STATS_LEN] */
int bhip_comm_allgather_stats(bhip_comm *comm, const double *stats_dev, double *all_dev);
/* the two names SURVEY.md 8(b) proposed for this pair: bhip_comm_init = bhip_comm_init_all (single-process form, one
 * context per device), bhip_allgather_stats = bhip_comm_allgather_stats.
 * THREADING RULE: a communicator made by bhip_comm_init / bhip_comm_init_all has all its ranks in ONE process; RCCL then
 * needs the ranks' calls of a collective inside one group (issued one after another from one thread, the first would wait
 * for peers that are never reached).  On such a communicator with more than one rank bhip_comm_allgather,
 * bhip_comm_allgather_stats and bhip_allgather_stats return BHIP_ESTATE: use bhip_comm_allgather_group (a world of one
 * may use either). */
int bhip_comm_init(int ndev, bhip_ctx *const *ctxs, bhip_comm **comms_out);
int bhip_allgather_stats(bhip_comm *comm, const double *stats_dev, double *all_dev);
/* single-process form: the n per-device gathers inside one ncclGroupStart/End */
int bhip_comm_allgather_group(int n, bhip_comm *const *comms, const double *const *send_dev, double *const *recv_dev, size_t count);
void bhip_comm_destroy(bhip_comm *comm);

/* ------------------------------------------------------------------ RNG specification helpers (host)
 * Philox4x32-10, key = (seed lo, hi), counter = (path, stream, iter, call).  bhip_normals_host: normals n0 .. n0+n-1 of stream 0
 * under the default specification bhip-philox-v4 (call q -> normals 4q .. 4q+3, word j of the call -> normal 4q + j);
 * bhip_normals_host_spec: under `spec` = 4, 3 (call q -> two Box-Muller pairs) or 2 (bhip-philox-v2: call h -> normals 2h, 2h+1;
 * BHIP_OPT_NOISE_SPEC) -- the very values the kernels draw, bit for bit. */
void bhip_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void bhip_normals_host(uint64_t seed, uint32_t path, uint32_t iter, int n0, int n, double *z);
void bhip_normals_host_spec(int spec, uint64_t seed, uint32_t path, uint32_t iter, int n0, int n, double *z);

#ifdef __cplusplus
}
#endif
#endif
