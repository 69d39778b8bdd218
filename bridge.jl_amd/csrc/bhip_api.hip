// bhip_api.hip -- implementation of the C ABI declared in include/bridgehip.h.
// Host logic only (context, proposal construction, launch orchestration); the device work is in
// bhip_path_kernel.h (instantiated per model in bhip_inst.hip) and bhip_util_kernels.h.
#include "bhip_host.hpp"
#include "bhip_path_kernel.h"
#include "bhip_chain_kernel.h"
#include "bhip_pc_kernel.h"
#include "bhip_comm.hpp"
#include "bhip_tile_kernel.h"
#include "bhip_guide_kernel.h"
#include "bhip_girsanov_kernel.h"
#include "bhip_rtc.hpp"
#include "bhip_util_kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <new>
#include <unordered_set>

using namespace bhip;

namespace bhip {
// one translation unit per model (bhip_inst.hip compiled with -DBHIP_INST=<id>)
launch_fn get_launch_ou(int, int, int, int);
launch_fn get_launch_linpro1(int, int, int, int);
launch_fn get_launch_linpro2(int, int, int, int);
launch_fn get_launch_linpro3(int, int, int, int);
launch_fn get_launch_fhn(int, int, int, int);
launch_fn get_launch_nclar(int, int, int, int);
launch_fn get_launch_intdiff(int, int, int, int);
launch_fn get_launch_lorenz(int, int, int, int);
launch_fn get_launch_fhn2(int, int, int, int);
launch_fn get_launch_pendulum(int, int, int, int);
launch_fn get_launch_wiener1(int, int, int, int);
launch_fn get_launch_wiener2(int, int, int, int);
launch_fn get_launch_wiener3(int, int, int, int);
launch_fn get_launch_ppr_lorenz(int, int);
launch_fn get_launch_mid4(int, int, int);
launch_fn get_launch_mid5(int, int, int);
launch_fn get_launch_mid6(int, int, int);
launch_fn get_launch_mid7(int, int, int);
launch_fn get_launch_mid8(int, int, int);
launch_fn get_launch_mid9(int, int, int);
launch_fn get_launch_mid10(int, int, int);
launch_fn get_launch_mid11(int, int, int);
launch_fn get_launch_mid12(int, int, int);
launch_fn get_launch_ppr_pendulum(int, int);
guide_launch_fn get_guide_launch_lorenz(int);
guide_launch_fn get_guide_launch_pendulum(int);
launch_fn get_launch_ppr_linpro1(int, int);
launch_fn get_launch_ppr_linpro2(int, int);
launch_fn get_launch_ppr_linpro3(int, int);
launch_fn get_launch_ppr_wiener1(int, int);
launch_fn get_launch_ppr_wiener2(int, int);
launch_fn get_launch_ppr_wiener3(int, int);
guide_launch_fn get_guide_launch_linpro1(int);
guide_launch_fn get_guide_launch_linpro2(int);
guide_launch_fn get_guide_launch_linpro3(int);
guide_launch_fn get_guide_launch_wiener1(int);
guide_launch_fn get_guide_launch_wiener2(int);
guide_launch_fn get_guide_launch_wiener3(int);
// the targets with a bderiv (the reference defines it for Lorenz, Pendulum, LinPro, Wiener): per-chain guide kernels
static launch_fn find_launch_ppr(const ModelHost &mh, int noise, int fl)
{
    switch (mh.id) {
    case BHIP_MODEL_LORENZ: return get_launch_ppr_lorenz(noise, fl);
    case BHIP_MODEL_PENDULUM: return get_launch_ppr_pendulum(noise, fl);
    case BHIP_MODEL_LINPRO: return mh.d == 1 ? get_launch_ppr_linpro1(noise, fl) : mh.d == 2 ? get_launch_ppr_linpro2(noise, fl) : mh.d == 3 ? get_launch_ppr_linpro3(noise, fl) : nullptr;
    case BHIP_MODEL_WIENER: return mh.d == 1 ? get_launch_ppr_wiener1(noise, fl) : mh.d == 2 ? get_launch_ppr_wiener2(noise, fl) : mh.d == 3 ? get_launch_ppr_wiener3(noise, fl) : nullptr;
    }
    return nullptr;
}
static guide_launch_fn find_guide_launch(const ModelHost &mh, int mo)
{
    switch (mh.id) {
    case BHIP_MODEL_LORENZ: return get_guide_launch_lorenz(mo);
    case BHIP_MODEL_PENDULUM: return get_guide_launch_pendulum(mo);
    case BHIP_MODEL_LINPRO: return mh.d == 1 ? get_guide_launch_linpro1(mo) : mh.d == 2 ? get_guide_launch_linpro2(mo) : mh.d == 3 ? get_guide_launch_linpro3(mo) : nullptr;
    case BHIP_MODEL_WIENER: return mh.d == 1 ? get_guide_launch_wiener1(mo) : mh.d == 2 ? get_guide_launch_wiener2(mo) : mh.d == 3 ? get_guide_launch_wiener3(mo) : nullptr;
    }
    return nullptr;
}
}  // namespace bhip
// the same getters of the fused builds (bhip_inst.hip compiled with -DBHIP_FUSED -ffp-contract=fast; their launch_fn is the
// layout-identical type of their own namespace)
namespace bhip_fused {
bhip::launch_fn get_launch_ou(int, int, int, int);
bhip::launch_fn get_launch_linpro1(int, int, int, int);
bhip::launch_fn get_launch_linpro2(int, int, int, int);
bhip::launch_fn get_launch_linpro3(int, int, int, int);
bhip::launch_fn get_launch_fhn(int, int, int, int);
bhip::launch_fn get_launch_nclar(int, int, int, int);
bhip::launch_fn get_launch_intdiff(int, int, int, int);
bhip::launch_fn get_launch_lorenz(int, int, int, int);
bhip::launch_fn get_launch_fhn2(int, int, int, int);
bhip::launch_fn get_launch_pendulum(int, int, int, int);
// (no fused Wiener: its drift is zero, there is no a*b + c to contract -- the exact build's kernels serve the option; 12 MB less library)
}  // namespace bhip_fused

#ifndef BHIP_MID_MAX_DEFAULT
#define BHIP_MID_MAX_DEFAULT 10   // largest dimension that runs one path per lane by default: 11 and 12 are instantiated but lose to the padded tile (profiles/r4_mid_dims.txt)
#endif
#ifndef BHIP_MID_MAX_CHAINS
#define BHIP_MID_MAX_CHAINS 8     // ... and the largest whose pCN chains do (16-byte slots)
#endif
#ifndef PC_FRESH_MAX_PATHS
#define PC_FRESH_MAX_PATHS 98304
#endif
// fresh proposals: up to how many paths the wave-specialised kernel is chosen over the one-lane kernel (BHIP_PC_FRESH_MAX in the
// environment: measurement hook for the A/B of the two, scripts/gpu_fresh_ab.py)
static long pc_fresh_max_paths()
{
    const char *e = getenv("BHIP_PC_FRESH_MAX");
    return e ? atol(e) : (long)PC_FRESH_MAX_PATHS;
}

// one device allocation of a chain ensemble (hipMalloc, or reserved / created / mapped through the virtual-memory API)
struct Arena {
    void *base = nullptr;
    void *base2 = nullptr;   // the proposal paths when they are an allocation of their own (large ensembles: placed, chains_place)
    bool owned = true;
    size_t bytes = 0;
    bool vmm = false;
    size_t va_bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> hsizes;
};
static void arena_free(Arena &ar);

struct bhip_ctx {
    int device = 0;
    bool host_only = false;   // device == -1: guide pre-computation only, every launch is refused
    hipStream_t stream = nullptr;
    std::string err;
    double *scratch = nullptr;
    size_t scratch_bytes = 0;
    bool wave_specialised = true;   // BHIP_OPT_WAVE_SPECIALISED: producer/consumer kernels (bhip_pc_kernel.h) where they exist
    int mid_max = BHIP_MID_MAX_DEFAULT;   // BHIP_OPT_MID_VALU: LinPro targets / component-wise drifts of dimension 4..mid_max one path per lane (0: all of them on the MFMA tile kernel)
    bool fused = false;             // BHIP_OPT_FUSED_ARITHMETIC: the d <= 3 kernels built with a*b + c contracted (tolerance parity)
    bool tune_placement = true;     // BHIP_OPT_TUNE_PLACEMENT: large chain ensembles place W and Xo in different pieces of the device memory (bhip_chains_init)
    // the piece map (chains_place below): which 96-GiB piece of the device memory the large buffers of the context's live ensembles lie in --
    // the reference points a new buffer is classified against; r_same = GB/s of two write streams into ONE piece, measured once
    struct PieceEnt { void *p; size_t bytes; int piece; };
    std::vector<PieceEnt> pieces;
    float r_same = 0.f;
    int noise_spec = 4;             // BHIP_OPT_NOISE_SPEC: 4 = bhip-philox-v4 (default), 3 = bhip-philox-v3, 2 = bhip-philox-v2 (bhip_rng.h)
    // lifetime: every proposal / chain ensemble / communicator holds a reference.  bhip_ctx_destroy with live children only
    // closes the context (garbage collectors -- Python at interpreter exit, Julia finalizers -- destroy handles in any order);
    // the last child releases it.  A closed context's stream is no longer synchronised (it was borrowed and may be gone).
    int refs = 0;
    bool closed = false;
    // buffers handed out by bhip_malloc: bhip_free releases the context's reference only for these (a foreign or already
    // freed pointer must not underflow the count and free the context under its live children)
    std::mutex buf_mu;
    std::unordered_set<void *> bufs;
};
static void ctx_free(bhip_ctx *ctx)
{
    if (ctx->scratch) { if (!ctx->closed) (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->scratch); }
    delete ctx;
}
static void ctx_retain(bhip_ctx *ctx) { __atomic_add_fetch(&ctx->refs, 1, __ATOMIC_SEQ_CST); }
static void ctx_release(bhip_ctx *ctx)
{
    if (__atomic_sub_fetch(&ctx->refs, 1, __ATOMIC_SEQ_CST) == 0 && ctx->closed) ctx_free(ctx);
}
// before a child releases device memory: wait for the context's stream unless the context was already closed
static void ctx_quiesce(bhip_ctx *ctx)
{
    if (ctx->host_only) return;
    (void)hipSetDevice(ctx->device);   // a process may drive several devices; finalizers run on whatever device is current
    if (!ctx->closed) (void)hipStreamSynchronize(ctx->stream);
    else (void)hipDeviceSynchronize();
}

struct bhip_proposal {
    bhip_ctx *ctx = nullptr;
    std::vector<double> tt;
    ModelHost mh;
    Aux aux;
    bool has_aux = false;
    Guide g;
    double *d_rows = nullptr;
    double *d_rows_innov = nullptr;   // LinPro target at 4 <= d <= 12: the (nu, H) rows innovations! reads (d_rows holds the regrouped ones)
    int rs_innov = 0;
    double *d_rows_qf = nullptr;      // LinPro target at d <= 3 with a guide: the REGROUPED rows (GUIDE_QF, dt folded in) the fused build runs on
    int rs_qf = 0;                    // under BHIP_OPT_FUSED_ARITHMETIC (do_launch); d_rows keeps the reference's form for everything else
    int rs = 0;
    double *d_rdtp = nullptr;   // rdtp[j] = sqrt(tt[j] - tt[j-1]), rdtp[0] = 0, zero padded to a multiple of 16 (bhip_pc_kernel.h)
    bool use_vend = false;
    double vend[BHIP_MAXD_LANE] = {0};
    // LinPro target of dimension 4..8: d_rows holds the coefficient rows in the (nu, H) form for the path-per-lane kernel (the
    // tile data serve its chains)
    bool mid = false;
    double *d_mpar = nullptr;
    // large-d (tile kernel) data: per-step fragment matrices, step header, constants
    double *d_steps = nullptr, *d_hdr = nullptr, *d_cst = nullptr, *d_tt = nullptr;
    bool tile_tda = false;   // the step rows carry -B~_i, c_i per step (component-wise user drift with a time-dependent auxiliary: k_tile<.., TDA = true>)
};

struct bhip_chains {
    bhip_ctx *ctx = nullptr;
    const bhip_proposal *po = nullptr;
    long n = 0, ld = 0;
    uint32_t path0 = 0;
    uint64_t seed = 0;
    int flags = 0;
    uint32_t iter = 0;
    bool inited = false;
    std::vector<double> x0;   // shared starting point (d doubles)
    bool lines = false;     // d <= 3 (noise dimension <= 3): W in the line layout of bhip_chain_kernel.h, else 16-byte slots
    bool tile = false;      // d > 3 on the MFMA tile kernel (tile-line layout); LinPro targets of dimension 4..8 stay on the path-per-lane kernel (slots)
    int nch = 0;            // lines per chain and parity half = ceil(N / 16)
    Arena arena;            // the allocation behind Wc / Xo
    double *Wc = nullptr;   // W slots [N][mp][ld][2]  |  lines [nch][ld][2][16]
    double *Xo = nullptr;   // proposal paths [N][d][ld] (BHIP_CHAINS_STORE_X); lives behind Wc in the same allocation
    // multi-segment chains with time-blocked paths (bhip_segchains.inc owns the memory): deferred proposals go there instead of Xo
    double *Xtb = nullptr, *xend = nullptr;
    long xtb_half = 0;
    const unsigned char *xsel = nullptr;   // the buffer of the ring that receives each chain's proposal (null: the half that is not cur[p])
    size_t wbytes = 0, xbytes = 0;
    // placement (bhip_chains_init): Xo allocations classified, GB/s of two write streams into one piece and into the kept (W, Xo) pair,
    // the pieces W and Xo were found in (-1: astride a cut / not placed)
    int place_tries = 0;
    float place_gbs_same = 0.f, place_gbs_kept = 0.f;
    int piece_w = -1, piece_xo = -1;
    int skip0 = 0;
    unsigned char *cur = nullptr;
    double *llcur = nullptr;
    unsigned int *acc = nullptr;
    double *statpart = nullptr;   // [256][6] per-block partial statistics
    bool shares_state = false;    // segment > 0 of a multi-segment ensemble: cur / acc / llcur belong to the owner (bhip_segchains)
    // per-chain coefficient rows / endpoint rule (owned by bhip_segchains after bhip_segchains_adapt_device), else null
    const double *prows = nullptr, *vend_pc = nullptr;
    const unsigned char *uv_pc = nullptr;
    int lna = 0;                  // the per-chain rows carry LinearNoiseAppr slopes instead of linearisation points
    int noise_spec = 4;           // the noise specification the ensemble was created under (the context's BHIP_OPT_NOISE_SPEC then); fixed for its life
};

static int fail(bhip_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}
#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(ctx, BHIP_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// end of a call that launched work reading / writing a temporary device buffer: wait for the context's stream, release the buffer, and
// report what the wait says -- an asynchronous fault of the work just launched surfaces HERE (or never: the next call would see it,
// without knowing whose it was) and must not be returned as BHIP_OK
static int sync_free_rc(bhip_ctx *ctx, void *tmp, int rc)
{
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (tmp) (void)hipFree(tmp);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, BHIP_EHIP, std::string("hipStreamSynchronize after the call's kernels: ") + hipGetErrorString(e));
    return BHIP_OK;
}

// every entry point that touches the device: refuse host-only contexts and make the context's device
// current for the calling thread (a process may drive several devices through several contexts)
#define NEED_DEVICE(ctx)                                                                          \
    do {                                                                                          \
        if ((ctx)->closed) return fail(ctx, BHIP_ESTATE, "the context was destroyed (its remaining children can only be destroyed)"); \
        if ((ctx)->host_only) return fail(ctx, BHIP_EHIP, "host-only context (device -1): no device work possible"); \
        if (hipSetDevice((ctx)->device) != hipSuccess) return fail(ctx, BHIP_EHIP, "hipSetDevice failed for the context's device"); \
    } while (0)
#define SAME_CTX(ctx, po)                                                                         \
    do {                                                                                          \
        if ((po)->ctx != (ctx)) return fail(ctx, BHIP_EINVAL, "the proposal belongs to another context"); \
    } while (0)
// the Philox counter holds the global path id in 32 bits
#define PATH_RANGE(ctx, path0, n)                                                                 \
    do {                                                                                          \
        if ((uint64_t)(path0) + (uint64_t)(n) > (1ull << 32)) return fail(ctx, BHIP_EINVAL, "path0 + npaths exceeds the 32-bit path id of the RNG counter"); \
    } while (0)

static int ensure_scratch(bhip_ctx *ctx, size_t bytes)
{
    NEED_DEVICE(ctx);
    if (ctx->scratch_bytes >= bytes) return BHIP_OK;
    if (ctx->scratch) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->scratch)); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
    HIPCHK(ctx, hipMalloc((void **)&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return BHIP_OK;
}

static launch_fn find_launch_exact(const ModelHost &mh, int gk, int mo, int noise, int fl)
{
    switch (mh.id) {
    case BHIP_MODEL_OU: return get_launch_ou(gk, mo, noise, fl);
    case BHIP_MODEL_LINPRO:
        if (mh.d == 1) return get_launch_linpro1(gk, mo, noise, fl);
        if (mh.d == 2) return get_launch_linpro2(gk, mo, noise, fl);
        if (mh.d == 3) return get_launch_linpro3(gk, mo, noise, fl);
        return nullptr;
    case BHIP_MODEL_FHN: return get_launch_fhn(gk, mo, noise, fl);
    case BHIP_MODEL_NCLAR: return get_launch_nclar(gk, mo, noise, fl);
    case BHIP_MODEL_INTDIFF: return get_launch_intdiff(gk, mo, noise, fl);
    case BHIP_MODEL_LORENZ: return get_launch_lorenz(gk, mo, noise, fl);
    case BHIP_MODEL_FHN2: return get_launch_fhn2(gk, mo, noise, fl);
    case BHIP_MODEL_PENDULUM: return get_launch_pendulum(gk, mo, noise, fl);
    case BHIP_MODEL_WIENER:
        if (mh.d == 1) return get_launch_wiener1(gk, mo, noise, fl);
        if (mh.d == 2) return get_launch_wiener2(gk, mo, noise, fl);
        if (mh.d == 3) return get_launch_wiener3(gk, mo, noise, fl);
        return nullptr;
    }
    return nullptr;
}

static launch_fn find_launch_fused(const ModelHost &mh, int gk, int mo, int noise, int fl)
{
    switch (mh.id) {
    case BHIP_MODEL_OU: return bhip_fused::get_launch_ou(gk, mo, noise, fl);
    case BHIP_MODEL_LINPRO:
        if (mh.d == 1) return bhip_fused::get_launch_linpro1(gk, mo, noise, fl);
        if (mh.d == 2) return bhip_fused::get_launch_linpro2(gk, mo, noise, fl);
        if (mh.d == 3) return bhip_fused::get_launch_linpro3(gk, mo, noise, fl);
        return nullptr;
    case BHIP_MODEL_FHN: return bhip_fused::get_launch_fhn(gk, mo, noise, fl);
    case BHIP_MODEL_NCLAR: return bhip_fused::get_launch_nclar(gk, mo, noise, fl);
    case BHIP_MODEL_INTDIFF: return bhip_fused::get_launch_intdiff(gk, mo, noise, fl);
    case BHIP_MODEL_LORENZ: return bhip_fused::get_launch_lorenz(gk, mo, noise, fl);
    case BHIP_MODEL_FHN2: return bhip_fused::get_launch_fhn2(gk, mo, noise, fl);
    case BHIP_MODEL_PENDULUM: return bhip_fused::get_launch_pendulum(gk, mo, noise, fl);
    case BHIP_MODEL_WIENER:
        if (mh.d == 1) return get_launch_wiener1(gk, mo, noise, fl);
        if (mh.d == 2) return get_launch_wiener2(gk, mo, noise, fl);
        if (mh.d == 3) return get_launch_wiener3(gk, mo, noise, fl);
        return nullptr;
    }
    return nullptr;
}

static launch_fn find_launch(const ModelHost &mh, int gk, int mo, int noise, int fl, bool fused = false)
{
    return fused ? find_launch_fused(mh, gk, mo, noise, fl) : find_launch_exact(mh, gk, mo, noise, fl);
}

extern "C" {

int bhip_version(void) { return BHIP_VERSION; }

int bhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bhip_ctx_create(int device, void *stream, bhip_ctx **out)
{
    if (!out) return BHIP_EINVAL;
    *out = nullptr;
    if (device == -1) {   // host-only context: proposals / guide coefficients can be built, nothing can run
        bhip_ctx *c = new (std::nothrow) bhip_ctx();
        if (!c) return BHIP_EHIP;
        c->device = -1; c->host_only = true;
        *out = c;
        return BHIP_OK;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return BHIP_EHIP;   // fail loudly: no CPU fallback
    if (device < 0 || device >= n) return BHIP_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return BHIP_EHIP;
    bhip_ctx *c = new (std::nothrow) bhip_ctx();
    if (!c) return BHIP_EHIP;
    c->device = device;
    c->stream = (hipStream_t)stream;
    *out = c;
    return BHIP_OK;
}

void bhip_ctx_destroy(bhip_ctx *ctx)
{
    if (!ctx) return;
    if (__atomic_load_n(&ctx->refs, __ATOMIC_SEQ_CST) > 0) {   // children alive: they keep using (and finally free) the context
        if (!ctx->host_only) (void)hipStreamSynchronize(ctx->stream);
        ctx->closed = true;
        return;
    }
    ctx_free(ctx);
}

int bhip_ctx_sync(bhip_ctx *ctx)
{
    if (!ctx) return BHIP_EINVAL;
    if (ctx->host_only) return BHIP_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BHIP_OK;
}

const char *bhip_last_error(const bhip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int bhip_ctx_set_option(bhip_ctx *ctx, int option, int value)
{
    if (!ctx) return BHIP_EINVAL;
    if (option == BHIP_OPT_WAVE_SPECIALISED) { ctx->wave_specialised = value != 0; return BHIP_OK; }
    if (option == BHIP_OPT_TUNE_PLACEMENT) { ctx->tune_placement = value != 0; return BHIP_OK; }
    if (option == BHIP_OPT_MID_VALU) {   // 0: off; 1: the default cut; 4..12: one path per lane up to that dimension
        if (value != 0 && value != 1 && (value < 4 || value > BHIP_MAXD_LANE)) return fail(ctx, BHIP_EINVAL, "BHIP_OPT_MID_VALU: 0, 1 (default) or the largest dimension 4..12 that runs one path per lane");
        ctx->mid_max = value == 0 ? 0 : value == 1 ? BHIP_MID_MAX_DEFAULT : value;
        return BHIP_OK;
    }
    if (option == BHIP_OPT_FUSED_ARITHMETIC) { ctx->fused = value != 0; return BHIP_OK; }
    if (option == BHIP_OPT_NOISE_SPEC) {
        if (value != 2 && value != 3 && value != 4)
            return fail(ctx, BHIP_EINVAL, "BHIP_OPT_NOISE_SPEC: 4 (bhip-philox-v4, the default), 3 (bhip-philox-v3) or 2 (bhip-philox-v2, full resolution)");
        ctx->noise_spec = value;
        return BHIP_OK;
    }
    return fail(ctx, BHIP_EINVAL, "bhip_ctx_set_option: unknown option");
}
int bhip_ctx_get_option(const bhip_ctx *ctx, int option, int *value)
{
    if (!ctx || !value) return BHIP_EINVAL;
    switch (option) {
    case BHIP_OPT_WAVE_SPECIALISED: *value = ctx->wave_specialised ? 1 : 0; return BHIP_OK;
    case BHIP_OPT_TUNE_PLACEMENT: *value = ctx->tune_placement ? 1 : 0; return BHIP_OK;
    case BHIP_OPT_MID_VALU: *value = ctx->mid_max; return BHIP_OK;                 // the largest dimension that runs one path per lane (0: none)
    case BHIP_OPT_FUSED_ARITHMETIC: *value = ctx->fused ? 1 : 0; return BHIP_OK;
    case BHIP_OPT_NOISE_SPEC: *value = ctx->noise_spec; return BHIP_OK;
    }
    return BHIP_EINVAL;
}

int bhip_malloc(bhip_ctx *ctx, size_t bytes, void **dev)
{
    if (!ctx || !dev) return BHIP_EINVAL;
    NEED_DEVICE(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(dev, bytes ? bytes : 8));
    { std::lock_guard<std::mutex> g(ctx->buf_mu); ctx->bufs.insert(*dev); }
    ctx_retain(ctx);   // a buffer is a child of its context too (a finalizer may free it after the context was destroyed)
    return BHIP_OK;
}
int bhip_free(bhip_ctx *ctx, void *dev)
{
    if (!ctx) return BHIP_EINVAL;
    if (!dev) return BHIP_OK;
    if (ctx->host_only) return fail(ctx, BHIP_EHIP, "host-only context (device -1): no device memory");
    {
        std::lock_guard<std::mutex> g(ctx->buf_mu);
        if (ctx->bufs.erase(dev) == 0) return fail(ctx, BHIP_EINVAL, "bhip_free: not a live bhip_malloc buffer of this context (foreign pointer or double free)");
    }
    ctx_quiesce(ctx);
    const hipError_t e = hipFree(dev);
    ctx_release(ctx);
    return e == hipSuccess ? BHIP_OK : BHIP_EHIP;
}
int bhip_memcpy_h2d(bhip_ctx *ctx, void *dev, const void *host, size_t bytes)
{
    if (!ctx) return BHIP_EINVAL;
    NEED_DEVICE(ctx);
    HIPCHK(ctx, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BHIP_OK;
}
int bhip_memcpy_d2h(bhip_ctx *ctx, void *host, const void *dev, size_t bytes)
{
    if (!ctx) return BHIP_EINVAL;
    NEED_DEVICE(ctx);
    HIPCHK(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BHIP_OK;
}
int bhip_memset(bhip_ctx *ctx, void *dev, int byte, size_t bytes)
{
    if (!ctx) return BHIP_EINVAL;
    NEED_DEVICE(ctx);
    HIPCHK(ctx, hipMemsetAsync(dev, byte, bytes, ctx->stream));
    return BHIP_OK;
}

int bhip_upload_aos(bhip_ctx *ctx, double *dev, int N, int dim, long ld, long p0, long np, const double *aos)
{
    if (!ctx || !dev || !aos || N < 1 || dim < 1 || np < 0 || p0 < 0 || p0 + np > ld) return fail(ctx, BHIP_EINVAL, "bhip_upload_aos: bad argument");
    if (np == 0) return BHIP_OK;
    const long E = (long)N * dim;
    const size_t bytes = sizeof(double) * (size_t)E * np;
    int rc = ensure_scratch(ctx, bytes);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->scratch, aos, bytes, hipMemcpyHostToDevice, ctx->stream));
    const long tot = E * np;
    hipLaunchKernelGGL(k_aos_to_soa, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, ctx->scratch, dev, E, ld, p0, np);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BHIP_OK;
}

int bhip_download_aos(bhip_ctx *ctx, const double *dev, int N, int dim, long ld, long p0, long np, double *aos)
{
    if (!ctx || !dev || !aos || N < 1 || dim < 1 || np < 0 || p0 < 0 || p0 + np > ld) return fail(ctx, BHIP_EINVAL, "bhip_download_aos: bad argument");
    if (np == 0) return BHIP_OK;
    const long E = (long)N * dim;
    const size_t bytes = sizeof(double) * (size_t)E * np;
    int rc = ensure_scratch(ctx, bytes);
    if (rc) return rc;
    const long tot = E * np;
    hipLaunchKernelGGL(k_soa_to_aos, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, dev, ctx->scratch, E, ld, p0, np);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(aos, ctx->scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BHIP_OK;
}

/* ------------------------------------------------------------------ user-defined drift (hipRTC) */
static int model_define(bhip_ctx *ctx, int d, int mp, int npar, const char *drift_src, const char *sigma_src, int *model_id)
{
    if (!ctx || !drift_src || !model_id) return BHIP_EINVAL;
    if (d > 3 && d <= 32 && mp == d && !sigma_src) {
        // (round 6) the FULL-FORM method body -- "o[0] = ...; o[1] = ...;" from (t, x, par), README.md:69-77 -- above d = 3: carried as a
        // component-wise model (bhip_model_define_components) whose component function evaluates the body into a local vector and returns
        // entry k.  One path per lane (d <= 12) the d calls of a step are inlined beside each other with k constant and the common
        // sub-expressions merge: the cost of the body once; on the tile kernel a lane holds 8 of a path's 32 rows and evaluates the whole
        // body for them (the component-wise text stays the fast form there).  Same parameter layout: [npar drift parameters, sigma (d x d)].
        std::string body = "double bhip_full_o[" + std::to_string(d) + "];\n        for (int bhip_q = 0; bhip_q < " + std::to_string(d) +
                           "; bhip_q++) bhip_full_o[bhip_q] = 0.0;\n        { double *o = bhip_full_o; (void)o;\n        " + std::string(drift_src) +
                           "\n        }\n        o = bhip_full_o[k];";
        return bhip_model_define_components(ctx, d, npar, body.c_str(), model_id);
    }
    if (d < 1 || d > 3 || mp < 1 || mp > 3)
        return fail(ctx, BHIP_EUNSUPPORTED, "bhip_model_define: d and m' in 1..3 on the path-per-lane kernel; 4 <= d <= 32 with m' = d and a constant dense sigma "
                                            "(full-form or component-wise text: bhip_model_define_components); a state-dependent sigma at d <= 3 only");
    const int derived = sigma_src ? 0 : d * mp + 2 * d * d;
    if (npar < 0 || npar + derived > 40) return fail(ctx, BHIP_EINVAL, "bhip_model_define: too many parameters (npar + d*mp + 2*d*d <= 40)");
    std::unique_ptr<UserModel> um(new UserModel());
    um->d = d; um->mp = mp; um->npar = npar; um->drift = drift_src;
    if (sigma_src) {
        um->sigma = sigma_src;
        if (um->sigma.find_first_not_of(" \t\r\n") == std::string::npos) return fail(ctx, BHIP_EINVAL, "bhip_model_define_sigma: empty sigma text");
    }
    // validate the text now (compilation needs no GPU): plain Euler-Maruyama instantiation
    std::vector<char> code;
    std::string low;
    const std::string log = rtc_compile(*um, BHIP_GUIDE_NONE, 1, NOISE_EXT, 1, code, low);
    if (!log.empty()) return fail(ctx, BHIP_EINVAL, log);
    std::lock_guard<std::mutex> lk(user_models_mutex());
    um->id = USER_MODEL_BASE + (int)user_models().size();
    *model_id = um->id;
    user_models().push_back(std::move(um));
    return BHIP_OK;
}

int bhip_model_define_components(bhip_ctx *ctx, int d, int npar, const char *component_src, int *model_id)
{
    if (!ctx || !component_src || !model_id) return BHIP_EINVAL;
    if (d < 4 || d > 32) return fail(ctx, BHIP_EUNSUPPORTED, "bhip_model_define_components: state dimension 4 <= d <= 32 (the MFMA tile kernel); d <= 3: bhip_model_define");
    if (npar < 0 || npar > 16) return fail(ctx, BHIP_EINVAL, "bhip_model_define_components: at most 16 drift parameters");
    std::unique_ptr<UserModel> um(new UserModel());
    um->d = d; um->mp = d; um->npar = npar; um->drift = component_src; um->components = true;
    std::vector<char> code;
    std::string low;
    const std::string log = rtc_tile_compile(*um, d <= 16 ? 16 : 32, 0, d != 16 && d != 32, code, low);   // validate the text now (needs no GPU)
    if (!log.empty()) return fail(ctx, BHIP_EINVAL, log);
    std::lock_guard<std::mutex> lk(user_models_mutex());
    um->id = USER_MODEL_BASE + (int)user_models().size();
    *model_id = um->id;
    user_models().push_back(std::move(um));
    return BHIP_OK;
}

int bhip_model_define(bhip_ctx *ctx, int d, int mp, int npar, const char *drift_src, int *model_id)
{
    return model_define(ctx, d, mp, npar, drift_src, nullptr, model_id);
}

int bhip_model_define_sigma(bhip_ctx *ctx, int d, int mp, int npar, const char *drift_src, const char *sigma_src, int *model_id)
{
    if (!sigma_src) return ctx ? fail(ctx, BHIP_EINVAL, "bhip_model_define_sigma: sigma text missing") : BHIP_EINVAL;
    return model_define(ctx, d, mp, npar, drift_src, sigma_src, model_id);
}

// ---- the sections of the library (one translation unit; split by section in round 6)
#include "bhip_api_proposal.inc"    /* proposals: creation, auxiliaries, guides, tile data */
#include "bhip_api_hotpath.inc"     /* launch arguments, dispatch, sample! / solve! / llikelihood ... */
#include "bhip_api_chains.inc"      /* pCN chain ensembles (includes bhip_api_placement.inc) */
#include "bhip_segchains.inc"       /* joint MH over chained segments */
#include "bhip_api_comm.inc"        /* the RCCL all-gather */
