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

/* ------------------------------------------------------------------ proposal */
int bhip_proposal_create(bhip_ctx *ctx, const double *tt, int N, int model, int d, const double *par, int npar, bhip_proposal **out)
{
    if (!ctx || !out) return BHIP_EINVAL;
    *out = nullptr;
    if (!tt || N < 2) return fail(ctx, BHIP_EINVAL, "bhip_proposal_create: need a grid with at least 2 points");
    if (npar < 0 || (npar > 0 && !par)) return fail(ctx, BHIP_EINVAL, "bhip_proposal_create: parameter vector missing");
    for (int i = 0; i + 1 < N; i++)
        if (!(tt[i + 1] > tt[i])) return fail(ctx, BHIP_EINVAL, "bhip_proposal_create: grid must be strictly increasing");
    bhip_proposal *po = new (std::nothrow) bhip_proposal();
    if (!po) return fail(ctx, BHIP_EHIP, "out of host memory");
    po->ctx = ctx;
    ctx_retain(ctx);
    po->tt.assign(tt, tt + N);
    std::string err;
    int rc = BHIP_OK;
    if (model >= USER_MODEL_BASE) {   // hipRTC-compiled drift: par = user parameters followed by sigma (d x m', column-major)
        std::lock_guard<std::mutex> lk(user_models_mutex());
        const UserModel *um = find_user_model(model);
        if (!um) { rc = BHIP_EINVAL; err = "unknown user model id"; }
        else if (d > 0 && d != um->d) { rc = BHIP_EINVAL; err = "dimension does not match the user model"; }
        else if (um->components) {   // d > 3: par = drift parameters, then the constant sigma (d x d, column-major)
            if (npar != um->npar + um->d * um->d) { rc = BHIP_EINVAL; err = "component-wise user model expects npar + d*d parameters (drift parameters, then sigma)"; }
            else {
                ModelHost &mh = po->mh;
                mh.id = model; mh.d = um->d; mh.mp = um->d;
                mh.par.assign(par, par + npar);
                const Mat S(um->d, um->d, par + um->npar);
                mh.a = outer(S);
                mh.dpar.assign(par, par + um->npar);
                if (um->d <= BHIP_MAXD_LANE) {
                    // dimensions 4..8 also run one path per lane (k_paths<MUser>, hipRTC): the device block behind the drift
                    // parameters holds sigma, a = sigma*sigma' and inv(sigma) (LU), streamed through the scalar unit like LinPro's
                    mh.dpar.insert(mh.dpar.end(), S.a.begin(), S.a.end());
                    mh.dpar.insert(mh.dpar.end(), mh.a.a.begin(), mh.a.a.end());
                    const Mat Si = det(S) != 0.0 ? inv(S) : Mat(um->d, um->d);
                    mh.dpar.insert(mh.dpar.end(), Si.a.begin(), Si.a.end());
                }
            }
        }
        else if (!um->sigma.empty()) {   // state-dependent sigma(t,x,P): nothing to derive on the host
            if (npar != um->npar) { rc = BHIP_EINVAL; err = "user model with a sigma text expects exactly its npar parameters"; }
            else {
                ModelHost &mh = po->mh;
                mh.id = model; mh.d = um->d; mh.mp = um->mp; mh.constdiff = false;
                mh.par.assign(par, par + npar);
                mh.a = Mat(um->d, um->d);
                mh.dpar = mh.par;
            }
        }
        else if (npar != um->npar + um->d * um->mp) { rc = BHIP_EINVAL; err = "user model expects npar + d*mp parameters (drift parameters, then sigma)"; }
        else {
            ModelHost &mh = po->mh;
            mh.id = model; mh.d = um->d; mh.mp = um->mp;
            mh.par.assign(par, par + npar);
            mh.a = outer(Mat(um->d, um->mp, par + um->npar));
            mh.dpar = mh.par;
            mh.dpar.insert(mh.dpar.end(), mh.a.a.begin(), mh.a.a.end());
            if (um->mp == um->d) {   // square sigma: inv(sigma) for innovations (zero block if singular)
                const Mat S(um->d, um->d, par + um->npar);
                const Mat Si = det(S) != 0.0 ? inv(S) : Mat(um->d, um->d);
                mh.dpar.insert(mh.dpar.end(), Si.a.begin(), Si.a.end());
            }
        }
    } else {
        rc = model_setup(model, d, par, npar, po->mh, err);
    }
    if (rc) { delete po; ctx_release(ctx); return fail(ctx, rc, "bhip_proposal_create: " + err); }
    po->g.kind = BHIP_GUIDE_NONE;
    *out = po;
    return BHIP_OK;
}

void bhip_proposal_destroy(bhip_proposal *po)
{
    if (!po) return;
    bhip_ctx *ctx = po->ctx;
    if (!ctx->host_only) {
        ctx_quiesce(ctx);
        if (po->d_rows) (void)hipFree(po->d_rows);
        if (po->d_rows_innov) (void)hipFree(po->d_rows_innov);
        if (po->d_rows_qf) (void)hipFree(po->d_rows_qf);
        if (po->d_tt) (void)hipFree(po->d_tt);
        if (po->d_rdtp) (void)hipFree(po->d_rdtp);
        if (po->d_steps) (void)hipFree(po->d_steps);
        if (po->d_hdr) (void)hipFree(po->d_hdr);
        if (po->d_cst) (void)hipFree(po->d_cst);
        if (po->d_mpar) (void)hipFree(po->d_mpar);
    }
    delete po;
    ctx_release(ctx);
}

int bhip_proposal_set_aux(bhip_proposal *po, int kind, const double *apar, int napar)
{
    if (!po) return BHIP_EINVAL;
    bhip_ctx *ctx = po->ctx;
    const int d = po->mh.d, mp = po->mh.mp;
    int need;
    if (kind == BHIP_AUX_AFFINE || kind == BHIP_AUX_LINPRO) need = d * d + d + d * mp;
    else if (kind == BHIP_AUX_FHN_STARTEND) { need = 9; if (d != 2 || mp != 1) return fail(ctx, BHIP_EINVAL, "FHN_STARTEND auxiliary needs d=2, scalar noise"); }
    else return fail(ctx, BHIP_EINVAL, "bhip_proposal_set_aux: unknown auxiliary kind");
    if (!apar || napar != need) return fail(ctx, BHIP_EINVAL, "bhip_proposal_set_aux: wrong number of parameters");
    po->aux = Aux();
    po->aux.kind = kind; po->aux.d = d; po->aux.mp = mp;
    po->aux.par.assign(apar, apar + napar);
    po->has_aux = true;
    return BHIP_OK;
}

// a component-wise user drift (bhip_model_define_components)
static bool components_model(const ModelHost &mh)
{
    if (mh.id < USER_MODEL_BASE || mh.d < 4) return false;
    std::lock_guard<std::mutex> lk(user_models_mutex());
    const UserModel *um = find_user_model(mh.id);
    return um && um->components;
}

int bhip_proposal_set_aux_linearappr(bhip_proposal *po, const double *xx, const double *B, const double *b, const double *Sigma)
{
    if (!po) return BHIP_EINVAL;
    bhip_ctx *ctx = po->ctx;
    if (!xx || !B || !b || !Sigma) return fail(ctx, BHIP_EINVAL, "bhip_proposal_set_aux_linearappr: null array");
    const int d = po->mh.d, mp = po->mh.mp;
    const size_t N = po->tt.size();
    // (d > 3 since round 5: LinPro targets -- one path per lane or the tile kernel, whose per-step coefficients take B~_i, beta~_i by
    // grid index like every time-dependent auxiliary; the per-chain device-built guides of bhip_segchains_adapt_device stay at d <= 3)
    // (a component-wise user drift of dimension 4..BHIP_MID_MAX_CHAINS takes them too: its one-path-per-lane rows carry B~_i, beta~_i per step;
    // the tile kernel keeps -B~ as a constant matrix beside a user drift, so such a proposal runs one path per lane only -- finish_guide)
    if (d > 3 && po->mh.id != BHIP_MODEL_LINPRO && !components_model(po->mh))
        return fail(ctx, BHIP_EUNSUPPORTED, "LinearAppr auxiliary at d > 3: LinPro targets, or component-wise user drifts (bhip_model_define_components)");
    if (!po->mh.constdiff) return fail(ctx, BHIP_EUNSUPPORTED, "LinearAppr auxiliary: the target must have a constant sigma");
    // constant-diffusivity log-likelihood: the linearisation's Sigma_i must be the target's sigma (a~ = a)
    const double *sg = po->mh.id == BHIP_MODEL_LINPRO ? po->mh.par.data() + d * d + d : nullptr;
    for (size_t i = 0; i < N; i++) {
        const Mat Si(d, mp, Sigma + i * d * mp);
        const Mat ai = outer(Si);
        for (int k = 0; k < d * d; k++)
            if (ai.a[k] != po->mh.a.a[k]) return fail(ctx, BHIP_EUNSUPPORTED, "LinearAppr auxiliary: Sigma_i*Sigma_i' differs from the target's a (state-dependent diffusivity is not supported here)");
        (void)sg;
    }
    po->aux = Aux();
    po->aux.kind = BHIP_AUX_LINEARAPPR; po->aux.d = d; po->aux.mp = mp;
    po->aux.la_tt = po->tt;
    po->aux.la_xx.assign(xx, xx + N * d);
    po->aux.la_B.assign(B, B + N * d * d);
    po->aux.la_b.assign(b, b + N * d);
    po->aux.la_S.assign(Sigma, Sigma + N * d * mp);
    po->has_aux = true;
    return BHIP_OK;
}

static Mat host_sigma(const ModelHost &mh);
static Mat host_sigma_fwd(const ModelHost &mh) { return host_sigma(mh); }
int bhip_linearappr(const bhip_proposal *po, const double *Y, double *B, double *b, double *Sigma)
{
    if (!po || !Y || !B || !b || !Sigma) return BHIP_EINVAL;
    bhip_ctx *ctx = po->ctx;
    const int d = po->mh.d, mp = po->mh.mp;
    const size_t N = po->tt.size();
    for (size_t i = 0; i < N; i++) {
        Mat J, bb;
        if (!host_bderiv(po->mh, Y + i * d, J) || !host_b(po->mh, Y + i * d, bb))
            return fail(ctx, BHIP_EUNSUPPORTED, "bhip_linearappr: bderiv is defined for Lorenz, Pendulum, LinPro and Wiener (as in the reference)");
        std::memcpy(B + i * d * d, J.a.data(), sizeof(double) * d * d);
        std::memcpy(b + i * d, bb.a.data(), sizeof(double) * d);
        const Mat S = host_sigma_fwd(po->mh);
        std::memcpy(Sigma + i * d * mp, S.a.data(), sizeof(double) * d * mp);
    }
    return BHIP_OK;
}

// sigma(t, x, P) of the built-in processes with a host drift (constant): the matrix whose outer product is the model's a
static Mat host_sigma(const ModelHost &mh)
{
    const int d = mh.d, mp = mh.mp;
    Mat S(d, mp);
    const double *p = mh.par.data();
    switch (mh.id) {
    case BHIP_MODEL_LORENZ: for (int k = 0; k < 3; k++) S(k, k) = p[3 + k]; break;
    case BHIP_MODEL_PENDULUM: S(0, 0) = 0.0; S(1, 0) = p[1]; break;
    case BHIP_MODEL_LINPRO: S = Mat(d, mp, p + d * d + d); break;
    default: for (int k = 0; k < d; k++) S(k, k) = 1.0;
    }
    return S;
}

// LinearNoiseAppr(tt, P, x, a, direction)  src/guip.jl:114-146: the deterministic path y' = b(t, y, P) by Ralston-3 --
// solve!(R3(), b, Y, x, P) forward from x at tt[1] (direction 1, src/ode.jl:178-184), solvebackward!(R3(), b, Y, x, P)
// backward from x at tt[N] (-1, src/ode.jl:88-97), zeros (0, :nothing)
int bhip_linearnoiseappr_path(const bhip_proposal *po, const double *x, int direction, double *Y)
{
    if (!po || !Y || (direction != 0 && !x)) return BHIP_EINVAL;
    bhip_ctx *ctx = po->ctx;
    const int d = po->mh.d, N = (int)po->tt.size();
    std::memset(Y, 0, sizeof(double) * N * d);
    if (direction == 0) return BHIP_OK;
    Mat probe;
    if (!host_b(po->mh, x, probe)) return fail(ctx, BHIP_EUNSUPPORTED, "bhip_linearnoiseappr_path: no host drift for this target (Lorenz, Pendulum, LinPro, Wiener have one)");
    auto F = [&](double, const Mat &y) { Mat o; host_b(po->mh, y.a.data(), o); return o; };
    Mat y(d, 1, x);
    const std::vector<double> &tt = po->tt;
    if (direction > 0) {
        std::memcpy(Y, y.a.data(), sizeof(double) * d);
        for (int i = 1; i < N; i++) { y = kernelr3(F, tt[i - 1], y, tt[i] - tt[i - 1]); std::memcpy(Y + (size_t)i * d, y.a.data(), sizeof(double) * d); }
    } else {
        std::memcpy(Y + (size_t)(N - 1) * d, y.a.data(), sizeof(double) * d);
        for (int i = N - 2; i >= 0; i--) { y = kernelr3(F, tt[i + 1], y, tt[i] - tt[i + 1]); std::memcpy(Y + (size_t)i * d, y.a.data(), sizeof(double) * d); }
    }
    return BHIP_OK;
}

// The auxiliary itself: B(t, P) = 0I, beta((i,t)) = (Y[i] - Y[i-1])/(tt[i] - tt[i-1]), _b = beta at max(i, 2), a = the target's a.
// (As committed `_b` calls an undefined `beta_` and `a((i,t), P)` has no method: restated with the evident intention)
// In the index-based Heun solver this is a LinearAppr with B_i = 0, xx_i = 0, b_i = that slope: carried as one.
int bhip_proposal_set_aux_linearnoiseappr(bhip_proposal *po, const double *Y)
{
    if (!po || !Y) return BHIP_EINVAL;
    bhip_ctx *ctx = po->ctx;
    const int d = po->mh.d, mp = po->mh.mp, N = (int)po->tt.size();
    if (d > 3 && po->mh.id != BHIP_MODEL_LINPRO) return fail(ctx, BHIP_EUNSUPPORTED, "LinearNoiseAppr auxiliary at d > 3: LinPro targets");
    if (!po->mh.constdiff) return fail(ctx, BHIP_EUNSUPPORTED, "LinearNoiseAppr auxiliary: the target must have a constant sigma");
    if (po->mh.id >= USER_MODEL_BASE || (po->mh.id != BHIP_MODEL_LORENZ && po->mh.id != BHIP_MODEL_PENDULUM && po->mh.id != BHIP_MODEL_LINPRO && po->mh.id != BHIP_MODEL_WIENER))
        return fail(ctx, BHIP_EUNSUPPORTED, "LinearNoiseAppr auxiliary: targets Lorenz, Pendulum, LinPro, Wiener");
    std::vector<double> xx((size_t)N * d, 0.0), B((size_t)N * d * d, 0.0), b((size_t)N * d), S((size_t)N * d * mp);
    const Mat Sg = host_sigma(po->mh);
    for (int j = 0; j < N; j++) {
        const int jj = j < 1 ? 1 : j;
        for (int k = 0; k < d; k++) b[(size_t)j * d + k] = (Y[(size_t)jj * d + k] - Y[(size_t)(jj - 1) * d + k]) / (po->tt[jj] - po->tt[jj - 1]);
        std::memcpy(S.data() + (size_t)j * d * mp, Sg.a.data(), sizeof(double) * d * mp);
    }
    int rc = bhip_proposal_set_aux_linearappr(po, xx.data(), B.data(), b.data(), S.data());
    if (rc) return rc;
    po->aux.la_noise = true;
    return BHIP_OK;
}

int bhip_proposal_set_aux_callback(bhip_proposal *po, bhip_aux_fn fn, void *user, int drift_form, const double *mu)
{
    if (!po) return BHIP_EINVAL;
    if (!fn) return fail(po->ctx, BHIP_EINVAL, "bhip_proposal_set_aux_callback: null callback");
    if (drift_form == 1 && !mu) return fail(po->ctx, BHIP_EINVAL, "LinPro drift form needs mu");
    po->aux = Aux();
    po->aux.kind = BHIP_AUX_CALLBACK; po->aux.d = po->mh.d; po->aux.mp = po->mh.mp;
    po->aux.fn = fn; po->aux.user = user; po->aux.cb_linpro = drift_form == 1;
    if (drift_form == 1) po->aux.cb_mu.assign(mu, mu + po->mh.d);
    po->has_aux = true;
    return BHIP_OK;
}


// ---- large state dimension: data for the MFMA tile kernel (bhip_tile_kernel.h)
// fragment order of a D x D matrix: Mf[(t*4T + ks)*64 + l] = M[16t + (l&15)][4ks + (l>>4)]
static void to_fragments(const Mat &M, double *out)
{
    const int D = M.r, T = D / 16;
    for (int t = 0; t < T; t++)
        for (int ks = 0; ks < 4 * T; ks++)
            for (int l = 0; l < 64; l++) out[((size_t)t * 4 * T + ks) * 64 + l] = M(16 * t + (l & 15), 4 * ks + (l >> 4));
}

// zero padding of a d x d matrix / d-vector to the tile kernel's dimension
static Mat pad_mat(const Mat &M, int Dp)
{
    Mat R(Dp, Dp);
    for (int j = 0; j < M.c; j++)
        for (int i = 0; i < M.r; i++) R(i, j) = M(i, j);
    return R;
}
static int tile_dim(int d) { return d <= 16 ? 16 : 32; }

static int build_tile_data(bhip_proposal *po)
{
    bhip_ctx *ctx = po->ctx;
    const int N = (int)po->tt.size(), d = po->mh.d;
    const bool plain = po->g.kind == BHIP_GUIDE_NONE;   // forward Euler-Maruyama: the guide matrices are zero
    // The tile kernel is instantiated for 16 and 32 components; every other dimension 4..31 runs zero padded (the noise keeps
    // the d-component counter layout: normal i*d + row, whatever the parity of d -- bhip_tile_kernel.h).
    const bool user = po->mh.id >= USER_MODEL_BASE;   // component-wise hipRTC drift (bhip_model_define_components): no B, mu
    if ((po->mh.id != BHIP_MODEL_LINPRO && !user) || d > 32)
        return fail(ctx, BHIP_EUNSUPPORTED, "large-d device path: LinPro target or a component-wise user drift, dimension 4 <= d <= 32");
    // the auxiliary's B~(t), beta~(t) are taken per grid point (src/partialbridge.jl:13-15: functions of t throughout the reference): constant
    // LinPro / affine forms, a caller's callback, LinearAppr / LinearNoiseAppr coefficients by grid index.  A component-wise user drift
    // keeps -B~ as a CONSTANT matrix in the kernel (bhip_tile_kernel.h, UD::ON): time-constant auxiliaries only.
    const bool aux_const = po->has_aux && (po->aux.kind == BHIP_AUX_AFFINE || po->aux.kind == BHIP_AUX_LINPRO);
    if (!plain && (!po->has_aux || po->aux.kind == BHIP_AUX_FHN_STARTEND))
        return fail(ctx, BHIP_EUNSUPPORTED, "large-d device path: no auxiliary process for a guided proposal");
    // (round 6) a component-wise user drift beside a TIME-DEPENDENT auxiliary: -B~_i and c_i travel with the step row (k_tile<.., TDA = true>)
    const bool tda = user && !plain && !aux_const;
    po->tile_tda = tda;
    const int Dp = tile_dim(d);
    // The step regrouped into products with path-independent accumulator starts (bhip_tile_kernel.h, head of the file):
    //   built-in (LinPro) target, three products:
    //     per step   A_i = -(B - B~)' Hm_i, P_i = I + dt_i (B - a Hm_i)  (fragment order),  b_i = (B - B~)' hnu_i - Hm_i' c,
    //                q_i = dt_i (a hnu_i - B mu),  dt_i, sqrt(dt_i), c0_i = c . hnu_i          (hnu_i = Hm_i nu_i, c = B~ mu~ - B mu - beta~)
    //     constant   sigma (fragment order), vend
    //   component-wise user drift (it takes the place of B (x - mu) as a vector term in the kernel: B = 0, mu = 0 here), four products:
    //     per step   -Hm_i, P_i (fragment order), hnu_i, q_i, dt_i, sqrt(dt_i)
    //     constant   sigma, B - B~ = -B~ (fragment order), vend, c
    const size_t DD = (size_t)Dp * Dp, STEP = (size_t)tile_step_doubles(Dp, tda), dd = (size_t)d * d;
    const double *par = po->mh.par.data();
    const double *sig = user ? par + (po->mh.par.size() - (size_t)dd) : par + dd + d;   // user: [drift parameters, sigma]; LinPro: [B, mu, sigma]
    const Mat Bm = user ? Mat(d, d) : Mat(d, d, par);
    Mat mu(d, 1);
    if (!user) std::memcpy(mu.a.data(), par + dd, sizeof(double) * d);
    const Mat Bmu = Bm * mu;
    // (the auxiliary at grid point i; for the constant forms the same matrices every time)
    auto aux_at = [&](int i, Mat &Bt, Mat &mua, Mat &beta) {
        Bt = Mat(d, d); mua = Mat(d, 1); beta = Mat(d, 1);
        if (plain) return;
        Bt = po->aux.B(po->tt[i]);
        if (po->aux.linpro_form()) std::memcpy(mua.a.data(), po->aux.mu(), sizeof(double) * d);   // B~ (x - mu~)
        else beta = po->aux.beta(po->tt[i]);                                                        // B~ x + beta~
    };
    Mat Bt, mua, beta;
    aux_at(0, Bt, mua, beta);
    Mat Dm = Bm - Bt, DmT = tr(Dm), c = Bt * mua - Bmu - beta;
    Mat Id(d, d);
    for (int k = 0; k < d; k++) Id(k, k) = 1.0;
    std::vector<double> steps((size_t)(N - 1) * STEP, 0.0), hdr((size_t)(N - 1) * 2);
    for (int i = 0; i < N - 1; i++) {
        // Every guide is brought to the form r = Hm_i (nu_i - x):
        //   GuidedBridge : Hdiamond_i \ (V_i - x)  ->  Hm = inv(Hdiamond_i) (LU, path-independent), nu = V_i
        //   (nu,H)       : H_i (nu_i - x) as in the reference (PartialBridge! likewise)
        //   PartialBridge: L'M(v - mu - Lx) = (L'ML)(nu - x) with any nu solving L nu = v - mu: nu = L'(LL')^-1 (v - mu)
        Mat Hm, nu;
        if (plain) { Hm = Mat(d, d); nu = Mat(d, 1); }
        else if (po->g.kind == BHIP_GUIDE_HV) { Hm = inv(po->g.Hd[i]); nu = po->g.V[i]; }
        else if (po->g.kind == BHIP_GUIDE_LMMU) {
            const Mat &L = po->g.L[i];
            Hm = (tr(L) * po->g.M[i]) * L;
            nu = tr(L) * solve(L * tr(L), po->g.v - po->g.mu[i]);
        } else { Hm = po->g.H[i]; nu = po->g.nu[i]; }
        const double dt = po->tt[i + 1] - po->tt[i];
        if (!aux_const && !plain) {   // b~(t_i, .) of the log-likelihood's left rule (src/guip.jl:434): the coefficients of grid point i
            aux_at(i, Bt, mua, beta);
            Dm = Bm - Bt; DmT = tr(Dm); c = Bt * mua - Bmu - beta;
        }
        const Mat aHm = po->mh.a * Hm, hnu = Hm * nu;
        const Mat P = Id + dt * (Bm - aHm), q = dt * (po->mh.a * hnu - Bmu);
        double *st = &steps[(size_t)i * STEP];
        if (user) {
            to_fragments(pad_mat(-Hm, Dp), st);
            std::memcpy(st + 2 * DD, hnu.a.data(), sizeof(double) * d);
            if (tda) {   // B = 0, mu = 0 beside a user drift: Dm = -B~_i, c = B~_i mu~ - beta~_i of grid point i (set just above)
                to_fragments(pad_mat(Dm, Dp), st + 2 * DD + 2 * Dp + 4);
                std::memcpy(st + 3 * DD + 2 * Dp + 4, c.a.data(), sizeof(double) * d);
            }
        } else {
            const Mat A = -(DmT * Hm), b = DmT * hnu - tr(Hm) * c;
            to_fragments(pad_mat(A, Dp), st);
            std::memcpy(st + 2 * DD, b.a.data(), sizeof(double) * d);
            st[2 * DD + 2 * Dp + 2] = dot(c, hnu);
        }
        to_fragments(pad_mat(P, Dp), st + DD);
        std::memcpy(st + 2 * DD + Dp, q.a.data(), sizeof(double) * d);
        hdr[2 * i] = dt;
        hdr[2 * i + 1] = std::sqrt(dt);
        st[2 * DD + 2 * Dp] = hdr[2 * i];           // travel to LDS with the step's matrices (no separate load in the time loop)
        st[2 * DD + 2 * Dp + 1] = hdr[2 * i + 1];
    }
    std::vector<double> cst(2 * DD + 2 * Dp, 0.0);
    to_fragments(pad_mat(Mat(d, d, sig), Dp), &cst[0]);
    if (user) {
        to_fragments(pad_mat(Dm, Dp), &cst[DD]);
        std::memcpy(&cst[2 * DD + Dp], c.a.data(), sizeof(double) * d);
    }
    if (po->g.kind == BHIP_GUIDE_HV) std::memcpy(&cst[2 * DD], po->g.V[N - 1].a.data(), sizeof(double) * d);       // vend
    for (double **q : {&po->d_steps, &po->d_hdr, &po->d_cst, &po->d_tt})
        if (*q) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(*q)); *q = nullptr; }
    if (user) {   // b_k(t, x, P) may depend on t
        HIPCHK(ctx, hipMalloc((void **)&po->d_tt, sizeof(double) * N));
        HIPCHK(ctx, hipMemcpy(po->d_tt, po->tt.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    }
    HIPCHK(ctx, hipMalloc((void **)&po->d_steps, sizeof(double) * steps.size()));
    HIPCHK(ctx, hipMalloc((void **)&po->d_hdr, sizeof(double) * hdr.size()));
    HIPCHK(ctx, hipMalloc((void **)&po->d_cst, sizeof(double) * cst.size()));
    HIPCHK(ctx, hipMemcpy(po->d_steps, steps.data(), sizeof(double) * steps.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(po->d_hdr, hdr.data(), sizeof(double) * hdr.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(po->d_cst, cst.data(), sizeof(double) * cst.size(), hipMemcpyHostToDevice));
    return BHIP_OK;
}

static int launch_tile_path(const bhip_proposal *po_c, const double *x0, const double *W_in, long ldWin, double *W_out, long ldWout,
                            double *X, long ldX, double *ll, int skip, long npaths, int noise, uint64_t seed, uint32_t iter, uint32_t path0,
                            int wstride = 1, const bhip_chains *ch = nullptr, double rho = 0.0, const double *x0_dev = nullptr, long ldx0 = 0,
                            uint32_t blk0 = 0, int defer_accept = 0, double w_new = -1.0 /* >= 0: the weight of the fresh noise given explicitly */)
{
    const bhip_proposal *po = po_c;
    bhip_ctx *ctx = po->ctx;
    NEED_DEVICE(ctx);
    const int d = po->mh.d;
    if (!po->d_steps) return fail(ctx, BHIP_ESTATE, "proposal has no large-d guide data (compute a guide first)");
    if (!x0 && !x0_dev) return fail(ctx, BHIP_EINVAL, "need a starting point");
    if (npaths < 1 || skip < 0) return fail(ctx, BHIP_EINVAL, "bad npaths/skip");
    TArgs a;
    std::memset(&a, 0, sizeof(a));
    if (x0) std::memcpy(a.x0, x0, sizeof(double) * d);   // by value in the kernel arguments (rows d..Dp-1 stay zero)
    a.x0_dev = x0_dev; a.ldx0 = ldx0;
    a.steps = po->d_steps; a.hdr = po->d_hdr; a.cst = po->d_cst;
    a.dtrue = d;
    a.N = (int)po->tt.size(); a.skip = skip; a.use_vend = po->use_vend; a.noise = noise; a.P = npaths;
    a.Win = W_in; a.ldWin = ldWin; a.Wout = W_out; a.ldWout = ldWout; a.X = X; a.ldX = ldX; a.ll = ll;
    a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.iter = iter; a.path0 = path0;
    a.wstride = wstride;
    a.noise_spec = ctx->noise_spec;
    if (noise == 2) {   // pCN chain step
        a.Wc = ch->Wc; a.ldC = ch->ld; a.cur = ch->cur; a.llcur = ch->llcur; a.acc = ch->acc;
        a.rho = rho; a.srho = w_new >= 0.0 ? w_new : std::sqrt(1 - rho * rho);
    }
    a.blk0 = blk0; a.defer_accept = defer_accept;
    if (po->mh.id >= USER_MODEL_BASE) {   // component-wise user drift: the hipRTC instantiation k_tile<D, noise, PAD, MUserBig>
        const int D = tile_dim(d);
        const bool pad = d != D;
        for (size_t k = 0; k < po->mh.par.size() - (size_t)po->mh.d * po->mh.d && k < 16; k++) a.upar[k] = po->mh.dpar[k];   // the drift parameters
        a.tt = po->d_tt;
        hipFunction_t fn = nullptr;
        {
            std::lock_guard<std::mutex> lk(user_models_mutex());
            UserModel *um = find_user_model(po->mh.id);
            if (!um || !um->components) return fail(ctx, BHIP_EINVAL, "unknown component-wise user model id");
            const std::vector<int> key = {ctx->device, -1, D, noise, pad ? 1 : 0, po->tile_tda ? 1 : 0};
            auto it = um->fns.find(key);
            if (it == um->fns.end()) {
                const std::string log = rtc_tile_build(*um, D, noise, pad, &fn, po->tile_tda);
                if (!log.empty()) return fail(ctx, BHIP_EHIP, log);
                um->fns[key] = fn;
            } else fn = it->second;
        }
        TArgs args = a;
        void *params[] = {&args};
        HIPCHK(ctx, hipModuleLaunchKernel(fn, (unsigned)((npaths + 63) / 64), 1, 1, 256, 1, 1, (unsigned)tile_lds_bytes(D, true, true, po->tile_tda), ctx->stream, params, nullptr));
        return BHIP_OK;
    }
    hipError_t le = hipSuccess;
    if (d == 32) le = launch_tile_noise<32, false>(a, noise, ctx->stream);
    else if (d == 16) le = launch_tile_noise<16, false>(a, noise, ctx->stream);
    else if (tile_dim(d) == 32) le = launch_tile_noise<32, true>(a, noise, ctx->stream);
    else le = launch_tile_noise<16, true>(a, noise, ctx->stream);
    HIPCHK(ctx, le);
    return BHIP_OK;
}

static int finish_guide(bhip_proposal *po)
{
    bhip_ctx *ctx = po->ctx;
    const int N = (int)po->tt.size(), d = po->mh.d;
    po->use_vend = false;
    if (po->g.kind == BHIP_GUIDE_HV) {   // endpoint(y, P::GuidedBridge)  src/euler.jl:241-242
        double n1 = 0;
        for (double x : po->g.Hd[N - 1].a) n1 += std::fabs(x);
        if (n1 < 2.220446049250313e-16) {
            po->use_vend = true;
            for (int k = 0; k < d && k < BHIP_MAXD_LANE; k++) po->vend[k] = po->g.V[N - 1].a[k];
        }
    }
    if (ctx->host_only) return BHIP_OK;   // coefficients stay on the host (bhip_proposal_guide_get)
    po->mid = false;
    if (d > 3) {
        bool comp = false;
        if (po->mh.id >= USER_MODEL_BASE) {
            std::lock_guard<std::mutex> lk(user_models_mutex());
            const UserModel *um = find_user_model(po->mh.id);
            comp = um && um->components;
        }
        // (until round 5 a component-wise user drift with a TIME-DEPENDENT auxiliary was built for the lanes alone, dimension 4..8; the tile
        // kernel now streams -B~_i, c_i with the step row -- build_tile_data, k_tile<.., TDA> -- and takes every dimension 4..32)
        const int rct = build_tile_data(po);
        if (rct || d > BHIP_MAXD_LANE || !(po->mh.id == BHIP_MODEL_LINPRO || comp)) return rct;
        po->mid = true;   // ... and the rows below, for one path per lane (LinPro targets and component-wise user drifts)
    }
    std::vector<double> rows;
    int rs = 0;
    // the guide in the form r = H_i (nu_i - x), as build_tile_data brings it for the tile kernel:
    //   GuidedBridge: H = inv(Hdiamond_i) (LU, path-independent), nu = V_i;  (L,M,mu): H = L'ML, nu = L'(LL')^-1 (v - mu);  (nu,H) as is
    auto nuh_form = [&](Guide &g2) {
        g2.kind = po->g.kind == BHIP_GUIDE_NONE ? BHIP_GUIDE_NONE : BHIP_GUIDE_NUH;
        g2.m = po->g.m;
        if (g2.kind == BHIP_GUIDE_NONE) return;
        g2.H.resize(N); g2.nu.resize(N);
        for (int i = 0; i < N; i++) {
            if (po->g.kind == BHIP_GUIDE_HV) { g2.H[i] = i < N - 1 ? inv(po->g.Hd[i]) : Mat(d, d); g2.nu[i] = po->g.V[i]; }
            else if (po->g.kind == BHIP_GUIDE_LMMU) {
                const Mat &L = po->g.L[i];
                g2.H[i] = (tr(L) * po->g.M[i]) * L;
                g2.nu[i] = tr(L) * solve(L * tr(L), po->g.v - po->g.mu[i]);
            } else { g2.H[i] = po->g.H[i]; g2.nu[i] = po->g.nu[i]; }
        }
    };
    // LinPro target: the REGROUPED step (bhip_path_kernel.h GUIDE_QF; the algebra of build_tile_data): per step A_i, bv_i, P_i, q_i, c0_i in
    // place of B~_i, beta~_i, H_i, nu_i, from the (nu, H) form g2.  times[i*ts .. +2] = t, dt, sqrt(dt) of step i.  fold_dt (the d <= 3
    // form): A_i, bv_i, c0_i multiplied by dt here, so that the step adds c0' + x.(bv' + A'x) to the log-likelihood as it is.
    auto regroup_rows = [&](const Guide &g2, const double *times, int ts, bool fold_dt, std::vector<double> &rowsq, int &rq) {
        rq = row_stride(BHIP_GUIDE_QF, d, 1, true);
        rowsq.assign((size_t)(N - 1) * rq, 0.0);
        const double *par = po->mh.par.data();
        const Mat Bm(d, d, par), mu(d, 1, par + (size_t)d * d);
        const Mat Bmu = Bm * mu;
        Mat Id(d, d);
        for (int k = 0; k < d; k++) Id(k, k) = 1.0;
        for (int i = 0; i < N - 1; i++) {
            Mat Bt = po->aux.B(po->tt[i]), mua(d, 1), beta(d, 1);
            if (po->aux.linpro_form()) std::memcpy(mua.a.data(), po->aux.mu(), sizeof(double) * d);
            else beta = po->aux.beta(po->tt[i]);
            const Mat Dm = Bm - Bt, DmT = tr(Dm), c = Bt * mua - Bmu - beta;
            const Mat &Hm = g2.H[i], &nu = g2.nu[i];
            const double dt = po->tt[i + 1] - po->tt[i], f = fold_dt ? dt : 1.0;
            const Mat hnu = Hm * nu;
            const Mat A = (-f) * (DmT * Hm), bv = f * (DmT * hnu - tr(Hm) * c);
            const Mat P = Id + dt * (Bm - po->mh.a * Hm), q = dt * (po->mh.a * hnu - Bmu);
            double *r = &rowsq[(size_t)i * rq];
            std::memcpy(r, times + (size_t)i * ts, 3 * sizeof(double));                       // t, dt, sqrt(dt)
            std::memcpy(r + 3, A.a.data(), sizeof(double) * d * d);
            std::memcpy(r + 3 + d * d, bv.a.data(), sizeof(double) * d);
            std::memcpy(r + 3 + d * d + d, P.a.data(), sizeof(double) * d * d);
            std::memcpy(r + 3 + 2 * d * d + d, q.a.data(), sizeof(double) * d);
            r[3 + 2 * d * d + 2 * d] = f * dot(c, hnu);
        }
    };
    if (po->d_rows_qf) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(po->d_rows_qf)); po->d_rows_qf = nullptr; po->rs_qf = 0; }
    if (po->mid) {
        Guide g2;
        nuh_form(g2);
        pack_rows(po->tt, po->mh, po->has_aux ? &po->aux : nullptr, g2, rows, rs);
        if (po->d_rows_innov) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(po->d_rows_innov)); po->d_rows_innov = nullptr; }
        if (po->mh.id == BHIP_MODEL_LINPRO && g2.kind != BHIP_GUIDE_NONE) {
            // innovations! keeps the (nu, H) rows packed above (a second, small array)
            HIPCHK(ctx, hipMalloc((void **)&po->d_rows_innov, sizeof(double) * rows.size()));
            HIPCHK(ctx, hipMemcpy(po->d_rows_innov, rows.data(), sizeof(double) * rows.size(), hipMemcpyHostToDevice));
            po->rs_innov = rs;
            std::vector<double> rowsq;
            int rq = 0;
            regroup_rows(g2, rows.data(), rs, false, rowsq, rq);
            rows.swap(rowsq);
            rs = rq;
        }
        if (!po->d_mpar) HIPCHK(ctx, hipMalloc((void **)&po->d_mpar, sizeof(double) * po->mh.dpar.size()));
        HIPCHK(ctx, hipMemcpy(po->d_mpar, po->mh.dpar.data(), sizeof(double) * po->mh.dpar.size(), hipMemcpyHostToDevice));
    } else {
        pack_rows(po->tt, po->mh, po->has_aux ? &po->aux : nullptr, po->g, rows, rs);
        // d <= 3, LinPro target, any guide, any auxiliary: the regrouped rows beside the reference-form ones -- what the fused build's
        // GUIDE_QF kernels read when the context runs under BHIP_OPT_FUSED_ARITHMETIC (do_launch).  Needs the (nu, H) form: a GuidedBridge
        // whose Hdiamond_i cannot be inverted keeps the reference form alone.
        if (po->mh.id == BHIP_MODEL_LINPRO && po->mh.constdiff && po->g.kind != BHIP_GUIDE_NONE && po->has_aux) {
            bool ok = true;
            if (po->g.kind == BHIP_GUIDE_HV)
                for (int i = 0; i < N - 1 && ok; i++) { const double c = std::fabs(det(po->g.Hd[i])); ok = c > 0x1.0p-200 && c < 0x1.0p200; }
            if (ok) {
                Guide g2;
                nuh_form(g2);
                std::vector<double> rowsq;
                int rq = 0;
                regroup_rows(g2, rows.data(), rs, true, rowsq, rq);
                for (double x : rowsq) ok = ok && std::isfinite(x);
                if (ok) {
                    HIPCHK(ctx, hipMalloc((void **)&po->d_rows_qf, sizeof(double) * rowsq.size()));
                    HIPCHK(ctx, hipMemcpy(po->d_rows_qf, rowsq.data(), sizeof(double) * rowsq.size(), hipMemcpyHostToDevice));
                    po->rs_qf = rq;
                }
            }
        }
    }
    if (po->g.kind == BHIP_GUIDE_HV && po->mid) {
        // 4 <= d <= 8: the rows above hold inv(Hdiamond_i) (the (nu, H) form); what has to hold is that the inverse exists
        for (int i = 0; i < N - 1; i++) {
            const double c = std::fabs(det(po->g.Hd[i]));
            if (!(c > 0.0 && std::isfinite(c)))
                return fail(ctx, BHIP_EUNSUPPORTED, "GuidedBridge: Hdiamond[" + std::to_string(i) + "] is singular or not finite");
        }
    } else if (po->g.kind == BHIP_GUIDE_HV) {
        // d <= 3 (the GUIDE_HV row layout): the kernels divide by Hd_i (d = 1) / det(Hd_i) (d = 2, 3) through the row's reciprocal
        // (bhip_smallmat.h sm_div_by): the bits of `/` as long as the hardware division would not pre-scale its operands
        const int c_at = 3 + d * d + d + (d == 1 ? 0 : d == 2 ? 4 : 9);
        for (int i = 0; i < N - 1; i++) {
            const double c = std::fabs(rows[(size_t)i * rs + c_at]);
            if (!(c > 0x1.0p-200 && c < 0x1.0p200))
                return fail(ctx, BHIP_EUNSUPPORTED, "GuidedBridge: Hdiamond[" + std::to_string(i) + "] is singular, not finite or outside 2^-200 < |det| < 2^200: "
                                                    "no device kernel divides by it the way the reference's `\\` would");
        }
    }
    if (po->d_rows) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(po->d_rows)); po->d_rows = nullptr; }
    HIPCHK(ctx, hipMalloc((void **)&po->d_rows, sizeof(double) * rows.size()));
    HIPCHK(ctx, hipMemcpy(po->d_rows, rows.data(), sizeof(double) * rows.size(), hipMemcpyHostToDevice));
    po->rs = rs;
    {   // the Wiener increment scale INTO grid point j -- the very values of the rows (producer waves, bhip_pc_kernel.h)
        const size_t np = ((size_t)N + 15) / 16 * 16;
        std::vector<double> rdtp(np, 0.0);
        for (int j = 1; j < N; j++) rdtp[j] = rows[(size_t)(j - 1) * rs + 2];
        if (po->d_rdtp) { HIPCHK(ctx, hipFree(po->d_rdtp)); po->d_rdtp = nullptr; }
        HIPCHK(ctx, hipMalloc((void **)&po->d_rdtp, np * sizeof(double)));
        HIPCHK(ctx, hipMemcpy(po->d_rdtp, rdtp.data(), np * sizeof(double), hipMemcpyHostToDevice));
    }
    return BHIP_OK;
}

int bhip_proposal_guide_hv(bhip_proposal *po, const double *v, const double *hT)
{
    if (!po || !v) return BHIP_EINVAL;
    if (!po->has_aux) return fail(po->ctx, BHIP_ESTATE, "bhip_proposal_guide_hv: set the auxiliary process first");
    const int d = po->mh.d;
    Mat vv(d, 1, v), h(d, d);
    if (hT) h = Mat(d, d, hT);
    if (po->aux.kind == BHIP_AUX_LINEARAPPR) guide_hv_heuni(po->tt, po->aux, vv, h, po->g);   // src/guip.jl:181-189
    else guide_hv(po->tt, po->aux, vv, h, po->g);
    return finish_guide(po);
}

int bhip_proposal_guide_lmmu(bhip_proposal *po, int m, const double *L, const double *v, const double *Sigma)
{
    if (!po || !L || !v) return BHIP_EINVAL;
    if (!po->has_aux) return fail(po->ctx, BHIP_ESTATE, "bhip_proposal_guide_lmmu: set the auxiliary process first");
    const int d = po->mh.d;
    if (m < 1 || m > d) return fail(po->ctx, BHIP_EINVAL, "observation dimension m must be in 1..d");
    Mat S(m, m);
    if (Sigma) S = Mat(m, m, Sigma);   // default Sigma = outer(zero(v)) = 0  (src/partialbridge.jl:42)
    guide_lmmu(po->tt, po->aux, Mat(m, d, L), Mat(m, 1, v), S, po->g);
    return finish_guide(po);
}

int bhip_proposal_guide_nuh(bhip_proposal *po, int m, const double *L, const double *v, double eps, const double *Sigma, int inplace)
{
    if (!po || !L || !v) return BHIP_EINVAL;
    if (!po->has_aux) return fail(po->ctx, BHIP_ESTATE, "bhip_proposal_guide_nuh: set the auxiliary process first");
    const int d = po->mh.d;
    if (m < 1 || m > d) return fail(po->ctx, BHIP_EINVAL, "observation dimension m must be in 1..d");
    Mat S(m, m);
    if (Sigma) S = Mat(m, m, Sigma);
    if (inplace) guide_nuh_inplace(po->tt, po->aux, Mat(m, d, L), Mat(m, 1, v), eps, S, po->g);
    else guide_nuh(po->tt, po->aux, Mat(m, d, L), Mat(m, 1, v), eps, S, po->g);
    return finish_guide(po);
}

int bhip_proposal_guide_arrays(bhip_proposal *po, int kind, int m, const double *A1, const double *A2, const double *A3, const double *A4)
{
    if (!po) return BHIP_EINVAL;
    if (!po->has_aux) return fail(po->ctx, BHIP_ESTATE, "bhip_proposal_guide_arrays: set the auxiliary process first");
    const int N = (int)po->tt.size(), d = po->mh.d;
    Guide &g = po->g;
    g = Guide();
    g.kind = kind;
    if (kind == BHIP_GUIDE_HV) {
        if (!A1 || !A2) return fail(po->ctx, BHIP_EINVAL, "HV guide needs Hd and V");
        g.m = d; g.Hd.resize(N); g.V.resize(N);
        for (int i = 0; i < N; i++) { g.Hd[i] = Mat(d, d, A1 + (size_t)i * d * d); g.V[i] = Mat(d, 1, A2 + (size_t)i * d); }
    } else if (kind == BHIP_GUIDE_LMMU) {
        if (!A1 || !A2 || !A3 || !A4 || m < 1 || m > d) return fail(po->ctx, BHIP_EINVAL, "LMMU guide needs L, M, mu, v and 1 <= m <= d");
        g.m = m; g.L.resize(N); g.M.resize(N); g.mu.resize(N); g.v = Mat(m, 1, A4);
        for (int i = 0; i < N; i++) { g.L[i] = Mat(m, d, A1 + (size_t)i * m * d); g.M[i] = Mat(m, m, A2 + (size_t)i * m * m); g.mu[i] = Mat(m, 1, A3 + (size_t)i * m); }
    } else if (kind == BHIP_GUIDE_NUH || kind == BHIP_GUIDE_NUH_INPLACE) {
        if (!A1 || !A2) return fail(po->ctx, BHIP_EINVAL, "NUH guide needs nu and H");
        g.m = d; g.nu.resize(N); g.H.resize(N);
        for (int i = 0; i < N; i++) { g.nu[i] = Mat(d, 1, A1 + (size_t)i * d); g.H[i] = Mat(d, d, A2 + (size_t)i * d * d); }
    } else return fail(po->ctx, BHIP_EINVAL, "unknown guide kind");
    return finish_guide(po);
}

int bhip_proposal_guide_get(const bhip_proposal *po, double *A1, double *A2, double *A3, double *A4)
{
    if (!po) return BHIP_EINVAL;
    const Guide &g = po->g;
    const int N = (int)po->tt.size();
    auto put = [&](double *dst, const std::vector<Mat> &src) {
        if (!dst) return;
        size_t off = 0;
        for (int i = 0; i < N; i++) { std::memcpy(dst + off, src[i].a.data(), sizeof(double) * src[i].a.size()); off += src[i].a.size(); }
    };
    if (g.kind == BHIP_GUIDE_HV) { put(A1, g.Hd); put(A2, g.V); }
    else if (g.kind == BHIP_GUIDE_LMMU) { put(A1, g.L); put(A2, g.M); put(A3, g.mu); if (A4) std::memcpy(A4, g.v.a.data(), sizeof(double) * g.v.a.size()); }
    else if (g.kind == BHIP_GUIDE_NUH || g.kind == BHIP_GUIDE_NUH_INPLACE) { put(A1, g.nu); put(A2, g.H); if (A3) A3[0] = g.C; }
    else return BHIP_ESTATE;
    return BHIP_OK;
}

int bhip_proposal_lptilde(const bhip_proposal *po, const double *u, double *out)
{
    if (!po || !u || !out) return BHIP_EINVAL;
    const Guide &g = po->g;
    const int d = po->mh.d;
    Mat uu(d, 1, u);
    if (g.kind == BHIP_GUIDE_HV) {   // logpdfnormal(V[1]-u, Hd[1]) - traceB(tt, Pt)   src/guip.jl:206
        if (!g.have_trB) return fail(po->ctx, BHIP_EUNSUPPORTED, "lptilde: traceB is not defined for this auxiliary (LinearAppr: index-based coefficients)");
        *out = logpdfnormal(g.V[0] - uu, g.Hd[0]) - g.trB;
        return BHIP_OK;
    }
    if (g.kind == BHIP_GUIDE_NUH) {  // -0.5*(nu1-u)'H1(nu1-u) - C
        const Mat w = g.nu[0] - uu;
        *out = -0.5 * dot(w, g.H[0] * w) - g.C;
        return BHIP_OK;
    }
    return BHIP_EUNSUPPORTED;
}

int bhip_proposal_info(const bhip_proposal *po, int *N, int *d, int *mp, int *m, int *kind)
{
    if (!po) return BHIP_EINVAL;
    if (N) *N = (int)po->tt.size();
    if (d) *d = po->mh.d;
    if (mp) *mp = po->mh.mp;
    if (m) *m = po->g.m;
    if (kind) *kind = po->g.kind;
    return BHIP_OK;
}

/* ------------------------------------------------------------------ hot path */
static int ensure_plain_rows(bhip_proposal *po)
{   // forward EM without a guide: rows = (t, dt, sqrt(dt))
    NEED_DEVICE(po->ctx);
    if (po->d_rows || po->d_steps) return BHIP_OK;
    return finish_guide(po);
}

#ifdef PC_STAMP   /* measurement builds only: the waves' cycle stamps (bhip_pc_kernel.h), 12 words per wave */
constexpr size_t PC_STAMP_WORDS = (size_t)12 << 16;
static unsigned long long *pc_stamp_buffer()
{
    static unsigned long long *buf = nullptr;
    if (!buf && hipMalloc((void **)&buf, PC_STAMP_WORDS * 8) == hipSuccess) (void)hipMemset(buf, 0, PC_STAMP_WORDS * 8);
    return buf;
}
extern "C" int bhip_debug_stamps(unsigned long long *out, size_t words)
{
    unsigned long long *b = pc_stamp_buffer();
    if (!b || !out) return BHIP_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return BHIP_EHIP;
    if (hipMemcpy(out, b, std::min(words, PC_STAMP_WORDS) * 8, hipMemcpyDeviceToHost) != hipSuccess) return BHIP_EHIP;
    (void)hipMemset(b, 0, PC_STAMP_WORDS * 8);
    return BHIP_OK;
}
#endif
static int fill_common(const bhip_proposal *po, KArgs &a, const double *x0, const double *x0_dev, long npaths, int skip)
{
    bhip_ctx *ctx = po->ctx;
    const int d = po->mh.d;
    std::memset(&a, 0, sizeof(a));
    NEED_DEVICE(ctx);
    if (d > 3 && !po->mid) return fail(ctx, BHIP_EUNSUPPORTED, "path-per-lane kernel covers d <= 3 (LinPro targets and component-wise user drifts: d <= 8)");
    if (!po->d_rows) return fail(ctx, BHIP_ESTATE, "proposal has no coefficient rows (compute a guide first)");
    if (npaths < 1) return fail(ctx, BHIP_EINVAL, "npaths must be positive");
    if (skip < 0) return fail(ctx, BHIP_EINVAL, "skip must be >= 0");
    if (!x0 && !x0_dev) return fail(ctx, BHIP_EINVAL, "need a starting point");
    a.rows = po->d_rows; a.rs = po->rs; a.N = (int)po->tt.size(); a.skip = skip;
    a.rdtp = po->d_rdtp;
    a.P = npaths;
    a.wstride = 1;
    a.noise_spec = ctx->noise_spec;
    const bool aux_linpro = po->has_aux && po->aux.linpro_form();   // b~ = B(x - mu~); else b~ = B x + beta~ (mu~ = 0)
    a.use_vend = po->use_vend;
    for (int k = 0; k < d; k++) {
        a.x0[k] = x0 ? x0[k] : 0.0;
        a.vend[k] = po->vend[k];
        a.mu_aux[k] = aux_linpro ? po->aux.mu()[k] : 0.0;
    }
#ifdef PC_STAMP
    a.stamp = pc_stamp_buffer();
#endif
    if (po->mid) a.mpar_dev = po->d_mpar;   // the parameter block of a LinPro<4..8> target stays in device memory
    else {
        if ((int)po->mh.dpar.size() > 40) return fail(ctx, BHIP_EINVAL, "model parameter block too large");
        for (size_t k = 0; k < po->mh.dpar.size(); k++) a.mpar[k] = po->mh.dpar[k];
    }
    return BHIP_OK;
}

static int do_launch(const bhip_proposal *po, int noise, const KArgs &a)
{
    bhip_ctx *ctx = po->ctx;
    const int gk = po->g.kind == BHIP_GUIDE_NUH_INPLACE ? BHIP_GUIDE_NUH : po->g.kind;
    const int gk_dispatch = noise == NOISE_INNOV ? gk : po->g.kind;   // NUH_INPLACE selects the two-dot log-likelihood instantiation
    int fl = 0;
    if (noise == NOISE_PCN || noise == NOISE_PCN_LINES) fl = a.Xo ? 1 : 0;
    else fl = (a.X ? 1 : 0) | (a.Wout ? 2 : 0);
    // BHIP_OPT_NOISE_SPEC = 2: the register-tight kernels hold the default stream only (bhip_path_kernel.h k_paths, bhip_chain_kernel.h):
    // the pCN step on the 16-byte slots is refused (chains at d > 3 were created on the tile kernel under this specification,
    // bhip_chains_create; at d <= 3 the slots only serve grids too long for the line layout), the one-lane line kernel gives way to
    // the wave-specialised one
    const bool v2 = a.noise_spec == 2 || a.noise_spec == 3;   // (a non-default specification)
    if (v2 && noise == NOISE_PCN)
        return fail(ctx, BHIP_EUNSUPPORTED, "BHIP_OPT_NOISE_SPEC = 2 / 3: the pCN step on the 16-byte slots draws the default noise stream only");
    const bool wave_spec = ctx->wave_specialised || v2 || a.Xtb;   // (time-blocked path stores exist in the wave-specialised kernel only)
    if (a.Xtb && !(noise == NOISE_PCN_LINES && a.rdtp && po->mh.mp <= 3 && a.wstride == 1 && !po->mid)) return fail(ctx, BHIP_ESTATE, "time-blocked paths need the line layout");
    if (po->mid) {   // LinPro, d = 4..12: rows in the (nu, H) form, one kernel family
        // LinPro targets: the regrouped rows (GUIDE_QF), except for innovations!, which reads the (nu, H) rows kept beside them; component-wise
        // user drifts: the (nu, H) rows
        const bool user_mid = po->mh.id >= USER_MODEL_BASE;
        const int gkm = po->g.kind == BHIP_GUIDE_NONE ? BHIP_GUIDE_NONE : (user_mid || noise == NOISE_INNOV) ? BHIP_GUIDE_NUH : BHIP_GUIDE_QF;
        KArgs am = a;
        if (!user_mid && gkm == BHIP_GUIDE_NUH) {
            if (!po->d_rows_innov) return fail(ctx, BHIP_ESTATE, "proposal has no (nu, H) coefficient rows");
            am.rows = po->d_rows_innov; am.rs = po->rs_innov;
        }
        if (am.rs != row_stride(gkm, po->mh.d, 1, true)) return fail(ctx, BHIP_ESTATE, "row stride mismatch");
        if (po->mh.id >= USER_MODEL_BASE) {   // component-wise user drift: k_paths<MUser (streamed), gk, 1, noise, fl> through hipRTC
            if (!(noise == NOISE_EXT || noise == NOISE_FRESH || noise == NOISE_PCN || noise == NOISE_LLONLY || noise == NOISE_INNOV) ||
                (gkm == BHIP_GUIDE_NONE && (noise == NOISE_PCN || noise == NOISE_LLONLY)))
                return fail(ctx, BHIP_EUNSUPPORTED, "no path-per-lane kernel for this mode at 4 <= d <= 12");
            const int flk = noise == NOISE_INNOV ? 2 : fl;
            hipFunction_t fn = nullptr;
            {
                std::lock_guard<std::mutex> lk(user_models_mutex());
                UserModel *um = find_user_model(po->mh.id);
                if (!um) return fail(ctx, BHIP_EINVAL, "unknown user model id");
                const std::vector<int> key = {ctx->device, gkm, 1, noise, flk, -1};   // (-1: the streamed one-path-per-lane family)
                auto it = um->fns.find(key);
                if (it == um->fns.end()) {
                    const std::string log = rtc_build(*um, gkm, 1, noise, flk, &fn, 0);
                    if (!log.empty()) return fail(ctx, BHIP_EHIP, log);
                    um->fns[key] = fn;
                } else fn = it->second;
            }
            KArgs args = a;
            void *params[] = {&args};
            HIPCHK(ctx, hipModuleLaunchKernel(fn, (unsigned)((a.P + 255) / 256), 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr));
            return BHIP_OK;
        }
        launch_fn fm = nullptr;
        switch (po->mh.d) {
        case 4: fm = get_launch_mid4(gkm, noise, fl); break;
        case 5: fm = get_launch_mid5(gkm, noise, fl); break;
        case 6: fm = get_launch_mid6(gkm, noise, fl); break;
        case 7: fm = get_launch_mid7(gkm, noise, fl); break;
        case 8: fm = get_launch_mid8(gkm, noise, fl); break;
        case 9: fm = get_launch_mid9(gkm, noise, fl); break;
        case 10: fm = get_launch_mid10(gkm, noise, fl); break;
        case 11: fm = get_launch_mid11(gkm, noise, fl); break;
        case 12: fm = get_launch_mid12(gkm, noise, fl); break;
        }
        if (!fm) return fail(ctx, BHIP_EUNSUPPORTED, "no path-per-lane kernel for this mode at 4 <= d <= 12");
        HIPCHK(ctx, fm(am, ctx->stream));
        return BHIP_OK;
    }
    if (ctx->fused && po->d_rows_qf && a.rows == po->d_rows && noise != NOISE_INNOV && po->mh.id == BHIP_MODEL_LINPRO) {
        // BHIP_OPT_FUSED_ARITHMETIC, LinPro target at d <= 3: the regrouped step (bhip_path_kernel.h GUIDE_QF) on the rows finish_guide
        // keeps beside the reference-form ones -- one dependent fused multiply-add per component and step instead of the reference's chain
        // (tolerance parity, tests/test_gpu_fused.py); innovations! needs _b itself and stays on the reference form
        KArgs aq = a;
        aq.rows = po->d_rows_qf; aq.rs = po->rs_qf;
        launch_fn f = nullptr;
        if (wave_spec && a.rdtp && a.wstride == 1) {
            if (noise == NOISE_FRESH && a.P <= pc_fresh_max_paths()) f = find_launch(po->mh, BHIP_GUIDE_QF, 1, NOISE_FRESH_PC, fl, true);
            else if (noise == NOISE_PCN_LINES) f = find_launch(po->mh, BHIP_GUIDE_QF, 1, NOISE_PCN_LINES_PC, fl, true);
        }
        if (!f) f = find_launch(po->mh, BHIP_GUIDE_QF, 1, noise, fl, true);
        if (f) { HIPCHK(ctx, f(aq, ctx->stream)); return BHIP_OK; }
    }
    if (a.rs != row_stride(gk, po->mh.d, po->g.m, po->mh.constdiff)) return fail(ctx, BHIP_ESTATE, "row stride mismatch");
    if (!po->mh.constdiff) {
        // constdiff(P) == false.  The reference's extra log-likelihood terms exist for PartialBridge only
        // (src/partialbridge.jl:79-84); its other !constdiff branches reference undefined names.
        const bool wants_ll = a.ll != nullptr || noise == NOISE_PCN || noise == NOISE_PCN_LINES || noise == NOISE_LLONLY;
        if (wants_ll && gk != BHIP_GUIDE_LMMU)
            return fail(ctx, BHIP_EUNSUPPORTED, "llikelihood with a state-dependent sigma is defined for PartialBridge (L,M,mu) only");
        if (noise == NOISE_INNOV) return fail(ctx, BHIP_EUNSUPPORTED, "innovations need a constant, invertible sigma");
    }
    const int mo = gk == BHIP_GUIDE_LMMU ? po->g.m : 1;
    if (po->mh.id >= USER_MODEL_BASE) {   // hipRTC-compiled user drift: compile this instantiation on first use
        if (gk_dispatch == BHIP_GUIDE_NUH_INPLACE) fl |= 4;
        // the wave-specialised kernels where they apply, with the workgroup shapes of launch_pc (module kernels take up to the
        // full 160 KB of dynamic LDS on gfx950 without an opt-in: probed, 48 ... 160 KB)
        int npair = 0, knoise = noise;
        const long groups = (a.P + 63) / 64;
        if (wave_spec && a.rdtp && po->mh.mp <= 3 && a.wstride == 1) {
            if (noise == NOISE_FRESH && a.P <= pc_fresh_max_paths()) knoise = NOISE_FRESH_PC;
            else if (noise == NOISE_PCN_LINES) knoise = NOISE_PCN_LINES_PC;
            if (knoise != noise) {   // the workgroup shape launch_pc would choose
                bool rlds = true;
                npair = pc_choose_npair_rt(a, groups, po->mh.d, a.rs, po->mh.mp, &rlds, knoise);
                if (!rlds && npair > 1) npair = -npair;
            }
        }
        hipFunction_t fn = nullptr;
        {
            std::lock_guard<std::mutex> lk(user_models_mutex());
            UserModel *um = find_user_model(po->mh.id);
            if (!um) return fail(ctx, BHIP_EINVAL, "unknown user model id");
            const std::vector<int> key = {ctx->device, gk, mo, knoise, fl, npair};   // a hipFunction_t belongs to the device it was loaded on
            auto it = um->fns.find(key);
            if (it == um->fns.end()) {
                const std::string log = rtc_build(*um, gk, mo, knoise, fl, &fn, npair);
                if (!log.empty()) return fail(ctx, BHIP_EHIP, log);
                um->fns[key] = fn;
            } else fn = it->second;
        }
        KArgs args = a;
        void *params[] = {&args};
        if (npair != 0) {
            const int spc = LINE_DOUBLES / line_mpp(po->mh.mp), np = npair < 0 ? -npair : npair;
            const unsigned lds = (unsigned)(pc_lds_bytes(a.noise_spec, np, npair > 1 ? spc * a.rs : 0) + (a.Xtb ? pc_xs_bytes(po->mh.d, np) : 0));
            HIPCHK(ctx, hipModuleLaunchKernel(fn, (unsigned)((groups + np - 1) / np), 1, 1, (unsigned)pc_threads(knoise, npair > 1, np, po->mh.d), 1, 1, lds, ctx->stream, params, nullptr));
            return BHIP_OK;
        }
        const long grid = (a.P + 255) / 256;
        const unsigned lds = noise == NOISE_PCN_LINES ? (unsigned)CHAIN_LINES_LDS : 0u;
        HIPCHK(ctx, hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, lds, ctx->stream, params, nullptr));
        return BHIP_OK;
    }
    launch_fn f = nullptr;
    if (wave_spec && a.rdtp && po->mh.mp <= 3 && a.wstride == 1) {
        // producer/consumer waves (bhip_pc_kernel.h): same results, the kernel of choice wherever it is instantiated
        // Fresh proposals: with 4 waves per SIMD the one-lane-does-everything kernel already issues at ~85 % of the VALU
        // rate and the hand-over only costs; the split pays below that (profiles/r2_small_configs.txt).
        if (noise == NOISE_FRESH && a.P <= pc_fresh_max_paths()) f = find_launch(po->mh, gk_dispatch, mo, NOISE_FRESH_PC, fl, ctx->fused);
        else if (noise == NOISE_PCN_LINES) f = find_launch(po->mh, gk_dispatch, mo, NOISE_PCN_LINES_PC, fl, ctx->fused);
    }
    if (!f) f = find_launch(po->mh, gk_dispatch, mo, noise, fl, ctx->fused);
    if (!f) return fail(ctx, BHIP_EUNSUPPORTED, "no device kernel for this (model, guide, noise) combination");
    HIPCHK(ctx, f(a, ctx->stream));
    return BHIP_OK;
}

int bhip_wiener_sample(bhip_ctx *ctx, const double *tt, int N, int mp, double *W_dev, long ld, long npaths, uint64_t seed, uint32_t iter, uint32_t path0)
{
    if (!ctx || !tt || !W_dev || N < 2 || mp < 1 || npaths < 1 || ld < npaths) return fail(ctx, BHIP_EINVAL, "bhip_wiener_sample: bad argument");
    NEED_DEVICE(ctx);
    PATH_RANGE(ctx, path0, npaths);
    std::vector<double> rdt(N - 1);
    for (int i = 0; i + 1 < N; i++) rdt[i] = std::sqrt(tt[i + 1] - tt[i]);
    int rc = ensure_scratch(ctx, sizeof(double) * (size_t)(N - 1));
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->scratch, rdt.data(), sizeof(double) * (N - 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // rdt is a stack-lifetime host buffer
    const dim3 grid((unsigned)((npaths + 255) / 256)), block(256);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (mp == 1) hipLaunchKernelGGL(k_wiener<1>, grid, block, 0, ctx->stream, ctx->scratch, N, W_dev, ld, npaths, k0, k1, iter, path0, ctx->noise_spec);
    else if (mp == 2) hipLaunchKernelGGL(k_wiener<2>, grid, block, 0, ctx->stream, ctx->scratch, N, W_dev, ld, npaths, k0, k1, iter, path0, ctx->noise_spec);
    else if (mp == 3) hipLaunchKernelGGL(k_wiener<3>, grid, block, 0, ctx->stream, ctx->scratch, N, W_dev, ld, npaths, k0, k1, iter, path0, ctx->noise_spec);
    else if (mp == 4) hipLaunchKernelGGL(k_wiener<4>, grid, block, 0, ctx->stream, ctx->scratch, N, W_dev, ld, npaths, k0, k1, iter, path0, ctx->noise_spec);
    else
        hipLaunchKernelGGL(k_wiener_big, grid, block, 0, ctx->stream, ctx->scratch, N, mp, W_dev, ld, npaths, (uint32_t)seed, (uint32_t)(seed >> 32), iter, path0, ctx->noise_spec);
    HIPCHK(ctx, hipGetLastError());
    return BHIP_OK;
}

int bhip_solve(bhip_ctx *ctx, const bhip_proposal *po, const double *x0, const double *x0_dev, const double *W_dev, long ldW,
               double *X_dev, long ldX, double *ll_dev, int skip, long npaths)
{
    if (!ctx || !po) return BHIP_EINVAL;
    SAME_CTX(ctx, po);
    if (!W_dev) return fail(ctx, BHIP_EINVAL, "bhip_solve: W_dev is required");
    if (po->g.kind == BHIP_GUIDE_NONE) {
        if (ll_dev) return fail(ctx, BHIP_EINVAL, "bhip_solve: llikelihood needs a guided proposal");
        int rc = ensure_plain_rows(const_cast<bhip_proposal *>(po));
        if (rc) return rc;
    }
    if (ldW < npaths || (X_dev && ldX < npaths)) return fail(ctx, BHIP_ELENGTH, "leading dimension smaller than npaths");
    if (x0_dev && ldX < npaths) return fail(ctx, BHIP_ELENGTH, "per-path starting points x0_dev are laid out [d][ldX]: ldX must be >= npaths");
    if (po->mh.d > 3 && !(po->mid && po->mh.d <= ctx->mid_max))
        return launch_tile_path(po, x0, W_dev, ldW, nullptr, 0, X_dev, ldX, ll_dev, skip, npaths, 0, 0, 0, 0, 1, nullptr, 0.0, x0_dev, ldX);
    KArgs a;
    int rc = fill_common(po, a, x0, x0_dev, npaths, skip);
    if (rc) return rc;
    a.x0_dev = x0_dev; a.ldx0 = ldX;
    a.Win = W_dev; a.ldWin = ldW; a.X = X_dev; a.ldX = ldX; a.ll = ll_dev;
    return do_launch(po, NOISE_EXT, a);
}

// the geometry of an ensemble in parts as the kernels address it: 1..3 buffers, paths [j*part, (j+1)*part) in buffer j (part a multiple of 64)
static int check_parts(bhip_ctx *ctx, const char *who, int nparts, const void *const *ptrs, long ld, long part_paths, long npaths)
{
    if (!ptrs || nparts < 1 || nparts > 3) return fail(ctx, BHIP_EINVAL, std::string(who) + ": 1..3 parts");
    for (int j = 0; j < nparts; j++) if (!ptrs[j]) return fail(ctx, BHIP_EINVAL, std::string(who) + ": null part");
    if (nparts == 1) return ld < npaths ? fail(ctx, BHIP_ELENGTH, "leading dimension smaller than npaths") : BHIP_OK;
    if (part_paths < 64 || part_paths % 64 != 0) return fail(ctx, BHIP_EINVAL, std::string(who) + ": part_paths must be a positive multiple of 64");
    if (ld < part_paths) return fail(ctx, BHIP_ELENGTH, "leading dimension smaller than part_paths");
    if ((long)nparts * part_paths < npaths) return fail(ctx, BHIP_ELENGTH, std::string(who) + ": nparts * part_paths must cover npaths");
    return BHIP_OK;
}

// bhip_solve with the driving W and / or the paths X kept in parts (the containers of large ensembles): ONE launch reads and writes all
// of them -- two launches of half the paths each leave half of the waves per SIMD to hide the recurrence's latency
int bhip_solve_parts(bhip_ctx *ctx, const bhip_proposal *po, const double *x0, int nwparts, const double *const *W_parts, long ldW, long wpart_paths,
                     int nxparts, double *const *X_parts, long ldX, long xpart_paths, double *ll_dev, int skip, long npaths)
{
    if (!ctx || !po || !x0) return BHIP_EINVAL;
    SAME_CTX(ctx, po);
    int rc = check_parts(ctx, "bhip_solve_parts (W)", nwparts, (const void *const *)W_parts, ldW, wpart_paths, npaths);
    if (!rc && X_parts) rc = check_parts(ctx, "bhip_solve_parts (X)", nxparts, (const void *const *)X_parts, ldX, xpart_paths, npaths);
    if (rc) return rc;
    if (nwparts == 1 && (!X_parts || nxparts == 1))
        return bhip_solve(ctx, po, x0, nullptr, W_parts[0], ldW, X_parts ? X_parts[0] : nullptr, ldX, ll_dev, skip, npaths);
    if (po->g.kind == BHIP_GUIDE_NONE) {
        if (ll_dev) return fail(ctx, BHIP_EINVAL, "bhip_solve: llikelihood needs a guided proposal");
        rc = ensure_plain_rows(const_cast<bhip_proposal *>(po));
        if (rc) return rc;
    }
    if (po->mh.d > 3 && !(po->mid && po->mh.d <= ctx->mid_max))
        return fail(ctx, BHIP_EUNSUPPORTED, "bhip_solve_parts: one path per lane (d <= 3, LinPro / component-wise user drifts up to BHIP_OPT_MID_VALU); the tile kernel takes one buffer per launch");
    KArgs a;
    rc = fill_common(po, a, x0, nullptr, npaths, skip);
    if (rc) return rc;
    a.Win = W_parts[0]; a.ldWin = ldW;
    if (nwparts > 1) { a.Winp1 = W_parts[1]; a.Winp2 = nwparts > 2 ? W_parts[2] : nullptr; a.wpart = wpart_paths; }
    if (X_parts) {
        a.X = X_parts[0]; a.ldX = ldX;
        if (nxparts > 1) { a.Xp1 = X_parts[1]; a.Xp2 = nxparts > 2 ? X_parts[2] : nullptr; a.xpart = xpart_paths; }
    }
    a.ll = ll_dev;
    return do_launch(po, NOISE_EXT, a);
}

// bhip_llikelihood of an ensemble kept in parts, by ONE launch
int bhip_llikelihood_parts(bhip_ctx *ctx, const bhip_proposal *po, int nparts, const double *const *X_parts, long ldX, long part_paths,
                           double *ll_dev, int skip, long npaths)
{
    if (!ctx || !po || !ll_dev) return BHIP_EINVAL;
    SAME_CTX(ctx, po);
    int rc = check_parts(ctx, "bhip_llikelihood_parts", nparts, (const void *const *)X_parts, ldX, part_paths, npaths);
    if (rc) return rc;
    if (nparts == 1) return bhip_llikelihood(ctx, po, X_parts[0], ldX, ll_dev, skip, npaths);
    if (po->g.kind == BHIP_GUIDE_NONE) return fail(ctx, BHIP_EINVAL, "bhip_llikelihood: needs a guided proposal");
    if (po->mh.d > 3 && !(po->mid && po->mh.d <= ctx->mid_max))
        return fail(ctx, BHIP_EUNSUPPORTED, "bhip_llikelihood_parts: one path per lane (d <= 3, LinPro / component-wise user drifts up to BHIP_OPT_MID_VALU); the tile kernel takes one buffer per launch");
    KArgs a;
    const double zero[BHIP_MAXD_LANE] = {0};
    rc = fill_common(po, a, zero, nullptr, npaths, skip);
    if (rc) return rc;
    a.Win = X_parts[0]; a.ldWin = ldX; a.Winp1 = X_parts[1]; a.Winp2 = nparts > 2 ? X_parts[2] : nullptr; a.wpart = part_paths; a.ll = ll_dev;
    return do_launch(po, NOISE_LLONLY, a);
}

int bhip_sample_solve(bhip_ctx *ctx, const bhip_proposal *po, const double *x0, const double *x0_dev, double *W_dev, long ldW,
                      double *X_dev, long ldX, double *ll_dev, int skip, long npaths, uint64_t seed, uint32_t iter, uint32_t path0)
{
    if (!ctx || !po) return BHIP_EINVAL;
    SAME_CTX(ctx, po);
    if (po->g.kind == BHIP_GUIDE_NONE) {
        if (ll_dev) return fail(ctx, BHIP_EINVAL, "bhip_sample_solve: llikelihood needs a guided proposal");
        int rc = ensure_plain_rows(const_cast<bhip_proposal *>(po));
        if (rc) return rc;
    }
    if ((W_dev && ldW < npaths) || (X_dev && ldX < npaths)) return fail(ctx, BHIP_ELENGTH, "leading dimension smaller than npaths");
    if (x0_dev && ldX < npaths) return fail(ctx, BHIP_ELENGTH, "per-path starting points x0_dev are laid out [d][ldX]: ldX must be >= npaths");
    PATH_RANGE(ctx, path0, npaths > 0 ? npaths : 0);
    if (po->mh.d > 3 && !(po->mid && po->mh.d <= ctx->mid_max))
        return launch_tile_path(po, x0, nullptr, 0, W_dev, ldW, X_dev, ldX, ll_dev, skip, npaths, 1, seed, iter, path0, 1, nullptr, 0.0, x0_dev, ldX);
    KArgs a;
    int rc = fill_common(po, a, x0, x0_dev, npaths, skip);
    if (rc) return rc;
    a.x0_dev = x0_dev; a.ldx0 = ldX;
    a.Wout = W_dev; a.ldWout = ldW; a.X = X_dev; a.ldX = ldX; a.ll = ll_dev;
    a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.iter = iter; a.path0 = path0;
    return do_launch(po, NOISE_FRESH, a);
}

// bhip_sample_solve with X kept in nparts buffers: paths [j*part_paths, (j+1)*part_paths) are written to X_parts[j] ([N][d][ldX] each,
// column p - j*part_paths) by ONE launch, so that the parts' write streams run side by side.  Values are those of bhip_sample_solve.
int bhip_sample_solve_parts(bhip_ctx *ctx, const bhip_proposal *po, const double *x0, int nparts, double *const *X_parts, long ldX, long part_paths,
                            double *ll_dev, int skip, long npaths, uint64_t seed, uint32_t iter, uint32_t path0)
{
    if (!ctx || !po || !X_parts || nparts < 1 || nparts > 3) return BHIP_EINVAL;
    SAME_CTX(ctx, po);
    for (int j = 0; j < nparts; j++) if (!X_parts[j]) return BHIP_EINVAL;
    if (nparts == 1) return bhip_sample_solve(ctx, po, x0, nullptr, nullptr, 0, X_parts[0], ldX, ll_dev, skip, npaths, seed, iter, path0);
    if (part_paths < 64 || part_paths % 64 != 0) return fail(ctx, BHIP_EINVAL, "bhip_sample_solve_parts: part_paths must be a positive multiple of 64");
    if (ldX < part_paths) return fail(ctx, BHIP_ELENGTH, "leading dimension smaller than part_paths");
    if ((long)nparts * part_paths < npaths) return fail(ctx, BHIP_ELENGTH, "bhip_sample_solve_parts: nparts * part_paths must cover npaths");
    if (po->g.kind == BHIP_GUIDE_NONE) {
        if (ll_dev) return fail(ctx, BHIP_EINVAL, "bhip_sample_solve: llikelihood needs a guided proposal");
        int rc = ensure_plain_rows(const_cast<bhip_proposal *>(po));
        if (rc) return rc;
    }
    PATH_RANGE(ctx, path0, npaths > 0 ? npaths : 0);
    if (po->mh.d > 3 && !(po->mid && po->mh.d <= ctx->mid_max))
        return fail(ctx, BHIP_EUNSUPPORTED, "bhip_sample_solve_parts: one path per lane (d <= 3, LinPro / component-wise user drifts up to BHIP_OPT_MID_VALU); the tile kernel writes one buffer");
    KArgs a;
    int rc = fill_common(po, a, x0, nullptr, npaths, skip);
    if (rc) return rc;
    a.X = X_parts[0]; a.Xp1 = X_parts[1]; a.Xp2 = nparts > 2 ? X_parts[2] : nullptr; a.xpart = part_paths; a.ldX = ldX; a.ll = ll_dev;
    a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.iter = iter; a.path0 = path0;
    return do_launch(po, NOISE_FRESH, a);
}

int bhip_llikelihood(bhip_ctx *ctx, const bhip_proposal *po, const double *X_dev, long ldX, double *ll_dev, int skip, long npaths)
{
    if (!ctx || !po || !X_dev || !ll_dev) return BHIP_EINVAL;
    SAME_CTX(ctx, po);
    if (po->g.kind == BHIP_GUIDE_NONE) return fail(ctx, BHIP_EINVAL, "bhip_llikelihood: needs a guided proposal");
    if (po->mh.d > 3 && !(po->mid && po->mh.d <= ctx->mid_max)) {
        if (ldX < npaths) return fail(ctx, BHIP_ELENGTH, "leading dimension smaller than npaths");
        const std::vector<double> zero(po->mh.d, 0.0);
        return launch_tile_path(po, zero.data(), X_dev, ldX, nullptr, 0, nullptr, 0, ll_dev, skip, npaths, 3, 0, 0, 0);
    }
    KArgs a;
    const double zero[BHIP_MAXD_LANE] = {0};
    int rc = fill_common(po, a, zero, nullptr, npaths, skip);
    if (rc) return rc;
    if (ldX < npaths) return fail(ctx, BHIP_ELENGTH, "leading dimension smaller than npaths");
    a.Win = X_dev; a.ldWin = ldX; a.ll = ll_dev;
    return do_launch(po, NOISE_LLONLY, a);
}

int bhip_innovations(bhip_ctx *ctx, const bhip_proposal *po, const double *X_dev, long ldX, double *W_dev, long ldW, long npaths)
{
    if (!ctx || !po || !X_dev || !W_dev) return BHIP_EINVAL;
    SAME_CTX(ctx, po);
    if (po->mh.d != po->mh.mp) return fail(ctx, BHIP_EINVAL, "bhip_innovations: needs a square, invertible sigma (d == m')");
    if (po->mh.d > 3 && !(po->mid && po->mh.d <= ctx->mid_max))
        return fail(ctx, BHIP_EUNSUPPORTED, "innovations: d <= 3, or a LinPro target of a dimension that runs one path per lane (4..10 by default, up to 12 with BHIP_OPT_MID_VALU)");
    if (po->g.kind == BHIP_GUIDE_NONE) {
        int rc = ensure_plain_rows(const_cast<bhip_proposal *>(po));
        if (rc) return rc;
    }
    KArgs a;
    const double zero[BHIP_MAXD_LANE] = {0};
    int rc = fill_common(po, a, zero, nullptr, npaths, 0);
    if (rc) return rc;
    if (ldX < npaths || ldW < npaths) return fail(ctx, BHIP_ELENGTH, "leading dimension smaller than npaths");
    a.Win = X_dev; a.ldWin = ldX; a.Wout = W_dev; a.ldWout = ldW;
    return do_launch(po, NOISE_INNOV, a);
}

int bhip_girsanov(bhip_ctx *ctx, const bhip_proposal *po, const double *par_t, int npar_t, const double *X_dev, long ldX,
                  double *out_dev, long npaths)
{
    if (!ctx || !po || !X_dev || !out_dev) return BHIP_EINVAL;
    SAME_CTX(ctx, po);
    NEED_DEVICE(ctx);
    const ModelHost &mh = po->mh;
    girsanov_fn f = nullptr;
    switch (mh.id) {
    case BHIP_MODEL_OU: f = launch_girsanov<MOU>; break;
    case BHIP_MODEL_LINPRO: f = mh.d == 1 ? launch_girsanov<MLinPro<1>> : mh.d == 2 ? launch_girsanov<MLinPro<2>> : mh.d == 3 ? launch_girsanov<MLinPro<3>> : nullptr; break;
    case BHIP_MODEL_LORENZ: f = launch_girsanov<MLorenz>; break;
    case BHIP_MODEL_FHN2: f = launch_girsanov<MFHN2>; break;
    default: break;
    }
    if (!f) return fail(ctx, BHIP_EUNSUPPORTED, "bhip_girsanov: needs a built-in target with invertible a (OU, LinPro d<=3, Lorenz, Models.FitzHughNagumo)");
    if (npaths < 1) return fail(ctx, BHIP_EINVAL, "npaths must be positive");
    if (ldX < npaths) return fail(ctx, BHIP_ELENGTH, "leading dimension smaller than npaths");
    if (po->g.kind == BHIP_GUIDE_NONE) {
        int rc = ensure_plain_rows(const_cast<bhip_proposal *>(po));
        if (rc) return rc;
    }
    if (!po->d_rows) return fail(ctx, BHIP_ESTATE, "proposal has no coefficient rows (compute a guide first)");
    GirsArgs a;
    std::memset(&a, 0, sizeof(a));
    a.rows = po->d_rows; a.rs = po->rs; a.N = (int)po->tt.size(); a.P = npaths;
    a.X = X_dev; a.ldX = ldX; a.out = out_dev;
    if (mh.dpar.size() > 40) return fail(ctx, BHIP_EUNSUPPORTED, "parameter block too large");
    std::copy(mh.dpar.begin(), mh.dpar.end(), a.mpar);
    if (par_t) {
        ModelHost mt;
        std::string err;
        int rc = model_setup(mh.id, mh.d, par_t, npar_t, mt, err);
        if (rc) return fail(ctx, rc, "bhip_girsanov: Pt: " + err);
        std::copy(mt.dpar.begin(), mt.dpar.end(), a.mpar_t);
    } else {
        a.zero_t = 1;
        std::copy(mh.dpar.begin(), mh.dpar.end(), a.mpar_t);
    }
    // Gamma(t,x,P) = inv(a(t,x,P)), src/types.jl:33; sigma::SDiagonal (src/Models.jl:19,57) keeps a
    // diagonal, whose inverse is taken entry by entry
    if (mh.id == BHIP_MODEL_LORENZ || mh.id == BHIP_MODEL_FHN2) {
        for (int k = 0; k < mh.d; k++) a.gam[k + mh.d * k] = 1.0 / mh.a(k, k);
    } else {
        const Mat G = inv(mh.a);
        std::copy(G.a.begin(), G.a.end(), a.gam);
    }
    HIPCHK(ctx, f(a, ctx->stream));
    return BHIP_OK;
}

int bhip_gpupdate(int d, int m, const double *Hd, const double *V, const double *L, const double *Sigma, const double *v,
                  double *Hd_out, double *V_out)
{
    if (d < 1 || m < 1 || !Hd || !V || !L || !Sigma || !v || !Hd_out || !V_out) return BHIP_EINVAL;
    Mat Ho, Vo;
    gpupdate(Mat(d, d, Hd), Mat(d, 1, V), Mat(m, d, L), Mat(m, m, Sigma), Mat(m, 1, v), Ho, Vo);
    std::memcpy(Hd_out, Ho.a.data(), sizeof(double) * d * d);
    std::memcpy(V_out, Vo.a.data(), sizeof(double) * d);
    return BHIP_OK;
}

/* ------------------------------------------------------------------ chains */
// Where a chain ensemble's memory lies.  On MI355X the pCN iteration -- three streams: read W, write Wo, write Xo -- runs at 1.53 ms
// (bench workload) when the chain lines and the proposal paths lie in DIFFERENT 96-GiB pieces of the device's physical memory and at
// 1.78 ms when they share one: each piece (the top level of the physical address map: 288 GiB = 3 x 96) has its own DRAM banks, and
// three streams inside one piece close each other's rows (profiles/r4_placement_regions.txt: counters of slow and fast allocations,
// sweeps inside one contiguous 200-GiB block).  A plain hipMalloc of W + Xo is assembled from the allocator's free blocks -- a
// mixture of pieces, hence round 3's "lottery".  HIP neither reports nor accepts physical addresses, so the policy is:
//   * ensembles below 1 GiB, or without proposal paths: ONE allocation (Xo behind W, 2 MiB aligned), as before;
//   * larger ones: W and Xo are two allocations, each physically contiguous (hipExtMallocWithFlags(hipDeviceMallocContiguous): one run
//     of addresses lies in one piece unless it straddles a cut), and bhip_chains_init makes sure they are in different pieces by
//     measuring (chains_place below).
static hipError_t alloc_run(void **q, size_t bytes)
{
    hipError_t e = hipExtMallocWithFlags(q, bytes, hipDeviceMallocContiguous);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(q, bytes); }   // no contiguous run of that size left: what the allocator has
    return e;
}
#ifdef BHIP_PLACE_EXPERIMENTS
#include "../../scripts/bhip_place_experiments.inc"   /* measurement builds only (EXTRA=-DBHIP_PLACE_EXPERIMENTS): the BHIP_PLACE hooks behind profiles/r4_placement_*.txt */
#endif
static void arena_free(Arena &ar)
{
    if (ar.base2) (void)hipFree(ar.base2);
    if (ar.base && ar.owned) {
#ifdef BHIP_PLACE_EXPERIMENTS
        if (ar.vmm) place_exp_free_vmm(ar); else
#endif
        (void)hipFree(ar.base);
    }
    ar = Arena();
}
constexpr size_t PLACE_SPLIT_BYTES = (size_t)1 << 30;   // from here on W and Xo are separate contiguous allocations and get placed (measured: the
                                                        // 32 768-chain shard, 1.05 GB, runs at 0.203 ms placed or not -- latency, not memory; 65 536 chains, 2.1 GB: 0.367 vs 0.424)
static bool chains_split_state(const bhip_chains *ch)
{
    // (the d > 3 tile kernel's chains are not: its own read and write streams are the two halves of the 34-GB tile-line array, which a
    // plain allocation already spreads over the pieces -- contiguous runs made them 5 % SLOWER, 17.36 vs 16.43 ms, whatever Xo did)
    return (ch->flags & BHIP_CHAINS_STORE_X) != 0 && ch->wbytes + ch->xbytes >= PLACE_SPLIT_BYTES && ch->lines;
}
static hipError_t chains_alloc_state(const bhip_chains *ch, Arena &ar, double **Wc, double **Xo)
{
    const size_t MB2 = (size_t)2 << 20;
    const bool want_x = (ch->flags & BHIP_CHAINS_STORE_X) != 0;
    ar = Arena();
    *Wc = nullptr; *Xo = nullptr;
#ifdef BHIP_PLACE_EXPERIMENTS
    { hipError_t ee; if (place_exp_alloc(ch, ar, Wc, Xo, &ee)) return ee; }
#endif
    if (chains_split_state(ch)) {
        hipError_t e = alloc_run(&ar.base, ch->wbytes);
        if (e == hipSuccess) e = alloc_run(&ar.base2, ch->xbytes);
        if (e != hipSuccess) { arena_free(ar); return e; }
        ar.bytes = ch->wbytes;
        *Wc = (double *)ar.base; *Xo = (double *)ar.base2;
        return hipSuccess;
    }
    const size_t wspan = (ch->wbytes + MB2 - 1) / MB2 * MB2;
    const hipError_t e = hipMalloc(&ar.base, want_x ? wspan + ch->xbytes : ch->wbytes);
    if (e != hipSuccess) return e;
    ar.bytes = want_x ? wspan + ch->xbytes : ch->wbytes;
    *Wc = (double *)ar.base;
    if (want_x) *Xo = (double *)((char *)ar.base + wspan);
    return e;
}

int bhip_chains_create(bhip_ctx *ctx, const bhip_proposal *po, long nchains, uint32_t path0, uint64_t seed, int flags, bhip_chains **out)
{
    if (!ctx || !po || !out) return BHIP_EINVAL;
    SAME_CTX(ctx, po);
    *out = nullptr;
    NEED_DEVICE(ctx);
    if (nchains < 1) return fail(ctx, BHIP_EINVAL, "nchains must be positive");
    PATH_RANGE(ctx, path0, nchains);
    if (po->g.kind == BHIP_GUIDE_NONE) return fail(ctx, BHIP_EINVAL, "chains need a guided proposal");
    // d > 3: one path per lane (slots) up to the chains' cut -- lower than the proposals' (the slots' traffic and registers: 8.5 vs 5.6 ms at
    // d = 9) --, and never under the full-resolution noise specification (the slot kernel draws the default stream only)
    const bool on_tile = po->mh.d > 3 && !(po->mid && po->mh.d <= std::min(ctx->mid_max, (int)BHIP_MID_MAX_CHAINS) && ctx->noise_spec == 4);
    if (on_tile && !po->d_steps)
        return fail(ctx, BHIP_ESTATE, "chains: the proposal has no large-d guide data (compute a guide first)");
    bhip_chains *ch = new (std::nothrow) bhip_chains();
    if (!ch) return fail(ctx, BHIP_EHIP, "out of host memory");
    ch->ctx = ctx; ch->po = po; ch->n = nchains; ch->ld = (nchains + 63) / 64 * 64;
#ifdef BHIP_PLACE_EXPERIMENTS
    { const char *e = getenv("BHIP_LD_PAD"); if (e) ch->ld += atol(e) / 64 * 64; }   // (leading dimension off the power of two)
#endif
    ctx_retain(ctx);
    ch->path0 = path0; ch->seed = seed; ch->flags = flags; ch->noise_spec = ctx->noise_spec;
    const size_t N = po->tt.size();
    ch->tile = on_tile;
    ch->lines = po->mh.d <= 3 && po->mh.mp <= 3;
    const size_t spc = LINE_DOUBLES / (ch->lines ? line_mpp(po->mh.mp) : 1);   // grid points per line (m' = 3: padded to 4 components)
    ch->nch = (int)((N + spc - 1) / spc);
    if (ch->nch > 65535) {   // the layout-conversion kernels index the chunks with gridDim.y: very long grids stay on the slots
        ch->lines = false;
        ch->nch = 0;
    }
    const size_t wbytes = ch->lines ? sizeof(double) * 2 * ch->nch * ch->ld * LINE_DOUBLES
                        : ch->tile ? sizeof(double) * 2 * N * (tile_dim(po->mh.d) / 16) * ch->ld * 16   // tile lines (bhip_tile_kernel.h)
                                       : sizeof(double) * 2 * N * po->mh.mp * ch->ld;
    const size_t xbytes = sizeof(double) * N * po->mh.d * ch->ld;
    ch->wbytes = wbytes; ch->xbytes = xbytes;
    hipError_t e = chains_alloc_state(ch, ch->arena, &ch->Wc, &ch->Xo);
    if (e == hipSuccess) e = hipMalloc((void **)&ch->cur, ch->ld);
    if (e == hipSuccess) e = hipMalloc((void **)&ch->llcur, sizeof(double) * ch->ld);
    if (e == hipSuccess) e = hipMalloc((void **)&ch->acc, sizeof(unsigned int) * ch->ld);
    if (e == hipSuccess) e = hipMalloc((void **)&ch->statpart, sizeof(double) * 256 * 6);
    if (e != hipSuccess) { bhip_chains_destroy(ch); return fail(ctx, BHIP_EHIP, std::string("chains allocation: ") + hipGetErrorString(e)); }
    *out = ch;
    return BHIP_OK;
}

void bhip_chains_destroy(bhip_chains *ch)
{
    if (!ch) return;
    bhip_ctx *ctx = ch->ctx;
    ctx_quiesce(ctx);
    for (size_t k = ctx->pieces.size(); k-- > 0;)   // the piece map forgets the buffers that go away
        if (ctx->pieces[k].p == ch->arena.base || ctx->pieces[k].p == ch->arena.base2) ctx->pieces.erase(ctx->pieces.begin() + (long)k);
    arena_free(ch->arena);
    if (ch->cur && !ch->shares_state) (void)hipFree(ch->cur);
    if (ch->llcur && !ch->shares_state) (void)hipFree(ch->llcur);
    if (ch->acc && !ch->shares_state) (void)hipFree(ch->acc);
    if (ch->statpart) (void)hipFree(ch->statpart);
    delete ch;
    ctx_release(ctx);
}

// x0_dev (optional): per-chain starting points [d][ldx0] (multi-segment ensembles: the end points of the previous segment);
// blk0: offset of the Philox block index (segment << 24)
static int chains_init_impl(bhip_chains *ch, const double *x0, const double *x0_dev, long ldx0, int skip, uint32_t blk0)
{
    bhip_ctx *ctx = ch->ctx;
    const bhip_proposal *po = ch->po;
    NEED_DEVICE(ctx);
    if (skip < 0) return fail(ctx, BHIP_EINVAL, "skip must be >= 0");
    if (ch->noise_spec != ctx->noise_spec) return fail(ctx, BHIP_ESTATE, "the context's noise specification changed after the ensemble was created");
    ch->x0.assign(x0, x0 + po->mh.d);
    if (!ch->shares_state) {
        HIPCHK(ctx, hipMemsetAsync(ch->cur, 0, ch->ld, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ch->acc, 0, sizeof(unsigned int) * ch->ld, ctx->stream));
    }
    if (ch->tile) {   // MFMA tile kernel: fresh W into a plain SoA scratch array, re-arranged into half 0 of the tile lines; X and ll of the initial state
        const int N = (int)po->tt.size(), d = po->mh.d, T = tile_dim(d) / 16;
        double *tmpW = nullptr;
        HIPCHK(ctx, hipMalloc((void **)&tmpW, sizeof(double) * N * d * ch->n));
        int rct = launch_tile_path(po, x0, nullptr, 0, tmpW, ch->n, ch->Xo, ch->ld, ch->llcur, skip, ch->n, 1, ch->seed, 0, ch->path0, 1, nullptr, 0.0, x0_dev, ldx0, blk0);
        if (!rct) {
            const long tot = (long)N * 16 * T * ch->n;
            hipLaunchKernelGGL(k_soa_to_tlines, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, tmpW, ch->Wc, N, d, T, ch->ld, ch->n);
            if (hipGetLastError() != hipSuccess) rct = fail(ctx, BHIP_EHIP, "k_soa_to_tlines launch failed");
        }
        rct = sync_free_rc(ctx, tmpW, rct);
        if (rct) return rct;
        ch->skip0 = skip; ch->iter = 0; ch->inited = true;
        return BHIP_OK;
    }
    KArgs a;
    int rc = fill_common(po, a, x0, x0_dev, ch->n, skip);
    if (rc) return rc;
    a.x0_dev = x0_dev; a.ldx0 = ldx0; a.blk0 = blk0;
    double *tmpW = nullptr;
    if (ch->lines) {   // the fresh W goes to a plain SoA scratch array and is re-arranged into half 0 of the lines
        HIPCHK(ctx, hipMalloc((void **)&tmpW, sizeof(double) * po->tt.size() * po->mh.mp * ch->ld));
        a.Wout = tmpW; a.ldWout = ch->ld; a.wstride = 1;
    } else {
        a.Wout = ch->Wc; a.ldWout = ch->ld; a.wstride = 2;   // half 0 of every slot, cur = 0
    }
    a.X = ch->Xo; a.ldX = ch->ld; a.ll = ch->llcur;
    ch->skip0 = skip;
    a.k0 = (uint32_t)ch->seed; a.k1 = (uint32_t)(ch->seed >> 32); a.iter = 0; a.path0 = ch->path0;
    rc = do_launch(po, NOISE_FRESH, a);
    if (!rc && ch->lines) {
        hipLaunchKernelGGL(k_soa_to_lines, dim3((unsigned)(ch->ld / 64), (unsigned)ch->nch), dim3(256), 0, ctx->stream, tmpW, ch->ld, (int)po->tt.size(), po->mh.mp, ch->nch,
                           ch->Wc, ch->ld, ch->n);
        if (hipGetLastError() != hipSuccess) rc = fail(ctx, BHIP_EHIP, "k_soa_to_lines launch failed");
    }
    if (tmpW) rc = sync_free_rc(ctx, tmpW, rc);
    if (rc) return rc;
    ch->iter = 0; ch->inited = true;
    return BHIP_OK;
}

extern "C" int bhip_chains_step(bhip_chains *ch, double rho, int iters, int skip);
// ms per pCN iteration on the ensemble's present allocations (one untimed iteration, then `reps`)
// Placement of a large ensemble (BHIP_OPT_TUNE_PLACEMENT, default on): W and Xo -- two physically contiguous allocations,
// chains_alloc_state -- have to lie in DIFFERENT 96-GiB pieces of the device memory (see above; profiles/r4_placement_regions.txt).
// Nothing reports where an allocation lies, but two plain write streams tell: into two buffers of one piece they run at ~4.4 TB/s,
// into buffers of different pieces at ~5.3-6.5.  Round 4 searched per ensemble with the ensemble's own kernel (a same-piece reference
// block of |W| + |Xo|, kernel-timed pairs, up to 16 candidates held: 23-60 ms and up to 67 GiB of transient memory per ensemble).
// Since round 5 the CONTEXT keeps a piece map -- the large buffers of its live ensembles with the piece each was found in -- and a new
// buffer is classified against one representative per piece:
//   1. r_same, the two-stream rate inside one piece, is measured once per context: the MEDIAN of six runs inside W and inside Xo
//      (head x tail, head x middle, middle x tail of each; runs too short for three regions: the mean of the two head x tail runs);
//   2. the pair: head and tail of W against head and tail of Xo, four two-stream runs judged TOGETHER (place_apart): done (tries = 1).
//      Otherwise further candidates for Xo are allocated while the earlier ones stay held (the allocator changes pieces every 4 to 8
//      blocks of this size at the latest: profiles/r4_alloc_sequence_raw.txt) until one passes;
//   3. when every candidate fails -- the allocator is still inside W's piece --, a spacer allocation and another W beyond it (twice at
//      most, four more candidates each time), the held candidates judged again; everything not kept is freed at once;
//   4. W and Xo of a good pair are classified against the map and entered.  No kernel-timed run, no reference block.
// Results are those of an ensemble placed anywhere (the state of iteration 0 is set up afresh at the end; tests/test_gpu_pc.py).
// Every threshold of the procedure:
struct PlaceParams {
    size_t stream_bytes = (size_t)512 << 20;   // bytes per write stream of a test: beyond the 256-MB Infinity Cache
    // ONE two-stream run does not separate the classes: the rate also depends on where in their runs the two regions lie (0.77-1.27 x
    // r_same for regions of one piece, 1.07-1.66 across pieces).  The MEAN of the four runs of a pair does.  Ground truth: 81 pairs at the
    // headline size, five processes, each with its pCN iteration timed as well (BHIP_PLACE_TRACE; profiles/r5_piece_map.txt) -- the 47
    // slow pairs (1.61-1.68 ms) have mean-of-four 0.92-1.107 x r_same, the 34 fast ones (1.39-1.45 under that short warm-up)
    // 1.171-1.39.  Against the SMALLER of two head x tail runs as r_same (the round's first version) the classes touched (slow up to
    // 1.19, fast from 1.18): one low run moved every ratio of the process.  The smallest of the four overlaps (fast pairs from 1.07,
    // slow ones up to 1.08): it only guards against a run astride a cut (two of its four corners in W's piece: ~1.0).
    float mean_min = 1.14f;        // mean of the four >= mean_min * r_same ...
    float smallest_min = 1.03f;    // ... and every one of them >= smallest_min * r_same: the two buffers lie in different pieces
    float same_mean_max = 1.11f;   // mean of the four <= this x r_same: they share a piece (the map's labels); between: not attributed
    int max_candidates = 8;     // Xo candidates held at once at most (never more than 8 consecutive 4-GiB blocks of one piece were seen) ...
    int more_candidates = 4;    // ... and this many more after each of the two W re-rolls
    size_t spacer_bytes = (size_t)24 << 30;   // held (unwritten) before each W re-roll when the free memory allows: 8 x 4 + 24 + 4 x 4 + 24 + ... GiB walk past a 96-GiB piece
    size_t held_bytes = (size_t)24 << 30;   // ... or, for smaller buffers (the allocator's runs inside one piece are longer in blocks), as many as fit here, 24 at most
    size_t min_bytes = (size_t)64 << 20;   // buffers below this are not classified (a test needs streams of some length)
};
static const PlaceParams PLACE;

// GB/s of two write streams of `bytes` each into a and b on the context's stream (k_two_write_streams); 0 on any error
static float two_stream_rate(bhip_ctx *ctx, void *a, void *b, size_t bytes)
{
    const size_t m = bytes / 16;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!m || hipEventCreate(&e0) != hipSuccess) return 0.f;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return 0.f; }
    float ms = 0.f;
    hipLaunchKernelGGL(k_two_write_streams, dim3(4096), dim3(256), 0, ctx->stream, (d2v *)a, (d2v *)b, m);   // untimed
    hipError_t e = hipEventRecord(e0, ctx->stream);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_two_write_streams, dim3(4096), dim3(256), 0, ctx->stream, (d2v *)a, (d2v *)b, m);
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess || !(ms > 0.f)) { (void)hipGetLastError(); return 0.f; }
    return (float)(2.0 * 2.0 * (double)bytes / (ms * 1e6));
}
static size_t place_stream_bytes(size_t a_bytes, size_t b_bytes)
{
    return std::min<size_t>({a_bytes, b_bytes, PLACE.stream_bytes}) / 4096 * 4096;
}
// The four two-stream runs of a pair of buffers -- head and tail of a against head and tail of b -- : the smallest rate and the mean.
// With stop_below > 0 the runs end at the first rate below it (the pair has failed by then; *mean is that of the runs made).
struct PairRates { float smallest, mean; int runs; };
static PairRates place_pair_rates(bhip_ctx *ctx, void *a, size_t abytes, void *b, size_t bbytes, float stop_below, bool trace)
{
    const size_t sp = place_stream_bytes(abytes / 2, bbytes / 2);   // (the size r_same was measured with: two disjoint regions of one run)
    char *ae[2] = {(char *)a, (char *)a + (abytes - sp) / 4096 * 4096}, *be[2] = {(char *)b, (char *)b + (bbytes - sp) / 4096 * 4096};
    PairRates pr{1e30f, 0.f, 0};
    for (int k = 0; k < 4; k++) {
        const float r = two_stream_rate(ctx, ae[k >> 1], be[k & 1], sp);
        if (trace) fprintf(stderr, "[bhip place]   %p %s x %p %s: %.0f GB/s (%.3f)\n", a, k >> 1 ? "tail" : "head", b, k & 1 ? "tail" : "head", r, r / ctx->r_same);
        pr.smallest = std::min(pr.smallest, r); pr.mean += r; pr.runs++;
        if (pr.smallest < stop_below) break;
    }
    pr.mean /= (float)pr.runs;
    return pr;
}
static bool place_apart(const bhip_ctx *ctx, const PairRates &pr)
{
    return pr.runs == 4 && pr.smallest >= PLACE.smallest_min * ctx->r_same && pr.mean >= PLACE.mean_min * ctx->r_same;
}
// the piece of [ptr, ptr + bytes) by the context's map: the id of the representative it shares a piece with; a NEW id (the smallest
// unused one, *is_new set) when it lies apart from every piece the map knows; -1 when the tests are inconclusive (a buffer astride
// a cut, a plain allocation mixed from several pieces), the map is empty, or there is no memory for the save area.  apart_piece: a piece
// the buffer is already known to lie apart from (the W of its own pair), not tested again.
// WHAT IS WRITTEN: the tested ranges of [ptr, ptr + bytes) -- its head and tail -- are overwritten.  The representatives are buffers of
// LIVE ensembles (their W holds the chains' state): the head and tail ranges of a representative that the write streams go over are saved
// to a scratch allocation first and restored afterwards, on the context's stream, in order -- an ensemble created, or a foreign buffer
// classified, beside older ensembles leaves their W / Xo bit for bit as they were (advisor r5; tests/test_gpu_pc.py steps the FIRST of six
// ensembles after the sixth was placed).
static int place_classify(bhip_ctx *ctx, void *ptr, size_t bytes, bool *is_new, int apart_piece = -1)
{
    if (is_new) *is_new = false;
    if (!(ctx->r_same > 0.f) || bytes < PLACE.min_bytes) return -1;
    bool seen[3] = {false, false, false}, all_apart = true;
    int known = 0, result = -2;
    char *save = nullptr; size_t save_bytes = 0;
    for (const bhip_ctx::PieceEnt &e : ctx->pieces) {
        if (e.piece < 0 || e.piece > 2 || seen[e.piece] || e.p == ptr || e.bytes < PLACE.min_bytes) continue;
        seen[e.piece] = true; known++;
        if (e.piece == apart_piece) continue;
        // the ranges place_pair_rates writes in the representative: sp bytes at its head and at its tail (same arithmetic)
        const size_t sp = place_stream_bytes(e.bytes / 2, bytes / 2);
        char *head = (char *)e.p, *tail = (char *)e.p + (e.bytes - sp) / 4096 * 4096;
        if (save_bytes < 2 * sp) {
            if (save) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(save); save = nullptr; }
            if (hipMalloc((void **)&save, 2 * sp) != hipSuccess) { (void)hipGetLastError(); save = nullptr; result = -1; break; }
            save_bytes = 2 * sp;
        }
        if (hipMemcpyAsync(save, head, sp, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(save + sp, tail, sp, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { (void)hipGetLastError(); result = -1; break; }
        const PairRates pr = place_pair_rates(ctx, e.p, e.bytes, ptr, bytes, 0.f, false);
        const bool restored = hipMemcpyAsync(head, save, sp, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
                              hipMemcpyAsync(tail, save + sp, sp, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess;
        if (!restored) { (void)hipGetLastError(); ctx->err = "bhip placement: restoring a live ensemble's buffer after a write-stream test failed"; result = -1; break; }
        if (!(pr.smallest > 0.f)) { result = -1; break; }
        if (pr.mean <= PLACE.same_mean_max * ctx->r_same) { result = e.piece; break; }
        if (!place_apart(ctx, pr)) all_apart = false;
    }
    if (save) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(save); }
    if (result != -2) return result;
    if (!known || !all_apart || known >= 3) return -1;
    for (int k = 0; k < 3; k++)
        if (!seen[k]) { if (is_new) *is_new = true; return k; }
    return -1;
}

// r_same of the context: the median of the two-stream runs INSIDE each of the given contiguous runs (head x tail, and head x middle, middle x tail
// where a run holds three regions of sb bytes); false when no run could be timed
static bool place_measure_one_piece_rate(bhip_ctx *ctx, int n, void *const *runs, const size_t *nbs, size_t sb, bool trace)
{
    std::vector<float> rr;
    for (int b = 0; b < n; b++) {
        char *p = (char *)runs[b];
        const size_t nb = nbs[b], mid = (nb - sb) / 2 / 4096 * 4096, tail = (nb - sb) / 4096 * 4096;
        rr.push_back(two_stream_rate(ctx, p, p + tail, sb));
        if (nb >= 3 * sb) { rr.push_back(two_stream_rate(ctx, p, p + mid, sb)); rr.push_back(two_stream_rate(ctx, p + mid, p + tail, sb)); }
    }
    if (trace) { fprintf(stderr, "[bhip place] one-piece runs:"); for (float r : rr) fprintf(stderr, " %.0f", r); fprintf(stderr, " GB/s\n"); }
    rr.erase(std::remove_if(rr.begin(), rr.end(), [](float r) { return !(r > 0.f); }), rr.end());
    if (rr.empty()) return false;
    std::sort(rr.begin(), rr.end());
    ctx->r_same = 0.5f * (rr[(rr.size() - 1) / 2] + rr[rr.size() / 2]);
    return true;
}

static int chains_step_once(bhip_chains *ch, double rho, int skip, bool store_x);
// BHIP_PLACE_TRACE only: ms per pCN iteration of the ensemble with W at wq and the proposal paths at q (the ground truth the
// two-stream readings are compared with in profiles/r5_piece_map.txt); the ensemble's state is set up again afterwards by the caller
static float place_trace_iteration_ms(bhip_chains *ch, const double *x0, int skip, void *wq, void *q)
{
    void *w0 = ch->arena.base, *x0p = ch->arena.base2;
    ch->arena.base = wq; ch->arena.base2 = q; ch->Wc = (double *)wq; ch->Xo = (double *)q;
    float ms = 0.f;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && !chains_init_impl(ch, x0, nullptr, 0, skip, 0u)) {
        for (int it = 0; it < 2; it++) (void)chains_step_once(ch, 0.9, skip, true);
        (void)hipEventRecord(e0, ch->ctx->stream);
        for (int it = 0; it < 4; it++) (void)chains_step_once(ch, 0.9, skip, true);
        (void)hipEventRecord(e1, ch->ctx->stream);
        if (hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    ch->arena.base = w0; ch->arena.base2 = x0p; ch->Wc = (double *)w0; ch->Xo = (double *)x0p;
    return ms / 4;
}

static int chains_place(bhip_chains *ch, const double *x0, int skip)
{
    bhip_ctx *ctx = ch->ctx;
    void *w = ch->arena.base, *xo = ch->arena.base2;
    const size_t sb = place_stream_bytes(ch->wbytes / 2, ch->xbytes / 2);
    if (sb < PLACE.min_bytes / 2) return BHIP_OK;
    // 1. the rate inside one piece, once per context: the median of six runs inside W and inside Xo
    const bool trace = getenv("BHIP_PLACE_TRACE") != nullptr;   // every two-stream run of the procedure to stderr (scripts/gpu_piece_map_probe.py)
    if (!(ctx->r_same > 0.f)) {
        void *runs[2] = {w, xo}; const size_t nbs[2] = {ch->wbytes, ch->xbytes};
        if (!place_measure_one_piece_rate(ctx, 2, runs, nbs, sb, trace)) return chains_init_impl(ch, x0, nullptr, 0, skip, 0u);   // no measurement: the pair stays as it is
    }
    auto can_alloc = [&](size_t bytes) { size_t f = 0, t = 0; return hipMemGetInfo(&f, &t) == hipSuccess && f >= 2 * bytes; };
    // 2./3. candidates for Xo.  A contiguous run straddles at most one cut, so its head and its tail tell where ALL of it lies: a pair is
    //    good when the four two-stream runs -- head and tail of W against head and tail of the candidate -- pass place_apart (the runs end
    //    at the first one at the one-piece rate).  (Head against head alone is no test: pairs of ONE piece read up to 1.27 x the one-piece
    //    rate there -- 1.65 ms per iteration on the round-5 profile box where 1.39 was due.)  Score = the mean of the four.  When every
    //    candidate fails against this W another W is allocated (twice at most) and the candidates -- still held -- are judged against
    //    it.  Without a good pair the best-scoring one is kept.
    struct Cand { void *w, *x; float score; bool good; };
    const float floor_rate = PLACE.smallest_min * ctx->r_same;
    if (trace) fprintf(stderr, "[bhip place] %ld chains, W %zu MiB, Xo %zu MiB, one-piece rate %.0f GB/s\n", ch->n, ch->wbytes >> 20, ch->xbytes >> 20, ctx->r_same);
    auto judge = [&](void *wq, void *q) {
        const PairRates pr = place_pair_rates(ctx, wq, ch->wbytes, q, ch->xbytes, trace ? 0.f : floor_rate, trace);
        if (trace) fprintf(stderr, "[bhip place]   -> smallest %.3f, mean %.3f, %.4f ms per iteration with this pair\n", pr.smallest / ctx->r_same, pr.mean / ctx->r_same, place_trace_iteration_ms(ch, x0, skip, wq, q));
        return Cand{wq, q, pr.runs == 4 ? pr.mean : std::min(pr.mean, pr.smallest), place_apart(ctx, pr)};   // (a pair judged on all four runs outranks one that fell at a corner)
    };
    const int cands0 = (int)std::min<size_t>(24, std::max<size_t>((size_t)PLACE.max_candidates, PLACE.held_bytes / std::max<size_t>(ch->xbytes, 1)));
    std::vector<void *> ws{w}, xs{xo}, spacers;
    Cand best{w, xo, 0.f, false};
    for (int attempt = 0; attempt < 3 && !best.good; attempt++) {
        void *wq = ws.back();
        const int max_cands = cands0 + attempt * PLACE.more_candidates;   // (a W re-roll walks on: the allocator has not left W's piece yet)
        for (size_t k = 0; k < xs.size() && !best.good; k++) {
            const Cand c = judge(wq, xs[k]);
            if (c.good || c.score > best.score) best = c;
        }
        while (!best.good && (int)xs.size() < max_cands && can_alloc(ch->xbytes)) {
            void *q = nullptr;
            if (alloc_run(&q, ch->xbytes) != hipSuccess) { (void)hipGetLastError(); break; }
            xs.push_back(q);
            const Cand c = judge(wq, q);
            if (c.good || c.score > best.score) best = c;
        }
        if (best.good || attempt == 2 || !can_alloc(ch->wbytes)) break;
        // every candidate so far shares W's piece: the allocator is walking through it (up to 96 GiB).  A spacer -- a plain allocation,
        // never written, freed below -- takes the next stretch of that walk in one step, so that the next W lies beyond it
        void *q = nullptr;
        if (can_alloc(PLACE.spacer_bytes + ch->wbytes) && hipMalloc(&q, PLACE.spacer_bytes) == hipSuccess) spacers.push_back(q);
        else (void)hipGetLastError();
        q = nullptr;
        if (alloc_run(&q, ch->wbytes) != hipSuccess) { (void)hipGetLastError(); break; }
        ws.push_back(q);
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (void *q : spacers) (void)hipFree(q);
    for (void *q : xs) if (q != best.x) (void)hipFree(q);
    for (void *q : ws) if (q != best.w) (void)hipFree(q);
    w = best.w; xo = best.x;
    ch->arena.base = w; ch->arena.base2 = xo;
    ch->Wc = (double *)w; ch->Xo = (double *)xo;
    // 4. the map learns both buffers of a good pair: W against the map (the first ensemble founds it: piece 0), then Xo
    bool fresh = false;
    int pw = -1, px = -1;
    if (best.good) {
        pw = ctx->pieces.empty() ? 0 : place_classify(ctx, w, ch->wbytes, &fresh);
        if (pw >= 0) {
            ctx->pieces.push_back(bhip_ctx::PieceEnt{w, ch->wbytes, pw});   // (so that Xo is not given W's id as a new one)
            px = place_classify(ctx, xo, ch->xbytes, &fresh, pw);
            if (px >= 0) ctx->pieces.push_back(bhip_ctx::PieceEnt{xo, ch->xbytes, px});
        }
    }
    ch->place_tries = (int)(xs.size() + ws.size() - 1); ch->place_gbs_same = ctx->r_same; ch->place_gbs_kept = best.score;
    ch->piece_w = pw; ch->piece_xo = px;
    return chains_init_impl(ch, x0, nullptr, 0, skip, 0u);   // the state of iteration 0: the write streams went over W and Xo
}

int bhip_chains_init(bhip_chains *ch, const double *x0, int skip)
{
    if (!ch || !x0) return BHIP_EINVAL;
    const int rc = chains_init_impl(ch, x0, nullptr, 0, skip, 0u);
    if (rc || !ch->ctx->tune_placement || !ch->arena.base2 || ch->shares_state || ch->place_tries > 0) return rc;
    return chains_place(ch, x0, skip);
}

int bhip_chains_placement_info(const bhip_chains *ch, int *tries, float *gbs_same_piece, float *gbs_kept)
{
    if (!ch) return BHIP_EINVAL;
    if (tries) *tries = ch->place_tries;
    if (gbs_same_piece) *gbs_same_piece = ch->place_gbs_same;
    if (gbs_kept) *gbs_kept = ch->place_gbs_kept;
    return BHIP_OK;
}

int bhip_chains_placement_pieces(const bhip_chains *ch, int *piece_w, int *piece_xo)
{
    if (!ch) return BHIP_EINVAL;
    if (piece_w) *piece_w = ch->piece_w;
    if (piece_xo) *piece_xo = ch->piece_xo;
    return BHIP_OK;
}

int bhip_ctx_piece_of(bhip_ctx *ctx, void *dev_ptr, size_t bytes, int *piece)
{
    if (!ctx || !dev_ptr || !piece) return BHIP_EINVAL;
    NEED_DEVICE(ctx);
    *piece = -1;
    if (ctx->pieces.empty() || !(ctx->r_same > 0.f)) return fail(ctx, BHIP_ESTATE, "bhip_ctx_piece_of: the context's piece map is empty (it is built by the first placed chain ensemble and lives with the ensembles)");
    if (bytes < PLACE.min_bytes) return fail(ctx, BHIP_EINVAL, "bhip_ctx_piece_of: the buffer is too small to be classified (64 MiB at least)");
    for (const bhip_ctx::PieceEnt &e : ctx->pieces)
        if (e.p == dev_ptr) { *piece = e.piece; return BHIP_OK; }   // a buffer the map holds: no test, nothing overwritten
    bool fresh = false;
    *piece = place_classify(ctx, dev_ptr, bytes, &fresh);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BHIP_OK;
}

// nparts (1..3) contiguous runs of `bytes` each that lie pairwise in DIFFERENT pieces of the device memory (the four-run test of
// chains_place between every two of them; candidates that fail stay held until the set is complete, then go back) -- for ensembles kept in
// parts (bhip_sample_solve_parts): one write stream per piece moves 5.8-6.0 TB/s over two pieces and 6.8-6.9 over three where one stream in
// one piece moves 4.3-4.4 (profiles/r5_three_pieces.txt).  *apart = how many of the parts ended up pairwise apart (nparts: all).
int bhip_alloc_apart(bhip_ctx *ctx, int nparts, size_t bytes, void **out, int *apart)
{
    if (!ctx || !out || nparts < 1 || nparts > 3 || !bytes) return BHIP_EINVAL;
    NEED_DEVICE(ctx);
    for (int j = 0; j < nparts; j++) out[j] = nullptr;
    if (apart) *apart = 0;
    auto release = [&](std::vector<void *> &v) { for (void *q : v) (void)hipFree(q); v.clear(); };
    std::vector<void *> kept, held, spacers;
    const size_t sb = place_stream_bytes(bytes / 2, bytes / 2);
    const bool testable = bytes >= PLACE.min_bytes && sb >= PLACE.min_bytes / 2;
    if (!testable) {
        // too small for write-stream tests (and for placement to matter): plain allocations, as they come.  NOT small contiguous runs: with
        // hipDeviceMallocContiguous buffers of 0.8-2.5 MB allocated and freed between other work, later downloads of OTHER buffers of the
        // process read page-sized stretches of zeros where the memory held values (tests/test_gpu_parts.py inside the whole suite, 3 of 3;
        // gone with plain allocations, 3 of 3) -- the flag stays with the large runs it was measured on
        for (int j = 0; j < nparts; j++)
            if (hipMalloc(&out[j], bytes) != hipSuccess) {
                (void)hipGetLastError();
                for (int k = 0; k < j; k++) { (void)hipFree(out[k]); out[k] = nullptr; }
                out[j] = nullptr;
                return fail(ctx, BHIP_EHIP, "bhip_alloc_apart: out of device memory");
            }
        for (int j = 0; j < nparts; j++) { { std::lock_guard<std::mutex> g(ctx->buf_mu); ctx->bufs.insert(out[j]); } ctx_retain(ctx); }
        return BHIP_OK;
    }
    void *p = nullptr;
    if (alloc_run(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, BHIP_EHIP, "bhip_alloc_apart: out of device memory"); }
    kept.push_back(p);
    if (!(ctx->r_same > 0.f)) (void)place_measure_one_piece_rate(ctx, 1, &p, &bytes, sb, false);
    const bool judge = ctx->r_same > 0.f;
    int n_apart = 1;
    const int max_cands = (int)std::min<size_t>(24, std::max<size_t>((size_t)PLACE.max_candidates + 2 * PLACE.more_candidates, PLACE.held_bytes / bytes));
    while ((int)kept.size() < nparts) {
        void *best = nullptr; float best_score = -1.f; bool good = false;
        auto score_of = [&](void *q, bool *ok) {   // the smallest mean-of-four against the parts kept so far
            float sc = 1e30f; *ok = true;
            for (void *k : kept) {
                const PairRates pr = place_pair_rates(ctx, k, bytes, q, bytes, PLACE.smallest_min * ctx->r_same, false);
                if (!place_apart(ctx, pr)) *ok = false;
                sc = std::min(sc, pr.runs == 4 ? pr.mean : std::min(pr.mean, pr.smallest));
                if (!*ok) break;
            }
            return sc;
        };
        for (void *q : held) {   // (a candidate that failed against an earlier part may do for this one)
            bool ok = false; const float sc = judge ? score_of(q, &ok) : 0.f;
            if (ok) { best = q; good = true; break; }
            if (sc > best_score) { best_score = sc; best = q; }
        }
        // candidates; when every one of them shares a piece with a kept part the allocator is still walking through that piece (up to 96 GiB,
        // block by block -- runs of 2-GB blocks are longer than the 4-GB ones the candidate count was chosen for): a spacer -- a plain allocation,
        // never written, freed below -- takes the next stretch of the walk in one step, twice at most, with a few more candidates beyond it
        // (the same device as chains_place; round 6: one default `proposals` container in four came out with both buffers in one piece)
        int extra = 0;
        for (int round = 0;; round++) {
            while (!good && (int)held.size() < max_cands + extra) {
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < 2 * bytes) break;
                void *q = nullptr;
                if (alloc_run(&q, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
                held.push_back(q);
                if (!judge) { best = q; good = false; break; }
                bool ok = false; const float sc = score_of(q, &ok);
                if (ok) { best = q; good = true; break; }
                if (sc > best_score) { best_score = sc; best = q; }
            }
            if (good || !judge || round == 2) break;
            size_t fr = 0, tot = 0;
            void *sp = nullptr;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < PLACE.spacer_bytes + 4 * bytes || hipMalloc(&sp, PLACE.spacer_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            spacers.push_back(sp);
            extra += 6;
        }
        if (!best) { (void)hipStreamSynchronize(ctx->stream); release(held); release(kept); release(spacers); return fail(ctx, BHIP_EHIP, "bhip_alloc_apart: out of device memory"); }
        held.erase(std::find(held.begin(), held.end(), best));
        kept.push_back(best);
        if (good) n_apart++;
    }
    (void)hipStreamSynchronize(ctx->stream);
    release(held);
    release(spacers);
    for (int j = 0; j < nparts; j++) { out[j] = kept[j]; { std::lock_guard<std::mutex> g(ctx->buf_mu); ctx->bufs.insert(out[j]); } ctx_retain(ctx); }
    if (apart) *apart = judge ? n_apart : 0;
    return BHIP_OK;
}

// the parts are children of the context like bhip_malloc buffers (each holds a reference: a finalizer may give them back after the
// context was destroyed -- Julia runs finalizers in no particular order at exit; advisor r5)
int bhip_free_apart(bhip_ctx *ctx, int nparts, void *const *ptrs)
{
    if (!ctx || !ptrs || nparts < 0) return BHIP_EINVAL;
    if (ctx->host_only) return fail(ctx, BHIP_EHIP, "host-only context (device -1): no device memory");
    int rc = BHIP_OK;
    bool quiesced = false;
    ctx_retain(ctx);   // (the last part's release may be the context's last reference: it must outlive the loop and its messages)
    for (int j = 0; j < nparts; j++) {
        if (!ptrs[j]) continue;
        {
            std::lock_guard<std::mutex> g(ctx->buf_mu);
            if (ctx->bufs.erase(ptrs[j]) == 0) { rc = fail(ctx, BHIP_EINVAL, "bhip_free_apart: not a live bhip_alloc_apart buffer of this context (foreign pointer or double free)"); continue; }
        }
        if (!quiesced) { ctx_quiesce(ctx); quiesced = true; }
        if (hipFree(ptrs[j]) != hipSuccess) { (void)hipGetLastError(); rc = BHIP_EHIP; }
        ctx_release(ctx);
    }
    ctx_release(ctx);
    return rc;
}

// ONE pCN proposal of every chain of a segment with the decision deferred (multi-segment ensembles): Wo = w_old*W + w_new*W2,
// starts from x0_dev, proposal paths to Xo, llo to llo_dev; cur / llcur / acc are left alone (bhip_segchains_step decides)
// launches on per-chain coefficient rows (device-built guides): the monolithic kernels, PerPathRow instantiations
static int launch_ppr(bhip_chains *ch, int noise, KArgs &a, hipStream_t st = nullptr, bool on_st = false)   // on_st: launch on `st` instead of the context's stream
{
    bhip_ctx *ctx = ch->ctx;
    const bhip_proposal *po = ch->po;
    a.prows = ch->prows; a.ldr = ch->ld; a.vend_pc = ch->vend_pc; a.uv_pc = ch->uv_pc; a.lna = ch->lna;
    const int fl = (noise == NOISE_PCN || noise == NOISE_PCN_LINES) ? (a.Xo ? 1 : 0) : 0;
    // (bit 1 of the selector: the monolithic line kernel instead of the wave-specialised one)
    if (a.noise_spec != 4 && noise == NOISE_PCN) return fail(ctx, BHIP_EUNSUPPORTED, "BHIP_OPT_NOISE_SPEC = 2 / 3: the pCN step on the 16-byte slots draws the default noise stream only");
    if (a.Xtb && !(noise == NOISE_PCN_LINES && a.rdtp)) return fail(ctx, BHIP_ESTATE, "time-blocked paths need the line layout");
    launch_fn f = find_launch_ppr(po->mh, noise, fl | ((noise == NOISE_PCN_LINES && !((ctx->wave_specialised || a.noise_spec != 4 || a.Xtb) && a.rdtp)) ? 2 : 0));
    if (!f) return fail(ctx, BHIP_EUNSUPPORTED, "no per-chain-guide kernel for this model");
    HIPCHK(ctx, f(a, on_st ? st : ctx->stream));
    return BHIP_OK;
}

static int chains_propose_deferred(bhip_chains *ch, double w_old, double w_new, const double *x0_dev, long ldx0, uint32_t iter, uint32_t blk0,
                                   double *llo_dev, int skip)
{
    const bhip_proposal *po = ch->po;
    if (ch->tile)   // the MFMA tile kernel's chain step with the decision deferred
        return launch_tile_path(po, ch->x0.data(), nullptr, 0, nullptr, 0, ch->Xo, ch->ld, llo_dev, skip, ch->n, 2, ch->seed, iter, ch->path0, 1, ch, w_old,
                                x0_dev, ldx0, blk0, 1, w_new);
    KArgs a;
    int rc = fill_common(po, a, ch->x0.data(), x0_dev, ch->n, skip);
    if (rc) return rc;
    a.x0_dev = x0_dev; a.ldx0 = ldx0; a.blk0 = blk0; a.defer_accept = 1; a.ll = llo_dev;
    a.Wc = ch->Wc; a.Xo = ch->Xo; a.ldC = ch->ld;
    if (ch->Xtb) { a.Xo = nullptr; a.Xtb = ch->Xtb; a.xtb_half = ch->xtb_half; a.xend = ch->xend; a.xsel = ch->xsel; }   // (the instantiation without the plain X store)
    a.cur = ch->cur; a.llcur = ch->llcur; a.acc = ch->acc;
    a.rho = w_old; a.srho = w_new;
    a.k0 = (uint32_t)ch->seed; a.k1 = (uint32_t)(ch->seed >> 32); a.path0 = ch->path0; a.iter = iter;
    if (ch->prows) return launch_ppr(ch, ch->lines ? NOISE_PCN_LINES : NOISE_PCN, a);
    return do_launch(po, ch->lines ? NOISE_PCN_LINES : NOISE_PCN, a);
}

// llikelihood(LeftRule(), X, Po) of every chain's path under the chain's OWN guide
static int chains_llikelihood_ppr(bhip_chains *ch, const double *X_dev, long ldX, double *ll_dev, int skip, hipStream_t st = nullptr, bool on_st = false)
{
    KArgs a;
    int rc = fill_common(ch->po, a, ch->x0.data(), nullptr, ch->n, skip);
    if (rc) return rc;
    a.Win = X_dev; a.ldWin = ldX; a.ll = ll_dev;
    return launch_ppr(ch, NOISE_LLONLY, a, st, on_st);
}

// argument checks of bhip_chains_step / bhip_chains_step_group for one ensemble; resolves BHIP_SKIP_OF_INIT
static int chains_step_check(bhip_chains *ch, double rho, int iters, int &skip)
{
    bhip_ctx *ctx = ch->ctx;
    if (!ch->inited) return fail(ctx, BHIP_ESTATE, "bhip_chains_step: call bhip_chains_init first");
    if (ch->noise_spec != ctx->noise_spec) return fail(ctx, BHIP_ESTATE, "the context's noise specification changed after the ensemble was created");
    if (iters < 0) return fail(ctx, BHIP_EINVAL, "iters must be >= 0");
    if (!(rho >= -1.0 && rho <= 1.0)) return fail(ctx, BHIP_EINVAL, "rho must lie in [-1, 1] (sqrt(1 - rho^2) is the weight of the fresh noise)");
    if (skip == BHIP_SKIP_OF_INIT) skip = ch->skip0;   // llo and ll then always sum the same terms
    if (skip < 0) return fail(ctx, BHIP_EINVAL, "skip must be >= 0");
    // the line kernels read cur[] and the W lines of whole 64-chain groups: ld is padded to 64 and those arrays are sized by ld
    if (ch->lines && (ch->ld % 64 != 0 || ch->ld < ch->n)) return fail(ctx, BHIP_ESTATE, "chain storage: leading dimension must be a multiple of 64 covering all chains");
    return BHIP_OK;
}
// ONE pCN iteration of one ensemble, launched on its context's stream.  The proposal buffer Xo is overwritten by every
// iteration, so within one call only the LAST iteration's store can ever be observed: the earlier ones (store_x = false) run
// the instantiation without the store.
static int chains_step_once(bhip_chains *ch, double rho, int skip, bool store_x)
{
    const bhip_proposal *po = ch->po;
    // (ch->iter counts COMPLETED iterations -- bhip_chains_iterations --: an iteration whose launch failed is not counted)
    if (ch->tile) {
        ++ch->iter;
        const int rct = launch_tile_path(po, ch->x0.data(), nullptr, 0, nullptr, 0, store_x ? ch->Xo : nullptr, ch->ld, nullptr, skip, ch->n, 2, ch->seed, ch->iter, ch->path0, 1, ch, rho);
        if (rct) --ch->iter;
        return rct;
    }
    KArgs a;
    int rc = fill_common(po, a, ch->x0.data(), nullptr, ch->n, skip);   // (makes the context's device current)
    if (rc) return rc;
    ++ch->iter;
    a.Wc = ch->Wc; a.Xo = store_x ? ch->Xo : nullptr; a.ldC = ch->ld;
    a.cur = ch->cur; a.llcur = ch->llcur; a.acc = ch->acc;
    a.rho = rho; a.srho = std::sqrt(1 - rho * rho);
    a.k0 = (uint32_t)ch->seed; a.k1 = (uint32_t)(ch->seed >> 32); a.path0 = ch->path0;
    a.iter = ch->iter;
    rc = do_launch(po, ch->lines ? NOISE_PCN_LINES : NOISE_PCN, a);
    if (rc) --ch->iter;
    return rc;
}

int bhip_chains_step(bhip_chains *ch, double rho, int iters, int skip)
{
    if (!ch) return BHIP_EINVAL;
    int rc = chains_step_check(ch, rho, iters, skip);
    for (int it = 0; it < iters && !rc; it++) rc = chains_step_once(ch, rho, skip, it == iters - 1);
    return rc;
}

// The same for n ensembles -- one per device of a node, each on its own context -- in ONE call: iteration by iteration the
// launches go out round-robin, each on its context's stream (asynchronous: a single host thread, e.g. a Julia ccall host,
// keeps every device busy and crosses the FFI once per call instead of once per device and iteration).  Results are those of
// stepping every ensemble by itself (tests/test_gpu_group.py).
int bhip_chains_step_group(int n, bhip_chains *const *chs, double rho, int iters, int skip)
{
    if (n < 1 || !chs) return BHIP_EINVAL;
    constexpr int MAXG = 64;
    if (n > MAXG) return BHIP_EINVAL;
    int skips[MAXG];
    for (int k = 0; k < n; k++) {
        if (!chs[k]) return BHIP_EINVAL;
        for (int j = 0; j < k; j++) if (chs[j] == chs[k]) return fail(chs[k]->ctx, BHIP_EINVAL, "bhip_chains_step_group: the same ensemble twice");
        skips[k] = skip;
        const int rc = chains_step_check(chs[k], rho, iters, skips[k]);
        if (rc) return rc;
    }
    for (int it = 0; it < iters; it++)
        for (int k = 0; k < n; k++) {
            const int rc = chains_step_once(chs[k], rho, skips[k], it == iters - 1);
            if (rc) return rc;
        }
    return BHIP_OK;
}

int bhip_chains_iterations(const bhip_chains *ch, uint32_t *iterations)
{
    if (!ch || !iterations) return BHIP_EINVAL;
    *iterations = ch->iter;
    return BHIP_OK;
}

int bhip_chains_stats(bhip_chains *ch, double *stats_dev)
{
    if (!ch || !stats_dev) return BHIP_EINVAL;
    bhip_ctx *ctx = ch->ctx;
    if (!ch->inited) return fail(ctx, BHIP_ESTATE, "bhip_chains_stats: chains not initialised");
    NEED_DEVICE(ctx);
    const int nparts = (int)std::min<long>(256, (ch->n + 1023) / 1024);
    hipLaunchKernelGGL(k_chain_stats_partial, dim3(nparts), dim3(256), 0, ctx->stream, ch->llcur, ch->acc, ch->n, ch->statpart);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_chain_stats_final, dim3(1), dim3(256), 0, ctx->stream, ch->statpart, nparts, ch->n, (double)ch->iter, stats_dev);
    HIPCHK(ctx, hipGetLastError());
    return BHIP_OK;
}

// bhip_chains_stats of n ensembles (one per device), each reduction on its own context's stream, in one call
int bhip_chains_stats_group(int n, bhip_chains *const *chs, double *const *stats_dev)
{
    if (n < 1 || !chs || !stats_dev) return BHIP_EINVAL;
    for (int k = 0; k < n; k++) {
        if (!chs[k] || !stats_dev[k]) return BHIP_EINVAL;
        const int rc = bhip_chains_stats(chs[k], stats_dev[k]);
        if (rc) return rc;
    }
    return BHIP_OK;
}

int bhip_chains_get(bhip_chains *ch, double *ll, int64_t *acc)
{
    if (!ch) return BHIP_EINVAL;
    bhip_ctx *ctx = ch->ctx;
    NEED_DEVICE(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ll) HIPCHK(ctx, hipMemcpy(ll, ch->llcur, sizeof(double) * ch->n, hipMemcpyDeviceToHost));
    if (acc) {
        std::vector<unsigned int> tmp(ch->n);
        HIPCHK(ctx, hipMemcpy(tmp.data(), ch->acc, sizeof(unsigned int) * ch->n, hipMemcpyDeviceToHost));
        for (long p = 0; p < ch->n; p++) acc[p] = tmp[p];
    }
    return BHIP_OK;
}

// current W of chains p0..p0+np gathered from the slots into a plain SoA array [N][mp][np]
static int gather_current_W(bhip_chains *ch, long p0, long np, double *W_soa)
{
    bhip_ctx *ctx = ch->ctx;
    if (ch->lines) {
        const int N = (int)ch->po->tt.size();
        hipLaunchKernelGGL(k_lines_to_soa, dim3((unsigned)((np + 63) / 64), (unsigned)ch->nch), dim3(256), 0, ctx->stream, ch->Wc, ch->cur, N, ch->po->mh.mp, ch->nch, ch->ld, p0, np, W_soa);
        HIPCHK(ctx, hipGetLastError());
        return BHIP_OK;
    }
    if (ch->tile) {
        const int N = (int)ch->po->tt.size(), d = ch->po->mh.d, T = tile_dim(d) / 16;
        const long tot = (long)N * d * np;
        hipLaunchKernelGGL(k_tlines_to_soa, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, ch->Wc, ch->cur, W_soa, N, d, T, ch->ld, p0, np);
        HIPCHK(ctx, hipGetLastError());
        return BHIP_OK;
    }
    const long E = (long)ch->po->tt.size() * ch->po->mh.mp;
    const long tot = E * np;
    hipLaunchKernelGGL(k_slots_to_soa, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, ch->Wc, ch->cur, W_soa, E, ch->ld, p0, np);
    HIPCHK(ctx, hipGetLastError());
    return BHIP_OK;
}

// The chain state is (W, ll, parity); the current X is a deterministic function of the current W
// (X = solve!(Euler(), x0, W, Po)), so it is re-materialised on demand -- bit-identical to the Xo
// the chain stored when that W was accepted -- instead of being double-buffered every iteration.
static int current_X(bhip_chains *ch, long p0, long np, double *W_soa, double *X_soa)
{
    int rc = gather_current_W(ch, p0, np, W_soa);
    if (rc) return rc;
    if (ch->tile) return launch_tile_path(ch->po, ch->x0.data(), W_soa, np, nullptr, 0, X_soa, np, nullptr, 0, np, 0, 0, 0, 0);
    KArgs a;
    rc = fill_common(ch->po, a, ch->x0.data(), nullptr, np, 0);
    if (rc) return rc;
    a.Win = W_soa; a.ldWin = np; a.X = X_soa; a.ldX = np;
    return do_launch(ch->po, NOISE_EXT, a);
}

int bhip_chains_get_paths(bhip_chains *ch, long p0, long np, double *X_aos, double *W_aos)
{
    if (!ch) return BHIP_EINVAL;
    bhip_ctx *ctx = ch->ctx;
    if (!ch->inited) return fail(ctx, BHIP_ESTATE, "chains not initialised");
    NEED_DEVICE(ctx);
    if (p0 < 0 || np < 0 || p0 + np > ch->n) return fail(ctx, BHIP_EINVAL, "chain range out of bounds");
    if (np == 0 || (!X_aos && !W_aos)) return BHIP_OK;
    const long N = (long)ch->po->tt.size(), d = ch->po->mh.d, mp = ch->po->mh.mp;
    double *tmp = nullptr;
    const size_t nW = (size_t)N * mp * np, nX = (size_t)N * d * np;
    HIPCHK(ctx, hipMalloc((void **)&tmp, sizeof(double) * (nW + nX)));
    int rc = X_aos ? current_X(ch, p0, np, tmp, tmp + nW) : gather_current_W(ch, p0, np, tmp);
    if (!rc && W_aos) rc = bhip_download_aos(ctx, tmp, (int)N, (int)mp, np, 0, np, W_aos);
    if (!rc && X_aos) rc = bhip_download_aos(ctx, tmp + nW, (int)N, (int)d, np, 0, np, X_aos);
    return sync_free_rc(ctx, tmp, rc);
}

int bhip_chains_current_X(bhip_chains *ch, double *X_dev, long ldX)
{
    if (!ch || !X_dev) return BHIP_EINVAL;
    bhip_ctx *ctx = ch->ctx;
    if (!ch->inited) return fail(ctx, BHIP_ESTATE, "chains not initialised");
    NEED_DEVICE(ctx);
    if (ldX < ch->n) return fail(ctx, BHIP_ELENGTH, "leading dimension smaller than the number of chains");
    const long N = (long)ch->po->tt.size(), mp = ch->po->mh.mp;
    double *tmp = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&tmp, sizeof(double) * (size_t)N * mp * ch->n));
    int rc = gather_current_W(ch, 0, ch->n, tmp);
    if (!rc && ch->tile) rc = launch_tile_path(ch->po, ch->x0.data(), tmp, ch->n, nullptr, 0, X_dev, ldX, nullptr, 0, ch->n, 0, 0, 0, 0);
    else if (!rc) {
        KArgs a;
        rc = fill_common(ch->po, a, ch->x0.data(), nullptr, ch->n, 0);
        if (!rc) { a.Win = tmp; a.ldWin = ch->n; a.X = X_dev; a.ldX = ldX; rc = do_launch(ch->po, NOISE_EXT, a); }
    }
    return sync_free_rc(ctx, tmp, rc);
}

/* ---- checkpoint / resume */
namespace {
struct ChainStateHeader {
    uint64_t magic;      // "BHIPCHN1"
    int64_t n, N, mp, d;
    uint64_t seed;
    uint32_t path0, iter;
    int32_t skip0, rng_spec;   // rng_spec: version of the noise specification the chains were driven with (bhip-philox-v<rng_spec>)
};
constexpr uint64_t CHAIN_MAGIC = 0x314E484350494842ULL;   // "BHIPCHN1" little endian
}  // namespace

int bhip_chains_state_bytes(const bhip_chains *ch, size_t *bytes)
{
    if (!ch || !bytes) return BHIP_EINVAL;
    const size_t N = ch->po->tt.size();
    *bytes = sizeof(ChainStateHeader) + sizeof(double) * N * ch->po->mh.mp * ch->n + sizeof(double) * ch->n + sizeof(unsigned int) * ch->n;
    return BHIP_OK;
}

int bhip_chains_save(bhip_chains *ch, void *host_buf)
{
    if (!ch || !host_buf) return BHIP_EINVAL;
    bhip_ctx *ctx = ch->ctx;
    NEED_DEVICE(ctx);
    if (!ch->inited) return fail(ctx, BHIP_ESTATE, "bhip_chains_save: chains not initialised");
    const size_t N = ch->po->tt.size(), nW = N * ch->po->mh.mp * ch->n;
    ChainStateHeader h{};
    h.magic = CHAIN_MAGIC; h.n = ch->n; h.N = (int64_t)N; h.mp = ch->po->mh.mp; h.d = ch->po->mh.d;
    h.seed = ch->seed; h.path0 = ch->path0; h.iter = ch->iter; h.skip0 = ch->skip0; h.rng_spec = ch->noise_spec;
    char *out = static_cast<char *>(host_buf);
    std::memcpy(out, &h, sizeof(h));
    double *tmp = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&tmp, sizeof(double) * nW));
    int rc = gather_current_W(ch, 0, ch->n, tmp);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(out + sizeof(h), tmp, sizeof(double) * nW, hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(out + sizeof(h) + sizeof(double) * nW, ch->llcur, sizeof(double) * ch->n, hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess)
        e = hipMemcpyAsync(out + sizeof(h) + sizeof(double) * (nW + ch->n), ch->acc, sizeof(unsigned int) * ch->n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, BHIP_EHIP, hipGetErrorString(e));
    return BHIP_OK;
}

int bhip_chains_load(bhip_chains *ch, const void *host_buf)
{
    if (!ch || !host_buf) return BHIP_EINVAL;
    bhip_ctx *ctx = ch->ctx;
    NEED_DEVICE(ctx);
    const bhip_proposal *po = ch->po;
    const size_t N = po->tt.size(), nW = N * po->mh.mp * ch->n;
    ChainStateHeader h;
    const char *in = static_cast<const char *>(host_buf);
    std::memcpy(&h, in, sizeof(h));
    if (h.magic != CHAIN_MAGIC) return fail(ctx, BHIP_EINVAL, "bhip_chains_load: not a chain state buffer");
    if (h.rng_spec != ch->noise_spec)   // 0: saved before the field existed (bhip-philox-v1 and the round-2 -v2 builds wrote 0 there)
        return fail(ctx, BHIP_EINVAL, "bhip_chains_load: the state was saved under another noise specification (" +
                                      (h.rng_spec ? "bhip-philox-v" + std::to_string(h.rng_spec) : std::string("bhip-philox-v1 or -v2: before the field existed")) +
                                      "); a resumed run would not reproduce the uninterrupted one");
    if (h.n != ch->n || h.N != (int64_t)N || h.mp != po->mh.mp || h.d != po->mh.d)
        return fail(ctx, BHIP_EINVAL, "bhip_chains_load: the state was saved for another ensemble shape (chains, grid, dimensions)");
    if (h.seed != ch->seed || h.path0 != ch->path0)
        return fail(ctx, BHIP_EINVAL, "bhip_chains_load: seed / path0 differ from the ensemble the state was saved from");
    if (!ch->inited && ch->x0.empty()) return fail(ctx, BHIP_ESTATE, "bhip_chains_load: call bhip_chains_init once first (it fixes the starting point)");
    double *tmp = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&tmp, sizeof(double) * nW));
    hipError_t e = hipMemcpyAsync(tmp, in + sizeof(h), sizeof(double) * nW, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ch->llcur, in + sizeof(h) + sizeof(double) * nW, sizeof(double) * ch->n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ch->acc, in + sizeof(h) + sizeof(double) * (nW + ch->n), sizeof(unsigned int) * ch->n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ch->cur, 0, ch->ld, ctx->stream);   // the restored W goes to half 0
    if (e == hipSuccess) {
        if (ch->lines) {
            hipLaunchKernelGGL(k_soa_to_lines, dim3((unsigned)(ch->ld / 64), (unsigned)ch->nch), dim3(256), 0, ctx->stream, tmp, ch->n, (int)N, po->mh.mp, ch->nch,
                               ch->Wc, ch->ld, ch->n);
        } else if (ch->tile) {
            const int T = tile_dim(po->mh.d) / 16;
            const long tot = (long)N * 16 * T * ch->n;
            hipLaunchKernelGGL(k_soa_to_tlines, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, tmp, ch->Wc, (int)N, po->mh.d, T, ch->ld, ch->n);
        } else {
            const long E = (long)N * po->mh.mp, tot = E * ch->n;
            hipLaunchKernelGGL(k_soa_to_slots, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, tmp, ch->Wc, E, ch->ld, ch->n);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(ctx, BHIP_EHIP, hipGetErrorString(e));
    ch->iter = h.iter; ch->skip0 = h.skip0; ch->inited = true;
    return BHIP_OK;
}

int bhip_chains_proposal_X(bhip_chains *ch, double **Xo_dev, long *ld)
{
    if (!ch || !Xo_dev || !ld) return BHIP_EINVAL;
    if (!ch->Xo) return fail(ch->ctx, BHIP_ESTATE, "chains were created without BHIP_CHAINS_STORE_X");
    *Xo_dev = ch->Xo; *ld = ch->ld;
    return BHIP_OK;
}

int bhip_chains_pathstats(bhip_chains *ch, double *mean, double *m2)
{
    if (!ch || !mean || !m2) return BHIP_EINVAL;
    bhip_ctx *ctx = ch->ctx;
    if (!ch->inited) return fail(ctx, BHIP_ESTATE, "chains not initialised");
    NEED_DEVICE(ctx);
    const int N = (int)ch->po->tt.size(), d = ch->po->mh.d;
    const size_t nm = (size_t)N * d, n2 = (size_t)N * d * d, nX = (size_t)N * d * ch->n;
    double *tmp = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&tmp, sizeof(double) * (nX + nm + n2)));
    int rc = bhip_chains_current_X(ch, tmp, ch->n);
    if (!rc) {
        hipLaunchKernelGGL(k_path_stats, dim3(N), dim3(256), 0, ctx->stream, tmp, d, ch->n, ch->n, tmp + nX, tmp + nX + nm);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(mean, tmp + nX, sizeof(double) * nm, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(m2, tmp + nX + nm, sizeof(double) * n2, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, BHIP_EHIP, hipGetErrorString(e));
    }
    return sync_free_rc(ctx, tmp, rc);
}

int bhip_welford_merge(long entries, int d, double *na, double *mean_a, double *m2_a, double nb, const double *mean_b, const double *m2_b)
{
    if (!na || !mean_a || !m2_a || !mean_b || !m2_b || d < 1 || entries < 0) return BHIP_EINVAL;
    const double n1 = *na, n2 = nb, n = n1 + n2;
    if (n2 == 0) return BHIP_OK;
    for (long e = 0; e < entries; e++) {
        double *ma = mean_a + e * d, *qa = m2_a + e * d * d;
        const double *mb = mean_b + e * d, *qb = m2_b + e * d * d;
        std::vector<double> delta(d);
        for (int k = 0; k < d; k++) delta[k] = mb[k] - ma[k];
        for (int c = 0; c < d; c++)
            for (int r = 0; r < d; r++) qa[r + d * c] = qa[r + d * c] + qb[r + d * c] + delta[r] * delta[c] * (n1 * n2 / n);
        for (int k = 0; k < d; k++) ma[k] = ma[k] + delta[k] * (n2 / n);
    }
    *na = n;
    return BHIP_OK;
}

#include "bhip_segchains.inc"

// ------------------------------------------------------------------ the collective (RCCL over xGMI), SURVEY 8(e)
struct bhip_comm {
    bhip_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
    // created by bhip_comm_init_all: every rank of the communicator is driven by THIS process.  RCCL then requires the
    // ranks' calls of one collective to sit inside one ncclGroupStart/End (issued one after the other from one thread the
    // first would wait for peers that are never reached): such a communicator gathers through bhip_comm_allgather_group only.
    bool single_process = false;
};

static int rccl_fail(bhip_ctx *ctx, const char *what, ncclResult_t r)
{
    RcclApi &api = rccl();
    return fail(ctx, BHIP_EHIP, std::string(what) + ": " + (api.GetErrorString ? api.GetErrorString(r) : "RCCL error"));
}
#define RCCL_READY(ctx)                                                                           \
    RcclApi &api = rccl();                                                                        \
    if (!api.err.empty()) return fail(ctx, BHIP_EHIP, api.err)

int bhip_comm_unique_id(void *id, size_t bytes)
{
    if (!id || bytes < sizeof(ncclUniqueId)) return BHIP_EINVAL;
    RcclApi &api = rccl();
    if (!api.err.empty()) return BHIP_EHIP;
    ncclUniqueId u;
    if (api.GetUniqueId(&u) != ncclSuccess) return BHIP_EHIP;
    std::memset(id, 0, bytes);
    std::memcpy(id, &u, sizeof(u));
    return BHIP_OK;
}

int bhip_comm_init_rank(bhip_ctx *ctx, int nranks, int rank, const void *id, bhip_comm **out)
{
    if (!ctx || !id || !out) return BHIP_EINVAL;
    *out = nullptr;
    NEED_DEVICE(ctx);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, BHIP_EINVAL, "bhip_comm_init_rank: need 0 <= rank < nranks");
    RCCL_READY(ctx);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    const ncclResult_t r = api.CommInitRank(&c, nranks, u, rank);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclCommInitRank", r);
    bhip_comm *cm = new (std::nothrow) bhip_comm();
    if (!cm) { api.CommDestroy(c); return fail(ctx, BHIP_EHIP, "out of host memory"); }
    cm->ctx = ctx; cm->comm = c; cm->nranks = nranks; cm->rank = rank;
    ctx_retain(ctx);
    *out = cm;
    return BHIP_OK;
}

int bhip_comm_init_all(int ndev, bhip_ctx *const *ctxs, bhip_comm **comms_out)
{
    if (ndev < 1 || !ctxs || !comms_out) return BHIP_EINVAL;
    for (int k = 0; k < ndev; k++) {
        comms_out[k] = nullptr;
        if (!ctxs[k] || ctxs[k]->host_only) return BHIP_EINVAL;
    }
    bhip_ctx *ctx = ctxs[0];
    RCCL_READY(ctx);
    std::vector<int> devs(ndev);
    for (int k = 0; k < ndev; k++) {
        devs[k] = ctxs[k]->device;
        for (int j = 0; j < k; j++)
            if (devs[j] == devs[k]) return fail(ctx, BHIP_EINVAL, "bhip_comm_init_all: one context per device (RCCL refuses two ranks on one device)");
    }
    std::vector<ncclComm_t> cs(ndev, nullptr);
    const ncclResult_t r = api.CommInitAll(cs.data(), ndev, devs.data());
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclCommInitAll", r);
    for (int k = 0; k < ndev; k++) {
        bhip_comm *cm = new (std::nothrow) bhip_comm();
        if (!cm) return fail(ctx, BHIP_EHIP, "out of host memory");
        cm->ctx = ctxs[k]; cm->comm = cs[k]; cm->nranks = ndev; cm->rank = k; cm->single_process = true;
        ctx_retain(ctxs[k]);
        comms_out[k] = cm;
    }
    return BHIP_OK;
}

int bhip_comm_info(const bhip_comm *comm, int *nranks, int *rank)
{
    if (!comm) return BHIP_EINVAL;
    if (nranks) *nranks = comm->nranks;
    if (rank) *rank = comm->rank;
    return BHIP_OK;
}

int bhip_comm_query(const bhip_comm *comm, int *rccl_version, int *rccl_nranks, int *rccl_rank)
{
    if (!comm) return BHIP_EINVAL;
    bhip_ctx *ctx = comm->ctx;
    RCCL_READY(ctx);
    if (!api.GetVersion || !api.CommCount || !api.CommUserRank) return fail(ctx, BHIP_EHIP, "RCCL without ncclGetVersion / ncclCommCount / ncclCommUserRank");
    int v = 0, n = 0, r = 0;
    ncclResult_t e = api.GetVersion(&v);
    if (e == ncclSuccess) e = api.CommCount(comm->comm, &n);
    if (e == ncclSuccess) e = api.CommUserRank(comm->comm, &r);
    if (e != ncclSuccess) return rccl_fail(ctx, "ncclGetVersion / ncclCommCount / ncclCommUserRank", e);
    if (rccl_version) *rccl_version = v;
    if (rccl_nranks) *rccl_nranks = n;
    if (rccl_rank) *rccl_rank = r;
    return BHIP_OK;
}

int bhip_comm_allgather(bhip_comm *comm, const double *send_dev, double *recv_dev, size_t count)
{
    if (!comm || !send_dev || !recv_dev || count == 0) return BHIP_EINVAL;
    bhip_ctx *ctx = comm->ctx;
    NEED_DEVICE(ctx);
    if (comm->single_process && comm->nranks > 1)
        return fail(ctx, BHIP_ESTATE, "this communicator came from bhip_comm_init_all (all ranks in one process): gather through "
                                      "bhip_comm_allgather_group -- one ungrouped collective per rank from one thread would deadlock");
    RCCL_READY(ctx);
    const ncclResult_t r = api.AllGather(send_dev, recv_dev, count, ncclDouble, comm->comm, ctx->stream);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclAllGather", r);
    return BHIP_OK;
}

int bhip_comm_allgather_stats(bhip_comm *comm, const double *stats_dev, double *all_dev)
{
    return bhip_comm_allgather(comm, stats_dev, all_dev, BHIP_STATS_LEN);
}
// SURVEY 8(b)'s proposed names.  bhip_comm_init is the single-process form; with more than one device its communicators
// gather through bhip_comm_allgather_group (bhip_allgather_stats then returns BHIP_ESTATE, see bhip_comm_allgather).
int bhip_comm_init(int ndev, bhip_ctx *const *ctxs, bhip_comm **comms_out) { return bhip_comm_init_all(ndev, ctxs, comms_out); }
int bhip_allgather_stats(bhip_comm *comm, const double *stats_dev, double *all_dev) { return bhip_comm_allgather_stats(comm, stats_dev, all_dev); }

int bhip_comm_allgather_group(int n, bhip_comm *const *comms, const double *const *send_dev, double *const *recv_dev, size_t count)
{
    if (n < 1 || !comms || !send_dev || !recv_dev || count == 0) return BHIP_EINVAL;
    for (int k = 0; k < n; k++)
        if (!comms[k] || !send_dev[k] || !recv_dev[k]) return BHIP_EINVAL;
    bhip_ctx *ctx = comms[0]->ctx;
    RCCL_READY(ctx);
    ncclResult_t r = api.GroupStart();
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclGroupStart", r);
    for (int k = 0; k < n && r == ncclSuccess; k++) {
        if (hipSetDevice(comms[k]->ctx->device) != hipSuccess) { r = ncclUnhandledCudaError; break; }
        r = api.AllGather(send_dev[k], recv_dev[k], count, ncclDouble, comms[k]->comm, comms[k]->ctx->stream);
    }
    const ncclResult_t re = api.GroupEnd();
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclAllGather (group)", r);
    if (re != ncclSuccess) return rccl_fail(ctx, "ncclGroupEnd", re);
    return BHIP_OK;
}

void bhip_comm_destroy(bhip_comm *comm)
{
    if (!comm) return;
    RcclApi &api = rccl();
    bhip_ctx *ctx = comm->ctx;
    if (api.err.empty() && comm->comm) {
        (void)hipSetDevice(ctx->device);
        ctx_quiesce(ctx);
        api.CommDestroy(comm->comm);
    }
    delete comm;
    ctx_release(ctx);
}


void bhip_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    const u32x4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

void bhip_normals_host_spec(int spec, uint64_t seed, uint32_t path, uint32_t iter, int n0, int n, double *z)
{
    double z0 = 0, z1 = 0; long have = -1;
    for (int j = 0; j < n; j++) {
        const int idx = n0 + j;
        if ((idx >> 1) != have) { normal_pair_spec(spec, (uint32_t)seed, (uint32_t)(seed >> 32), path, iter, (uint32_t)(idx >> 1), z0, z1); have = idx >> 1; }
        z[j] = (idx & 1) ? z1 : z0;
    }
}

void bhip_normals_host(uint64_t seed, uint32_t path, uint32_t iter, int n0, int n, double *z)
{
    bhip_normals_host_spec(4, seed, path, iter, n0, n, z);   // the default specification
}

}  // extern "C"
