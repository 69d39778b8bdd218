/* lorenz_smoothing.c -- the adaptive smoothing loop of supplements/smoothing/smoothing.jl:60-213 through the C ABI of
 * libbridgehip.so, from plain C (no Python, no PyTorch): a Lorenz system observed with noise at the knots of m segments,
 * GuidedBridge proposals with LinearAppr auxiliaries linked backwards by gpupdate, an ensemble of chains with ONE
 * Metropolis-Hastings decision per iteration over all segments, mcnext! per chain, and the adaptation block (:130-160)
 * run for every chain at once on the device (each chain re-linearises around its OWN running mean).
 *
 *   gcc -O2 -I include examples/lorenz_smoothing.c -L bridge.jl_amd -lbridgehip -Wl,-rpath,$PWD/bridge.jl_amd -lm -o lorenz_smoothing
 *   ./lorenz_smoothing [nchains] [iterations] [adaptit]
 *
 * Prints, for the first four chains, "chain <p> acc <count> ll <sum over segments>" (hexadecimal floating point, compared
 * bit for bit with the Python mirror by tests/test_c_example.py). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "bridgehip.h"

#define CHECK(call)                                                                             \
    do {                                                                                        \
        int rc_ = (call);                                                                       \
        if (rc_ != BHIP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bhip_last_error(ctx)); return 1; } \
    } while (0)

enum { M_SEG = 3, STEPS = 40, NG = STEPS + 1, D = 3 };

static void lorenz_b(const double *th, const double *y, double *o)   /* src/Models.jl:47 */
{
    o[0] = th[0] * (y[1] - y[0]); o[1] = y[0] * (th[1] - y[2]) - y[1]; o[2] = y[0] * y[1] - th[2] * y[2];
}

int main(int argc, char **argv)
{
    const long nchains = argc > 1 ? atol(argv[1]) : 4096;
    const int iterations = argc > 2 ? atoi(argv[2]) : 12;
    const int adaptit = argc > 3 ? atoi(argv[3]) : 5;
    const double par[6] = {10.0, 20.0, 8.0 / 3, 3.0, 3.0, 3.0};           /* Lorenz(theta, sigma)   test/smoothing.jl:19-20 */
    static double tgrid[M_SEG * STEPS + 1], ref[(M_SEG * STEPS + 1) * D], obs[(M_SEG + 1) * D];
    for (int i = 0; i <= M_SEG * STEPS; i++) tgrid[i] = 0.24 * i / (M_SEG * STEPS);
    /* a reference trajectory (drift-only Euler) for the first linearisation and as the "truth" behind the observations */
    double y[D] = {1.5, -1.5, 25.0};
    for (int i = 0; i <= M_SEG * STEPS; i++) {
        for (int k = 0; k < D; k++) ref[i * D + k] = y[k];
        if (i < M_SEG * STEPS) { double b[D]; lorenz_b(par, y, b); for (int k = 0; k < D; k++) y[k] = y[k] + b[k] * (tgrid[i + 1] - tgrid[i]); }
    }
    for (int j = 0; j <= M_SEG; j++)
        for (int k = 0; k < D; k++) obs[j * D + k] = ref[j * STEPS * D + k] + 0.3 * ((j + k) % 3 - 1);   /* deterministic "noise" */
    const double L[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Sigma[9] = {0.25, 0, 0, 0, 0.25, 0, 0, 0, 0.25};

    bhip_ctx *ctx = NULL;
    if (bhip_ctx_create(0, NULL, &ctx) != BHIP_OK) { fprintf(stderr, "no HIP device: bridgehip has no CPU path\n"); return 2; }
    /* H, v = gpupdate(piH*I, 0, L, Sigma, V.yy[end])                                    smoothing.jl:75 */
    double Hprior[9] = {1e3, 0, 0, 0, 1e3, 0, 0, 0, 1e3}, vprior[3] = {0, 0, 0}, HT[9], vT[3], H[9], v[3], Hn[9], vn[3];
    CHECK(bhip_gpupdate(D, D, Hprior, vprior, L, Sigma, obs + M_SEG * D, HT, vT));
    for (int k = 0; k < 9; k++) H[k] = HT[k];
    for (int k = 0; k < 3; k++) v[k] = vT[k];
    bhip_proposal *po[M_SEG];
    static double B[NG * 9], b[NG * 3], S[NG * 9], Hd[NG * 9], V[NG * 3];
    for (int i = M_SEG - 1; i >= 0; i--) {                                               /* :77-95 */
        CHECK(bhip_proposal_create(ctx, tgrid + i * STEPS, NG, BHIP_MODEL_LORENZ, D, par, 6, &po[i]));
        const double *Y = ref + (size_t)i * STEPS * D;
        CHECK(bhip_linearappr(po[i], Y, B, b, S));                                       /* linearappr(Y, P)   src/linpro.jl:196 */
        CHECK(bhip_proposal_set_aux_linearappr(po[i], Y, B, b, S));
        CHECK(bhip_proposal_guide_hv(po[i], v, H));                                      /* GuidedBridge(tt, P, Pt, v, H) */
        CHECK(bhip_proposal_guide_get(po[i], Hd, V, NULL, NULL));
        CHECK(bhip_gpupdate(D, D, Hd, V, L, Sigma, obs + i * D, Hn, vn));                /* gpupdate(Po[i], L, Sigma, V.yy[i]) */
        for (int k = 0; k < 9; k++) H[k] = Hn[k];
        for (int k = 0; k < 3; k++) v[k] = vn[k];
    }
    /* pi0 = Gaussian(v, Hermitian(H)): the lower Cholesky factor from the upper triangle (3 x 3 closed form) */
    double C[9] = {0};
    C[0] = sqrt(H[0]); C[1] = H[3] / C[0]; C[4] = sqrt(H[4] - C[1] * C[1]);
    C[2] = H[6] / C[0]; C[5] = (H[7] - C[1] * C[2]) / C[4]; C[8] = sqrt(H[8] - C[2] * C[2] - C[5] * C[5]);

    bhip_segchains *sc = NULL;
    CHECK(bhip_segchains_create(ctx, M_SEG, (const bhip_proposal *const *)po, nchains, 0, 7, BHIP_SEGCHAINS_MCNEXT, &sc));
    CHECK(bhip_segchains_init(sc, v, C, 0));
    const double w_new = sqrt(0.1), w_old = sqrt(0.9);
    for (int it = 1; it <= iterations; it++) {
        if (adaptit > 0 && it % adaptit == 0)                                            /* :130-160, every chain, on the device */
            CHECK(bhip_segchains_adapt_device(sc, D, L, Sigma, obs, HT, vT, 0, BHIP_SEG_NEWBLOCK | (it == adaptit ? BHIP_SEG_DOACCEPT : 0)));
        CHECK(bhip_segchains_step(sc, &w_old, &w_new, 1));
    }
    double *ll = malloc(sizeof(double) * M_SEG * nchains);
    int64_t *acc = malloc(sizeof(int64_t) * nchains);
    CHECK(bhip_segchains_get(sc, ll, acc, NULL));
    double accsum = 0;
    for (long p = 0; p < nchains; p++) accsum += (double)acc[p];
    printf("chains %ld iterations %d acceptance %.4f\n", nchains, iterations, accsum / ((double)nchains * iterations));
    for (long p = 0; p < 4 && p < nchains; p++) {
        double s = 0;
        for (int i = 0; i < M_SEG; i++) s += ll[(size_t)i * nchains + p];
        printf("chain %ld acc %lld ll %a\n", p, (long long)acc[p], s);
    }
    free(ll); free(acc);
    /* any order: the context is reference counted by its children */
    bhip_ctx_destroy(ctx);
    bhip_segchains_destroy(sc);
    for (int i = 0; i < M_SEG; i++) bhip_proposal_destroy(po[i]);
    return 0;
}
