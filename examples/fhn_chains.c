/* fhn_chains.c -- the C ABI of libbridgehip.so used directly from C (no Python, no PyTorch):
 * the MCMC loop of project_partialbridge/partialbridge_fitzhugh.jl:121-176 for an ensemble of chains.
 *
 *   gcc -O2 -I include examples/fhn_chains.c -L bridge.jl_amd -lbridgehip -Wl,-rpath,$PWD/bridge.jl_amd -lm -o fhn_chains
 *   ./fhn_chains [nchains] [iterations]
 *
 * Prints the ensemble statistics and, for the first four chains, "chain <p> acc <count> ll <value>" with the value in
 * hexadecimal floating point so that tests/test_c_example.py can compare it bit for bit with the Python mirror. */
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

int main(int argc, char **argv)
{
    const long nchains = argc > 1 ? atol(argv[1]) : 4096;
    const int iterations = argc > 2 ? atoi(argv[2]) : 20;
    enum { N = 1001 };
    const double T = 2.0, v = 1.1, rho = 0.9;
    /* FitzhughDiffusion(0.1, 0.0, 1.5, 0.8, 0.3), x0 = (-0.5, -0.6)            partialbridge_fitzhugh.jl:48-50 */
    const double par[5] = {0.1, 0.0, 1.5, 0.8, 0.3}, x0[2] = {-0.5, -0.6};
    const double eps = par[0], s = par[1], gam = par[2], bet = par[3], sig = par[4];
    /* auxiliary "linearised_end": B = [1/eps - 3v^2/eps, -1/eps; gamma, -1], beta = (s/eps + 2v^3/eps, beta)   :99-100
     * as AFFINE parameters: B (column-major), beta, sigma (d x m') */
    const double apar[8] = {1 / eps - 3 * v * v / eps, gam, -1 / eps, -1.0, s / eps + 2 * v * v * v / eps, bet, 0.0, sig};
    const double L[2] = {1.0, 0.0}, vobs[1] = {v}, Sigma[1] = {1e-10};
    static double tt[N];
    const double step = T / (N - 1);
    for (int i = 0; i < N; i++) {   /* tau(s) = s(2 - s/T) on s = i*step (the reference's range 0:dt:T)  :11-14 */
        const double u = i == N - 1 ? T : i * step;
        tt[i] = u * (2 - u / T);
    }

    bhip_ctx *ctx = NULL;
    if (bhip_ctx_create(0, NULL, &ctx) != BHIP_OK) { fprintf(stderr, "no HIP device: bridgehip has no CPU path\n"); return 2; }
    bhip_proposal *po = NULL;
    CHECK(bhip_proposal_create(ctx, tt, N, BHIP_MODEL_FHN, 2, par, 5, &po));
    CHECK(bhip_proposal_set_aux(po, BHIP_AUX_AFFINE, apar, 8));
    CHECK(bhip_proposal_guide_lmmu(po, 1, L, vobs, Sigma));            /* PartialBridge(tt, P, Pt, L, v, Sigma)  :118 */

    bhip_chains *ch = NULL;
    CHECK(bhip_chains_create(ctx, po, nchains, 0, 44, 0, &ch));
    CHECK(bhip_chains_init(ch, x0, 0));
    CHECK(bhip_chains_step(ch, rho, iterations, 0));

    double *stats_dev = NULL, stats[BHIP_STATS_LEN];
    CHECK(bhip_malloc(ctx, sizeof(stats), (void **)&stats_dev));
    CHECK(bhip_chains_stats(ch, stats_dev));
    CHECK(bhip_memcpy_d2h(ctx, stats, stats_dev, sizeof(stats)));
    printf("chains %.0f iterations %.0f acceptance %.4f mean ll %.6f\n", stats[0], stats[1], stats[2] / (stats[0] * stats[1]), stats[3] / stats[0]);

    double *ll = malloc(sizeof(double) * nchains);
    int64_t *acc = malloc(sizeof(int64_t) * nchains);
    CHECK(bhip_chains_get(ch, ll, acc));
    for (long p = 0; p < 4 && p < nchains; p++) printf("chain %ld acc %lld ll %a\n", p, (long long)acc[p], ll[p]);
    /* the current path of chain 0 ends at the observation */
    static double X[N * 2];
    CHECK(bhip_chains_get_paths(ch, 0, 1, X, NULL));
    printf("chain 0 endpoint x1 %.6f (observed %.1f)\n", X[(N - 1) * 2], v);

    free(ll); free(acc);
    CHECK(bhip_free(ctx, stats_dev));
    bhip_chains_destroy(ch);
    bhip_proposal_destroy(po);
    bhip_ctx_destroy(ctx);
    return 0;
}
