/* fhn_chains_multi.c -- the multi-GPU form of examples/fhn_chains.c through the C ABI alone (no Python, no PyTorch, no
 * launcher): ONE single-threaded process -- what a Julia `ccall` host is -- drives every visible GPU.
 *
 * The unit that is sharded is one chain of project_partialbridge/partialbridge_fitzhugh.jl:143-176: device k owns the
 * contiguous global chain ids [k*n, (k+1)*n) (path0 = k*n keys the Philox counter by the GLOBAL id, so the chains do not
 * depend on the number of devices), grid / model / guide are replicated (each context integrates the same guide ODE on the
 * host), there is no data-path exchange, and the ONE communication is the RCCL all-gather of the 64-byte statistics
 * block: bhip_comm_init_all (one context per device) + bhip_comm_allgather_group.  Kernel launches are asynchronous, so
 * the single host thread keeps all devices busy: it issues iteration i on every device before iteration i+1.
 *
 *   gcc -O2 -I include examples/fhn_chains_multi.c -L bridge.jl_amd -lbridgehip -Wl,-rpath,$PWD/bridge.jl_amd -lm -o fhn_chains_multi
 *   ./fhn_chains_multi [chains per device] [iterations] [devices (0 = all visible)]
 *
 * Prints "devices <n>", the ensemble statistics combined from the gathered blocks as every device received them, and
 * "chain <global id> acc <count> ll <hex>" for the first two chains of every device (tests/test_c_example.py compares them
 * bit for bit with ONE ensemble of n*chains chains on a single device). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bridgehip.h"

#define MAXDEV 16
#define CHECK(ctx, call)                                                                        \
    do {                                                                                        \
        int rc_ = (call);                                                                       \
        if (rc_ != BHIP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bhip_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    const long nchains = argc > 1 ? atol(argv[1]) : 4096;
    const int iterations = argc > 2 ? atoi(argv[2]) : 20;
    int ndev = argc > 3 ? atoi(argv[3]) : 0;
    enum { N = 1001 };
    const double T = 2.0, v = 1.1, rho = 0.9;
    const double par[5] = {0.1, 0.0, 1.5, 0.8, 0.3}, x0[2] = {-0.5, -0.6};
    const double eps = par[0], s = par[1], gam = par[2], bet = par[3], sig = par[4];
    const double apar[8] = {1 / eps - 3 * v * v / eps, gam, -1 / eps, -1.0, s / eps + 2 * v * v * v / eps, bet, 0.0, sig};
    const double L[2] = {1.0, 0.0}, vobs[1] = {v}, Sigma[1] = {1e-10};
    static double tt[N];
    const double step = T / (N - 1);
    for (int i = 0; i < N; i++) {
        const double u = i == N - 1 ? T : i * step;
        tt[i] = u * (2 - u / T);
    }

    const int visible = bhip_device_count();
    if (visible < 1) { fprintf(stderr, "no HIP device: bridgehip has no CPU path\n"); return 2; }
    if (ndev <= 0) ndev = visible;
    if (ndev > visible || ndev > MAXDEV) { fprintf(stderr, "asked for %d devices, %d visible (at most %d here)\n", ndev, visible, MAXDEV); return 2; }

    bhip_ctx *ctx[MAXDEV] = {0};
    bhip_proposal *po[MAXDEV] = {0};
    bhip_chains *ch[MAXDEV] = {0};
    bhip_comm *comm[MAXDEV] = {0};
    double *stats_dev[MAXDEV] = {0}, *all_dev[MAXDEV] = {0};
    for (int k = 0; k < ndev; k++) {
        if (bhip_ctx_create(k, NULL, &ctx[k]) != BHIP_OK) { fprintf(stderr, "bhip_ctx_create(%d) failed\n", k); return 2; }
        CHECK(ctx[k], bhip_proposal_create(ctx[k], tt, N, BHIP_MODEL_FHN, 2, par, 5, &po[k]));
        CHECK(ctx[k], bhip_proposal_set_aux(po[k], BHIP_AUX_AFFINE, apar, 8));
        CHECK(ctx[k], bhip_proposal_guide_lmmu(po[k], 1, L, vobs, Sigma));
        /* the shard: global chain ids [k*nchains, (k+1)*nchains) */
        CHECK(ctx[k], bhip_chains_create(ctx[k], po[k], nchains, (uint32_t)(k * nchains), 44, 0, &ch[k]));
        CHECK(ctx[k], bhip_chains_init(ch[k], x0, 0));
        CHECK(ctx[k], bhip_malloc(ctx[k], sizeof(double) * BHIP_STATS_LEN, (void **)&stats_dev[k]));
        CHECK(ctx[k], bhip_malloc(ctx[k], sizeof(double) * BHIP_STATS_LEN * ndev, (void **)&all_dev[k]));
    }
    CHECK(ctx[0], bhip_comm_init_all(ndev, ctx, comm));

    /* ONE call steps every device: iteration by iteration the launches go out round-robin, each on its context's stream
     * (asynchronous), and one more reduces every device's statistics -- two crossings of the ABI for the whole loop */
    CHECK(ctx[0], bhip_chains_step_group(ndev, ch, rho, iterations, 0));
    CHECK(ctx[0], bhip_chains_stats_group(ndev, ch, stats_dev));
    /* the one collective: every device receives every device's block */
    CHECK(ctx[0], bhip_comm_allgather_group(ndev, comm, (const double *const *)stats_dev, all_dev, BHIP_STATS_LEN));
    /* the ungrouped per-communicator call is refused on a multi-rank single-process communicator (it would deadlock) */
    if (ndev > 1 && bhip_comm_allgather_stats(comm[0], stats_dev[0], all_dev[0]) != BHIP_ESTATE) { fprintf(stderr, "ungrouped gather was not refused\n"); return 1; }
    for (int k = 0; k < ndev; k++) CHECK(ctx[k], bhip_ctx_sync(ctx[k]));

    printf("devices %d\n", ndev);
    double ref[MAXDEV * BHIP_STATS_LEN];
    for (int k = 0; k < ndev; k++) {
        double got[MAXDEV * BHIP_STATS_LEN];
        CHECK(ctx[k], bhip_memcpy_d2h(ctx[k], got, all_dev[k], sizeof(double) * BHIP_STATS_LEN * ndev));
        if (k == 0) memcpy(ref, got, sizeof(double) * BHIP_STATS_LEN * ndev);
        else if (memcmp(ref, got, sizeof(double) * BHIP_STATS_LEN * ndev) != 0) { fprintf(stderr, "device %d received different blocks\n", k); return 1; }
    }
    double n = 0, acc = 0, sll = 0, iters = 0;
    for (int k = 0; k < ndev; k++) {
        const double *b = ref + k * BHIP_STATS_LEN;   /* {n, iterations, sum acc, sum ll, sum ll^2, min ll, max ll, sum acc^2} */
        n += b[0]; iters = b[1] > iters ? b[1] : iters; acc += b[2]; sll += b[3];
    }
    printf("chains %.0f iterations %.0f acceptance %.6f mean ll %.9f\n", n, iters, acc / (n * iters), sll / n);
    double *ll = malloc(sizeof(double) * nchains);
    int64_t *ac = malloc(sizeof(int64_t) * nchains);
    for (int k = 0; k < ndev; k++) {
        CHECK(ctx[k], bhip_chains_get(ch[k], ll, ac));
        for (long p = 0; p < 2 && p < nchains; p++) printf("chain %ld acc %lld ll %a\n", k * nchains + p, (long long)ac[p], ll[p]);
    }
    free(ll); free(ac);
    for (int k = 0; k < ndev; k++) {
        bhip_comm_destroy(comm[k]);
        CHECK(ctx[k], bhip_free(ctx[k], stats_dev[k]));
        CHECK(ctx[k], bhip_free(ctx[k], all_dev[k]));
        bhip_chains_destroy(ch[k]);
        bhip_proposal_destroy(po[k]);
        bhip_ctx_destroy(ctx[k]);
    }
    return 0;
}
