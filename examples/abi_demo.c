/* abi_demo.c -- the C ABI of include/kabc.h used from plain C99, no Python, no torch:
 * what a Julia `ccall`, a cgo or a JNI binding would do.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/abi_demo.c -o /tmp/abi_demo \
 *       -Lkissabc.jl_amd/lib -lkabc_hip -lpthread -Wl,-rpath,$PWD/kissabc.jl_amd/lib
 *   /tmp/abi_demo
 *
 * 1. sample(ApproxKernelizedPosterior(Factored(Normal(0,5), Normal(0,5)), cost, 0.1),
 *           AIS(4096), 2000 generations after 500 discarded; ntransitions = 1)
 *    with cost = ||x - (1, -0.5)||: the posterior is Gaussian with mean c * 2500/2501
 *     -- prints the sample mean.
 * 2. smc(prior, cost; nparticles = 2000, alpha = 0.9, epstol = 0.05) on the same cost.
 * 3. the same AIS ensemble sharded over two ranks driven from this one process
 *    (kabc_comm_init_all + kabc_ais_create_dist + kabc_ais_*_multi).
 * 4. the smc of step 2 with its PARTICLES sharded over two ranks, one host thread per rank
 *    (kabc_smc_run_dist_mode, KABC_SMC_DIST_PARTICLES): every rank returns step 2's result.
 * Prints one line of key=value pairs; tests/test_gpu_abi_demo.py compares it with the
 * Python mirror's result for the same seeds (bit-identical). */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kabc.h"
#include "kabc_costs.h" /* cost ids */

#define CHECK(call)                                                          \
    do {                                                                     \
        kabc_status_t st_ = (call);                                          \
        if (st_ != KABC_OK) {                                                \
            fprintf(stderr, "%s failed (%d): %s\n", #call, (int)st_, kabc_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

/* one rank of step 4: a collective call, every rank with the same arguments */
typedef struct rank_job {
    kabc_comm_t* comm;
    const kabc_prior_t* prior;
    const kabc_cost_t* cost;
    const kabc_smc_opts_t* opts;
    kabc_smc_result_t res;
    kabc_status_t status;
} rank_job_t;

static void* rank_main(void* arg) {
    rank_job_t* j = (rank_job_t*)arg;
    j->status = kabc_smc_run_dist_mode(j->comm, j->prior, 2, j->cost, j->opts, KABC_SMC_DIST_PARTICLES, &j->res);
    return NULL;
}

int main(void) {
    if (kabc_device_count() < 1) {
        fprintf(stderr, "no gfx950 device\n");
        return 2;
    }
    kabc_ctx_t* ctx = NULL;
    CHECK(kabc_ctx_create(0, NULL, &ctx));

    kabc_prior_t prior[2];
    memset(prior, 0, sizeof prior);
    prior[0].kind = prior[1].kind = KABC_PRIOR_NORMAL;
    prior[0].p[0] = prior[1].p[0] = 0.0;
    prior[0].p[1] = prior[1].p[1] = 5.0;
    const double centre[2] = {1.0, -0.5};
    kabc_model_t model;
    memset(&model, 0, sizeof model);
    model.prior = prior;
    model.D = 2;
    model.posterior = KABC_POSTERIOR_KERNELIZED;
    model.eps = 0.1;
    model.cost.id = KABC_COST_GAUSS_DIST;
    model.cost.nparams = 2;
    model.cost.params = centre;

    /* ---- AIS ---- */
    const int64_t N = 4096, keep = 200;
    kabc_ais_t* ais = NULL;
    CHECK(kabc_ais_create(ctx, &model, N, 7u, &ais));
    CHECK(kabc_ais_init(ais, 100));
    kabc_stats_t stats;
    memset(&stats, 0, sizeof stats);
    CHECK(kabc_ais_advance(ais, 300, 1, NULL, &stats)); /* discard_initial */
    double* trace = NULL;
    CHECK(kabc_host_alloc(sizeof(double) * (size_t)(keep * N * 2), (void**)&trace));
    CHECK(kabc_ais_advance(ais, keep, 1, trace, &stats));
    double m0 = 0.0, m1 = 0.0;
    for (int64_t i = 0; i < keep * N; ++i) {
        m0 += trace[2 * i];
        m1 += trace[2 * i + 1];
    }
    m0 /= (double)(keep * N);
    m1 /= (double)(keep * N);
    const double last0 = trace[2 * (keep * N - 1)], last1 = trace[2 * (keep * N - 1) + 1];
    CHECK(kabc_host_free(trace));
    CHECK(kabc_ais_destroy(ais));

    /* ---- smc ---- */
    kabc_smc_opts_t o;
    kabc_smc_default_opts(&o);
    o.nparticles = 2000;
    o.alpha = 0.9;
    o.epstol = 0.05;
    o.seed = 11u;
    kabc_smc_result_t r;
    memset(&r, 0, sizeof r);
    r.theta = (double*)malloc(sizeof(double) * 2000 * 2);
    r.cost = (double*)malloc(sizeof(double) * 2000);
    r.alive = (uint8_t*)malloc(2000);
    CHECK(kabc_smc_run(ctx, prior, 2, &model.cost, &o, &r));
    double s0 = 0.0;
    for (int i = 0; i < 2000; ++i) s0 += r.theta[2 * i];

    /* ---- walker-sharded AIS: two ranks in ONE process (here both on device 0, exchange by
     * the P2P pull kernel; with dev = {0, 1} and KABC_COMM_RCCL the same calls drive two
     * GPUs over RCCL).  The sharded trajectory equals the single-handle one bit for bit. */
    int sharded_equal = 0;
    {
        const int32_t devs[2] = {0, 0};
        kabc_ctx_t* cx[2];
        kabc_comm_t* cm[2];
        kabc_ais_t* sh[2];
        kabc_ais_t* one = NULL;
        kabc_stats_t st2;
        double* xa = (double*)malloc(sizeof(double) * (size_t)N * 2);
        double* xb = (double*)malloc(sizeof(double) * (size_t)N * 2);
        memset(&st2, 0, sizeof st2);
        CHECK(kabc_comm_init_all(2, devs, KABC_COMM_P2P, cx, cm));
        for (int rk = 0; rk < 2; ++rk) CHECK(kabc_ais_create_dist(cm[rk], &model, N, 7u, &sh[rk]));
        CHECK(kabc_ais_init_multi(sh, 2, 100));
        CHECK(kabc_ais_advance_multi(sh, 2, 50, 3, &st2));
        CHECK(kabc_ais_get_ensemble(sh[1], xa));
        CHECK(kabc_ais_create(ctx, &model, N, 7u, &one));
        CHECK(kabc_ais_init(one, 100));
        CHECK(kabc_ais_advance(one, 50, 3, NULL, NULL));
        CHECK(kabc_ais_get_ensemble(one, xb));
        sharded_equal = memcmp(xa, xb, sizeof(double) * (size_t)N * 2) == 0 &&
                        st2.proposals == (uint64_t)N * 150u;
        CHECK(kabc_ais_destroy(one));
        for (int rk = 0; rk < 2; ++rk) CHECK(kabc_ais_destroy(sh[rk]));
        for (int rk = 0; rk < 2; ++rk) CHECK(kabc_comm_destroy(cm[rk]));
        free(xa);
        free(xb);
    }

    /* ---- smc with sharded particles: two ranks, a host thread each (with one process per GPU and
     * kabc_comm_init_rank every process makes the one call itself) */
    int smc_sharded_equal = 0;
    {
        const int32_t devs[2] = {0, 0};
        kabc_ctx_t* cx[2];
        kabc_comm_t* cm[2];
        rank_job_t job[2];
        pthread_t th[2];
        CHECK(kabc_comm_init_all(2, devs, KABC_COMM_P2P, cx, cm));
        memset(job, 0, sizeof job);
        for (int rk = 0; rk < 2; ++rk) {
            job[rk].comm = cm[rk];
            job[rk].prior = prior;
            job[rk].cost = &model.cost;
            job[rk].opts = &o;
            job[rk].res.theta = (double*)malloc(sizeof(double) * 2000 * 2);
            job[rk].res.cost = (double*)malloc(sizeof(double) * 2000);
            job[rk].res.alive = (uint8_t*)malloc(2000);
            if (pthread_create(&th[rk], NULL, rank_main, &job[rk]) != 0) return 3;
        }
        smc_sharded_equal = 1;
        for (int rk = 0; rk < 2; ++rk) {
            pthread_join(th[rk], NULL);
            smc_sharded_equal = smc_sharded_equal && job[rk].status == KABC_OK && job[rk].res.eps == r.eps &&
                                job[rk].res.iterations == r.iterations &&
                                memcmp(job[rk].res.theta, r.theta, sizeof(double) * 2000 * 2) == 0 &&
                                memcmp(job[rk].res.cost, r.cost, sizeof(double) * 2000) == 0 &&
                                memcmp(job[rk].res.alive, r.alive, 2000) == 0;
            free(job[rk].res.theta);
            free(job[rk].res.cost);
            free(job[rk].res.alive);
        }
        for (int rk = 0; rk < 2; ++rk) CHECK(kabc_comm_destroy(cm[rk]));
    }

    printf("version=%d sharded_equal=%d smc_sharded_equal=%d proposals=%llu accepted=%llu mean0=%.17g mean1=%.17g last0=%.17g last1=%.17g "
           "smc_eps=%.17g smc_iterations=%lld smc_alive=%lld smc_sum0=%.17g\n",
           (int)kabc_version(), sharded_equal, smc_sharded_equal, (unsigned long long)stats.proposals,
           (unsigned long long)stats.accepted, m0, m1, last0, last1, r.eps,
           (long long)r.iterations, (long long)r.n_alive, s0);
    free(r.theta);
    free(r.cost);
    free(r.alive);
    CHECK(kabc_ctx_destroy(ctx));
    return 0;
}
