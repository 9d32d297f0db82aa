/* A plain C host of the drop-in symbol: compiled with gcc against include/FastClusterWrapper.h and include/fluidaudio_hip.h
 * and linked to libfluidaudio_hip.so the way the reference's SwiftPM target links its FastClusterWrapper.
 * Mode "args": only the argument contract (FastClusterWrapper.cpp:203-226), needs no GPU.
 * Mode "cluster <file> [fault-site]": the whole clustering stage through fa_offline_cluster on a session written by the test (int64 n,
 *   int32 d, int32 rho_dim, float emb[n*d], double rho[n*rho_dim], int32 chunk[n], double phi[rho_dim]); prints one label per line.
 *   With a fault site (fa_debug_inject_fault) the stage must degrade the way the reference does (VBxClustering.swift:136-141,
 *   AHCClustering.swift:52-55) and still return SUCCESS.
 * Mode "pool": the device set (fa_pool_*): a pool over device 0 listed twice; fa_mel_batch_sharded must write exactly what fa_mel_batch
 *   writes, and 6 pthreads calling the context-free drop-in symbol at once must each get the dendrogram of a lone call.
 * Mode "run": a tie-free variant of the 6-point orthogonal-groups case probed on the reference build in SURVEY.md §8(c); prints the dendrogram. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "FastClusterWrapper.h"
#include "fluidaudio_hip.h"

struct job { const double *x; size_t n, d; double *z; int status; };
static void *job_main(void *p) {
    struct job *j = p;
    j->status = (int)fastcluster_compute_centroid_linkage(j->x, j->n, j->d, j->z, (j->n - 1) * 4);
    return NULL;
}
static double lcg(unsigned long long *s) { *s = *s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(*s >> 11) / 9007199254740992.0 - 0.5; }

static int pool_mode(void) {
    int bad = 0;
    /* (1) sharded mel == unsharded mel, byte for byte */
    const int devs[2] = {0, 0};
    fa_pool *pool = NULL;
    if (fa_pool_create(devs, 2, &pool) != FA_SUCCESS || fa_pool_size(pool) != 2) return 80;
    enum { B = 7 };
    const long long lens[B] = {16000, 3000, 48000, 0, 24000, 160, 32000};
    long long off[B + 1] = {0};
    for (int b = 0; b < B; ++b) off[b + 1] = off[b] + lens[b];
    float *pcm = malloc(sizeof(float) * off[B]);
    unsigned long long seed = 7;
    for (long long i = 0; i < off[B]; ++i) pcm[i] = (float)(0.2 * lcg(&seed));
    fa_mel_config cfg;
    fa_mel_default_config(&cfg);
    const int fs = fa_mel_padded_frames(&cfg, fa_mel_num_frames(&cfg, 48000));
    const size_t total = (size_t)B * cfg.n_mels * fs;
    float *a = calloc(total, sizeof(float)), *b2 = calloc(total, sizeof(float));
    int la[B], lb[B];
    fa_ctx *ctx = fa_pool_context(pool, 0);
    bad += fa_mel_batch(ctx, &cfg, pcm, (const int64_t *)off, B, NULL, NULL, 0, a, la) != FA_SUCCESS;
    bad += fa_mel_batch_sharded(pool, &cfg, pcm, (const int64_t *)off, B, NULL, NULL, 0, b2, lb) != FA_SUCCESS;
    bad += memcmp(a, b2, total * sizeof(float)) != 0;
    bad += memcmp(la, lb, sizeof(la)) != 0;
    fa_pool_destroy(pool);
    /* (2) concurrent callers of the drop-in symbol (default pool: FLUIDAUDIO_HIP_DEVICES) */
    enum { K = 6 };
    struct job jobs[K], alone[K];
    pthread_t th[K];
    for (int k = 0; k < K; ++k) {
        const size_t n = 200 + 31 * k, d = 24;
        double *x = malloc(sizeof(double) * n * d);
        for (size_t i = 0; i < n * d; ++i) x[i] = lcg(&seed);
        jobs[k] = (struct job){x, n, d, calloc((n - 1) * 4, sizeof(double)), -1};
        alone[k] = (struct job){x, n, d, calloc((n - 1) * 4, sizeof(double)), -1};
        job_main(&alone[k]);
    }
    for (int k = 0; k < K; ++k) pthread_create(&th[k], NULL, job_main, &jobs[k]);
    for (int k = 0; k < K; ++k) pthread_join(th[k], NULL);
    for (int k = 0; k < K; ++k)
        bad += jobs[k].status != 0 || alone[k].status != 0 || memcmp(jobs[k].z, alone[k].z, sizeof(double) * (jobs[k].n - 1) * 4) != 0;
    printf("pool mode: mismatches %d\n", bad);
    return bad;
}

int main(int argc, char **argv) {
    double z[20] = {0};
    if (argc > 1 && strcmp(argv[1], "pool") == 0) return pool_mode();
    if (argc > 1 && strcmp(argv[1], "args") == 0) {
        double x[6] = {1, 0, 0, 1, 1, 1};
        int bad = 0;
        bad += fastcluster_compute_centroid_linkage(NULL, 3, 2, z, 8) != FASTCLUSTER_WRAPPER_INVALID_ARGUMENT;
        bad += fastcluster_compute_centroid_linkage(x, 3, 2, NULL, 8) != FASTCLUSTER_WRAPPER_INVALID_ARGUMENT;
        bad += fastcluster_compute_centroid_linkage(x, 0, 2, z, 8) != FASTCLUSTER_WRAPPER_SUCCESS;
        bad += fastcluster_compute_centroid_linkage(x, 3, 0, z, 8) != FASTCLUSTER_WRAPPER_INVALID_ARGUMENT;
        bad += fastcluster_compute_centroid_linkage(x, 3, 2, z, 7) != FASTCLUSTER_WRAPPER_OUTPUT_TOO_SMALL;
        bad += fastcluster_compute_centroid_linkage(x, 1, 2, z, 0) != FASTCLUSTER_WRAPPER_SUCCESS;
        printf("version %s; argument contract violations: %d\n", fa_version(), bad);
        return bad;
    }
    if (argc > 2 && strcmp(argv[1], "cluster") == 0) {
        FILE *f = fopen(argv[2], "rb");
        long long n = 0;
        int d = 0, rd = 0;
        if (!f || fread(&n, 8, 1, f) != 1 || fread(&d, 4, 1, f) != 1 || fread(&rd, 4, 1, f) != 1) return 90;
        float *emb = malloc(sizeof(float) * n * d);
        double *rho = malloc(sizeof(double) * n * rd), *phi = malloc(sizeof(double) * rd), *cen = malloc(sizeof(double) * (64 < n ? n : 64) * d);
        int *chunk = malloc(sizeof(int) * n), *labels = malloc(sizeof(int) * n);
        if (fread(emb, sizeof(float), n * d, f) != (size_t)(n * d) || fread(rho, sizeof(double), n * rd, f) != (size_t)(n * rd) ||
            fread(chunk, sizeof(int), n, f) != (size_t)n || fread(phi, sizeof(double), rd, f) != (size_t)rd) return 91;
        fclose(f);
        fa_ctx *ctx = NULL;
        if (fa_ctx_create(0, NULL, &ctx) != FA_SUCCESS) return 92;
        fa_offline_cluster_config cfg;
        fa_offline_cluster_default_config(&cfg);
        fa_offline_cluster_info info;
        int k = 0;
        if (argc > 3) fa_debug_inject_fault(atoi(argv[3]), 1);
        const fa_status st = fa_offline_cluster(ctx, emb, n, d, rho, rd, chunk, phi, &cfg, 0, labels, cen, 64 < n ? (int)n : 64, &k, &info);
        printf("status %d clusters %d training %lld initial %d vbx_iterations %d constrained %d vbx_degraded %d ahc_degraded %d\n", (int)st, k,
               (long long)info.training_rows, info.initial_clusters, info.vbx_iterations, info.constrained, info.vbx_degraded, info.ahc_degraded);
        for (long long i = 0; i < n; ++i) printf("%d\n", labels[i]);
        fa_ctx_destroy(ctx);
        return st;
    }
    /* two orthogonal groups: (0, 2) and (3, 5) nearly parallel, 1 and 4 a little further; all distances distinct */
    const double x[6 * 3] = {1.00, 0.00, 0.0,   0.99, 0.00, 0.141067,   0.998614, 0.052631, 0.0,
                             0.00, 1.00, 0.0,   0.00, 0.985, 0.172,      0.061, 0.998138, 0.0};
    const fastcluster_wrapper_status st = fastcluster_compute_centroid_linkage(x, 6, 3, z, 20);
    printf("status %d\n", (int)st);
    for (int r = 0; r < 5; ++r) printf("%.0f %.0f %.17g %.0f\n", z[4 * r], z[4 * r + 1], z[4 * r + 2], z[4 * r + 3]);
    return st;
}
