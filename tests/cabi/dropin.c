/* A plain C host of the drop-in symbol: compiled with gcc against include/FastClusterWrapper.h and include/fluidaudio_hip.h
 * and linked to libfluidaudio_hip.so the way the reference's SwiftPM target links its FastClusterWrapper.
 * Mode "args": only the argument contract (FastClusterWrapper.cpp:203-226), needs no GPU.
 * Mode "cluster <file>": the whole clustering stage through fa_offline_cluster on a session written by the test (int64 n, int32 d,
 *   int32 rho_dim, float emb[n*d], double rho[n*rho_dim], int32 chunk[n], double phi[rho_dim]); prints one label per line.
 * Mode "run": a tie-free variant of the 6-point orthogonal-groups case probed on the reference build in SURVEY.md §8(c); prints the dendrogram. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "FastClusterWrapper.h"
#include "fluidaudio_hip.h"

int main(int argc, char **argv) {
    double z[20] = {0};
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
        double *rho = malloc(sizeof(double) * n * rd), *phi = malloc(sizeof(double) * rd), *cen = malloc(sizeof(double) * 64 * d);
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
        const fa_status st = fa_offline_cluster(ctx, emb, n, d, rho, rd, chunk, phi, &cfg, 0, labels, cen, 64, &k, &info);
        printf("status %d clusters %d training %lld initial %d vbx_iterations %d constrained %d\n", (int)st, k, (long long)info.training_rows,
               info.initial_clusters, info.vbx_iterations, info.constrained);
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
