/*
 * ahc_model.c — CPU MODEL of the GPU merge-round algorithm (test infrastructure only).
 *
 * Mirrors, step for step, the round structure of fluidaudio_amd/csrc/ahc.hip
 * (slot matrix M, per-row minima with lazy rescans, optional Lance-Williams filter with
 * exact re-verification) so that the algorithm itself can be checked against the
 * reference build (oracle/_ref) on the CPU, and so that round counts can be measured
 * before any GPU time is spent.  It is not an oracle (it restates nothing from the
 * reference) and nothing in the product path uses it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    long merges, rescans, rounds, ambiguous, exact_evals;
} ahc_model_stats;

static double sqdist_cols(const double *xt, size_t np, size_t d, size_t a, size_t b) {
    double s = 0.0;
    for (size_t k = 0; k < d; ++k) { const double diff = xt[k * np + a] - xt[k * np + b]; s += diff * diff; }
    return s;
}

/* mode 0: exact rows (every matrix entry is the reference's sequential fp64 sum)
 * mode 1: Lance-Williams rows + exact verification of the selected pair; when several pairs lie within
 *         2*eps of the minimum (|S_eps| != one mutual pair) a WINDOW round re-evaluates every such matrix
 *         entry exactly.  eps = eps_scale * N * u * dmax (the library uses eps_scale = 16). */
int ahc_model_linkage(const double *data, size_t n, size_t d, double *z, int mode, double eps_scale,
                      ahc_model_stats *st) {
    memset(st, 0, sizeof(*st));
    if (n < 2) return 0;
    const size_t np = n;
    double *xt = (double *)malloc(sizeof(double) * d * np);
    double *M = (double *)malloc(sizeof(double) * np * np);
    double *rowmin = (double *)malloc(sizeof(double) * np);
    long *rownn = (long *)malloc(sizeof(long) * np);
    char *valid = (char *)malloc(np), *active = (char *)malloc(np);
    double *size = (double *)malloc(sizeof(double) * np);
    long *node = (long *)malloc(sizeof(long) * np);
    if (!xt || !M || !rowmin || !rownn || !valid || !active || !size || !node) return 4;
    for (size_t i = 0; i < n; ++i)
        for (size_t k = 0; k < d; ++k) xt[k * np + i] = data[i * d + k];
    double dmax = 0.0;
    for (size_t i = 0; i < n; ++i) {
        for (size_t j = 0; j < n; ++j) {
            double v = i == j ? INFINITY : sqdist_cols(xt, np, d, i, j);
            M[i * np + j] = v;
            if (i != j && v > dmax) dmax = v;
        }
    }
    for (size_t i = 0; i < n; ++i) {
        double mv = INFINITY; long mi = -1;
        for (size_t j = 0; j < n; ++j) if (M[i * np + j] < mv) { mv = M[i * np + j]; mi = (long)j; }
        rowmin[i] = mv; rownn[i] = mi; valid[i] = 1; active[i] = 1; size[i] = 1.0; node[i] = (long)i;
    }
    const double eps = mode == 1 ? eps_scale * (double)n * 1.1102230246251565e-16 * dmax : 0.0;
    size_t step = 0;
    while (step + 1 < n) {
        st->rounds++;
        /* K1: global min over row minima (stale rows contribute their lower bound) */
        double v = INFINITY;
        for (size_t i = 0; i < n; ++i) if (active[i] && rowmin[i] < v) v = rowmin[i];
        const double lim = v + 2.0 * eps;
        long stale = -1, cnt = 0, r = -1;
        for (size_t i = 0; i < n; ++i) {
            if (!active[i] || !(rowmin[i] <= lim)) continue;
            if (!valid[i]) { if (stale < 0) stale = (long)i; continue; }
            ++cnt;
            if (r < 0 && rowmin[i] == v) r = (long)i;
        }
        if (stale >= 0) { /* RESCAN round */
            double mv = INFINITY; long mi = -1;
            for (size_t j = 0; j < n; ++j) if (M[(size_t)stale * np + j] < mv) { mv = M[(size_t)stale * np + j]; mi = (long)j; }
            rowmin[stale] = mv; rownn[stale] = mi; valid[stale] = 1;
            st->rescans++;
            continue;
        }
        if (r < 0) return 5;
        long q = rownn[r];
        size_t a = (size_t)(r < q ? r : q), b = (size_t)(r < q ? q : r);
        double dab = M[a * np + b];
        if (mode == 1) {
            if (!(cnt == 2 && rowmin[q] <= lim && rownn[q] == r)) {
                /* WINDOW round: every matrix entry <= lim in the candidate rows is re-evaluated exactly;
                 * the exact minimum wins, ties -> lexicographically lowest (a, b). */
                st->ambiguous++;
                st->rounds++;
                double best = INFINITY; long ba = -1, bb = -1;
                for (size_t i = 0; i < n; ++i) {
                    if (!active[i] || !valid[i] || !(rowmin[i] <= lim)) continue;
                    for (size_t j = 0; j < n; ++j) {
                        if (!(M[i * np + j] <= lim)) continue;
                        const long pa = (long)(i < j ? i : j), pb = (long)(i < j ? j : i);
                        const double e = sqdist_cols(xt, np, d, (size_t)pa, (size_t)pb);
                        st->exact_evals++;
                        if (e < best || (e == best && (pa < ba || (pa == ba && pb < bb)))) { best = e; ba = pa; bb = pb; }
                    }
                }
                a = (size_t)ba; b = (size_t)bb; dab = best;
            } else { dab = sqdist_cols(xt, np, d, a, b); st->exact_evals++; }
        }
        const double ma = size[a], mb = size[b], den = ma + mb;
        z[4 * step + 0] = (double)(node[a] < node[b] ? node[a] : node[b]);
        z[4 * step + 1] = (double)(node[a] < node[b] ? node[b] : node[a]);
        z[4 * step + 2] = sqrt(dab);
        z[4 * step + 3] = den;
        for (size_t k = 0; k < d; ++k)
            xt[k * np + a] = (xt[k * np + a] * ma + xt[k * np + b] * mb) / den;
        size[a] = den; node[a] = (long)(n + step); active[b] = 0; rowmin[b] = INFINITY;
        ++step; st->merges++;
        /* K2: new row/column a, kill row/column b, maintain row minima */
        const double wa = ma / den, wb = mb / den, wab = (ma * mb) / (den * den);
        double nmv = INFINITY; long nmi = -1;
        for (size_t x = 0; x < n; ++x) {
            double dc;
            if (!active[x] || x == a) dc = INFINITY;
            else if (mode == 0) dc = sqdist_cols(xt, np, d, a, x);
            else dc = wa * M[a * np + x] + wb * M[b * np + x] - wab * dab;
            if (mode == 1 && dc < 0 && dc != INFINITY) dc = 0.0;
            M[a * np + x] = dc; M[x * np + a] = dc;
            M[b * np + x] = INFINITY; M[x * np + b] = INFINITY;
            if (!active[x] || x == a) continue;
            if (dc < nmv) { nmv = dc; nmi = (long)x; }
            if (dc < rowmin[x] || (valid[x] && dc == rowmin[x] && (long)a <= rownn[x])) {
                rowmin[x] = dc; rownn[x] = (long)a; valid[x] = 1;
            } else if (valid[x] && (rownn[x] == (long)a || rownn[x] == (long)b)) {
                valid[x] = 0;
            }
        }
        rowmin[a] = nmv; rownn[a] = nmi; valid[a] = 1;
    }
    free(xt); free(M); free(rowmin); free(rownn); free(valid); free(active); free(size); free(node);
    return 0;
}
