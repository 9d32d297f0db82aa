/*
 * ahc_model.c — CPU MODEL of the GPU merge-round algorithm (test infrastructure only).
 *
 * Mirrors, step for step, the round structure of fluidaudio_amd/csrc/ahc_round_body.h
 * (asymmetric slot matrix M, per-row minima with lazy rescans, Lance-Williams filter with
 * mutual-nearest certification and exact window re-evaluation, exact heights after the loop) so that the algorithm itself can be checked against the
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

static double sqdist_rows(const double *c, size_t d, long a, long b) { /* the reference's sequential sum */
    double s = 0.0;
    for (size_t k = 0; k < d; ++k) { const double diff = c[(size_t)a * d + k] - c[(size_t)b * d + k]; s += diff * diff; }
    return s;
}
static double treesum_rows(const double *c, size_t d, long a, long b) { /* pairwise-tree sum (what the device uses inside Lance-Williams) */
    double part[256];
    for (int t = 0; t < 256; ++t) part[t] = 0.0;
    for (size_t k = 0; k < d; ++k) { const double diff = c[(size_t)a * d + k] - c[(size_t)b * d + k]; part[k & 255] += diff * diff; }
    for (int w = 128; w > 0; w >>= 1) for (int t = 0; t < w; ++t) part[t] += part[t + w];
    return part[0];
}

static int g_piggy = 0;  /* piggy-back re-scans per merge round (0 = the lazy scheme only) */
void ahc_model_set_piggyback(int k) { g_piggy = k; }
/* round-2 row bookkeeping of the device (csrc/ahc.hip): e2[x] = lower bound of the entries of row x other than its nearest
 * neighbour's — exact second minimum after a full scan of the start-up, the row minimum after a later re-scan (no second minimum is
 * computed there), lowered by new entries.  A row whose nearest neighbour is one of the merged slots keeps the new cluster as its
 * neighbour when the new entry is STRICTLY below e2; otherwise it goes stale with the bound e2.  Also: pairs of nodes that both
 * existed at the start-up read the row copy (both triangles were written). */
static int g_second = 0;
void ahc_model_set_second_bound(int on) { g_second = on; }

#define DEAD 0x7fffffffL
/* valid copy of the pair (x, y): the row of the slot holding the younger node */
#define VAL(x, y) ((node[x] > node[y] || (g_second && node[x] < (long)n && node[y] < (long)n)) ? M[(size_t)(x) * np + (y)] : M[(size_t)(y) * np + (x)])

/* mode 0: exact rows (every new-row entry is the reference's sequential fp64 sum)
 * mode 1: Lance-Williams rows; the pair is taken from them only when it is the unique mutual-nearest pair with
 *         every other row minimum beyond 2*eps; otherwise every matrix entry inside the window is re-evaluated
 *         exactly.  eps = eps_scale * N * u * dmax (the library uses eps_scale = 16).
 * Storage is asymmetric like the device's: a merge writes ONE row; VAL() picks the valid copy.
 * Heights are recomputed exactly after the loop from the stored centroids. */
int ahc_model_linkage(const double *data, size_t n, size_t d, double *z, int mode, double eps_scale,
                      ahc_model_stats *st) {
    memset(st, 0, sizeof(*st));
    if (n < 2) return 0;
    const size_t np = n;
    double *C = (double *)malloc(sizeof(double) * d * 2 * n);
    double *M = (double *)malloc(sizeof(double) * np * np);
    double *d1 = (double *)malloc(sizeof(double) * np);
    double *e2 = (double *)malloc(sizeof(double) * np);
    long *nn = (long *)malloc(sizeof(long) * np);
    long *node = (long *)malloc(sizeof(long) * np);
    double *size = (double *)malloc(sizeof(double) * 2 * n);
    if (!C || !M || !d1 || !e2 || !nn || !node || !size) return 4;
    memcpy(C, data, sizeof(double) * n * d);
    double dmax = 0.0;
    for (size_t i = 0; i < n; ++i) {
        node[i] = (long)i; size[i] = 1.0;
        for (size_t j = 0; j < n; ++j) {
            double v = i == j ? INFINITY : sqdist_rows(C, d, (long)i, (long)j);
            M[i * np + j] = v;
            if (i != j && v > dmax) dmax = v;
        }
    }
    for (size_t i = 0; i < n; ++i) {
        double mv = INFINITY, m2 = INFINITY; long mi = -1;
        for (size_t j = 0; j < n; ++j) { const double v = M[i * np + j]; if (v < mv) { m2 = mv; mv = v; mi = (long)j; } else if (v < m2) m2 = v; }
        d1[i] = mv; nn[i] = mi; e2[i] = m2;
    }
    const double eps = mode == 1 ? eps_scale * (double)n * 1.1102230246251565e-16 * dmax : 0.0;
    size_t step = 0;
    while (step + 1 < n) {
        st->rounds++;
        /* three smallest row minima (value, row) */
        double g1 = INFINITY, g2 = INFINITY, g3 = INFINITY; long R1 = -1, R2 = -1;
        for (size_t i = 0; i < n; ++i) {
            if (node[i] == DEAD) continue;
            const double v = d1[i];
            if (v < g1) { g3 = g2; g2 = g1; R2 = R1; g1 = v; R1 = (long)i; }
            else if (v < g2) { g3 = g2; g2 = v; R2 = (long)i; }
            else if (v < g3) g3 = v;
        }
        if (R1 < 0) return 5;
        long rescan = -1, a = -1, b = -1; double dab_exact = -1.0;
        if (nn[R1] < 0) rescan = R1;
        else if (mode == 0) { a = R1 < nn[R1] ? R1 : nn[R1]; b = R1 < nn[R1] ? nn[R1] : R1; }
        else {
            const double lim = g1 + 2.0 * eps;
            if (g2 <= lim && !(g3 <= lim) && R2 == nn[R1] && nn[R2] == R1) { a = R1 < R2 ? R1 : R2; b = R1 < R2 ? R2 : R1; }
            else {
                /* COLLECT: stale rows inside the window are re-scanned first */
                st->ambiguous++; st->rounds++;
                for (size_t i = 0; i < n && rescan < 0; ++i) if (node[i] != DEAD && d1[i] <= lim && nn[i] < 0) rescan = (long)i;
                if (rescan < 0) {
                    st->rounds++;
                    double best = INFINITY; long ba = -1, bb = -1;
                    for (size_t i = 0; i < n; ++i) {
                        if (node[i] == DEAD || !(d1[i] <= lim)) continue;
                        for (size_t j = 0; j < n; ++j) {
                            if (j == i || node[j] == DEAD || !(VAL(i, j) <= lim)) continue;
                            const long pa = (long)(i < j ? i : j), pb = (long)(i < j ? j : i);
                            const double e = sqdist_rows(C, d, node[pa], node[pb]);
                            st->exact_evals++;
                            if (e < best || (e == best && (pa < ba || (pa == ba && pb < bb)))) { best = e; ba = pa; bb = pb; }
                        }
                    }
                    a = ba; b = bb; dab_exact = best;
                }
            }
        }
        if (rescan >= 0) {
            double mv = INFINITY; long mi = -1;
            for (size_t j = 0; j < n; ++j) {
                if ((long)j == rescan || node[j] == DEAD) continue;
                const double v = VAL(rescan, j);
                if (v < mv) { mv = v; mi = (long)j; }
            }
            d1[rescan] = mv; nn[rescan] = mi; e2[rescan] = mv;
            st->rescans++;
            continue;
        }
        const long na = node[a], nb = node[b], nnew = (long)(n + step);
        const double ma = size[na], mb = size[nb], den = ma + mb;
        z[4 * step + 0] = (double)(na < nb ? na : nb);
        z[4 * step + 1] = (double)(na < nb ? nb : na);
        z[4 * step + 2] = 0.0;
        z[4 * step + 3] = den;
        for (size_t k = 0; k < d; ++k)
            C[(size_t)nnew * d + k] = (C[(size_t)na * d + k] * ma + C[(size_t)nb * d + k] * mb) / den;
        const double dab = dab_exact >= 0.0 ? dab_exact : treesum_rows(C, d, na, nb);
        const double wa = ma / den, wb = mb / den, wab = (ma * mb) / (den * den);
        double nmv = INFINITY; long nmi = -1;
        /* new row a (old node ids still in place while VAL() is evaluated) */
        double *newrow = (double *)malloc(sizeof(double) * n);
        for (size_t x = 0; x < n; ++x) {
            if (node[x] == DEAD || (long)x == a || (long)x == b) { newrow[x] = INFINITY; continue; }
            double dc;
            if (mode == 0) dc = sqdist_rows(C, d, nnew, node[x]);
            else { dc = wa * VAL(a, x) + wb * VAL(b, x) - wab * dab; if (dc < 0) dc = 0.0; }
            newrow[x] = dc;
        }
        size[nnew] = den; node[a] = nnew; node[b] = DEAD; d1[b] = INFINITY; nn[b] = -1;
        for (size_t x = 0; x < n; ++x) {
            if (node[x] == DEAD || (long)x == a) continue;
            const double dc = newrow[x];
            M[(size_t)a * np + x] = dc;  /* the only matrix write of the merge */
            if (dc < nmv) { nmv = dc; nmi = (long)x; }
            const int vld = nn[x] >= 0;
            if (!g_second) {
                if (dc < d1[x] || (vld && dc == d1[x] && a <= nn[x])) { d1[x] = dc; nn[x] = a; }
                else if (vld && (nn[x] == a || nn[x] == b)) nn[x] = -1;
            } else {
                const int hit = vld && (nn[x] == a || nn[x] == b);
                if (!hit) {
                    if (dc < d1[x] || (vld && dc == d1[x] && a <= nn[x])) { e2[x] = d1[x]; d1[x] = dc; nn[x] = a; }
                    else if (dc < e2[x]) e2[x] = dc;
                } else if (dc < e2[x]) { d1[x] = dc; nn[x] = a; }
                else { d1[x] = e2[x]; nn[x] = -1; }
            }
        }
        free(newrow);
        d1[a] = nmv; nn[a] = nmi; e2[a] = nmv;
        ++step; st->merges++;
        /* piggy-back: the same round also re-scans the stale row(s) with the smallest lower bound */
        for (int pg = 0; pg < g_piggy; ++pg) {
            long srow = -1; double sv = INFINITY;
            for (size_t i = 0; i < n; ++i) if (node[i] != DEAD && nn[i] < 0 && d1[i] < sv) { sv = d1[i]; srow = (long)i; }
            if (srow < 0) break;
            double mv = INFINITY; long mi = -1;
            for (size_t j = 0; j < n; ++j) {
                if ((long)j == srow || node[j] == DEAD) continue;
                const double v = VAL(srow, j);
                if (v < mv) { mv = v; mi = (long)j; }
            }
            d1[srow] = mv; nn[srow] = mi; e2[srow] = mv;
        }
    }
    for (size_t s = 0; s + 1 < n; ++s) z[4 * s + 2] = sqrt(sqdist_rows(C, d, (long)z[4 * s], (long)z[4 * s + 1]));
    free(C); free(M); free(d1); free(e2); free(nn); free(node); free(size);
    return 0;
}
