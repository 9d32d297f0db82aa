// ahc_reforder.h — the reference's ORDER among exactly tied distances.
//
// generic_linkage_vector_alternative (reference: Sources/FastClusterWrapper/fastcluster_internal.hpp:1625-1800) always merges a pair of
// globally minimal distance; WHICH one it takes when several pairs tie exactly is decided by its bookkeeping: a binary min-heap over
// the nodes keyed by (a lower bound of) their distance to the nearest node of lower index (:778-935), a linked list of the active nodes
// (:299-349), strict `<` scans in index order (:1653-1678, :1705-1734, :1782-1790) and the rule which heap entry is dropped after a
// merge (:1792-1797).  None of this is arithmetic — the distances and centroids are — but it fixes the row order of the dendrogram and,
// where tied pairs overlap, even the tree.  The round kernel of ahc.hip breaks ties by (value, row, column); whenever its exact window
// evaluation meets an exact tie at the minimum, the run is repeated in "reference order": the device evaluates every distance the
// reference evaluates (same sequential sums) and a single thread replays the selection below, so the output is row-for-row the
// reference's on tied input too.
//
// This header is that selection logic, written for host and device (FA_HD): the same text is compiled into the HIP kernels (ahc.hip)
// and into a CPU emulation used only by the tests (tests/cpu/ahc_reforder_emul.cpp) to check it against the reference build on
// tie-heavy inputs without a GPU.  The heap below RESTATES the reference's binary_min_heap (fastcluster_internal.hpp:845-922: remove / replace /
// update_geq_ / update_leq_, statement for statement with other array names): the order a heap breaks ties in IS its sequence of swaps — which
// comparison is strict, which child wins, what moves where — so reproducing that order admits no other sequence.
#pragma once
#include <cstdint>

#ifndef FA_HD
#if defined(__HIPCC__)
#define FA_HD __host__ __device__ inline
#else
#define FA_HD inline
#endif
#endif

namespace fa_ro {

// Binary min-heap of node ids keyed by an EXTERNAL array key[node] (the reference's mindist).  pos[node] = place of the node in the
// heap, at[place] = node.  Ties: a child moves up only if it is STRICTLY smaller than its parent; on the way down the left child is
// preferred unless the right one is strictly smaller (:905-922).
struct Heap {
    double *key;      // [2N]   by node id (owned by the caller; written through set())
    int32_t *at;      // [N]    place -> node
    int32_t *pos;     // [2N]   node -> place
    int32_t size;

    FA_HD double at_key(const int32_t place) const { return key[at[place]]; }
    FA_HD void swap_places(const int32_t i, const int32_t j) {
        const int32_t t = at[i];
        at[i] = at[j];
        at[j] = t;
        pos[at[i]] = i;
        pos[at[j]] = j;
    }
    FA_HD void sift_up(int32_t i) {                       // :905-909
        while (i > 0) {
            const int32_t p = (i - 1) >> 1;
            if (!(at_key(i) < at_key(p))) break;
            swap_places(i, p);
            i = p;
        }
    }
    FA_HD void sift_down(int32_t i) {                     // :911-921
        for (;;) {
            int32_t j = 2 * i + 1;
            if (j >= size) break;
            if (at_key(j) >= at_key(i)) {                 // left child not smaller: only a strictly smaller right child moves up
                ++j;
                if (j >= size || at_key(j) >= at_key(i)) break;
            } else if (j + 1 < size && at_key(j + 1) < at_key(j)) {
                ++j;                                      // both smaller: the right one only if strictly smaller than the left
            }
            swap_places(i, j);
            i = j;
        }
    }
    // nodes first .. first + n - 1 in index order, then the bottom-up build (:845-856, :858-870)
    FA_HD void init_identity(const int32_t n, const int32_t first) {
        size = n;
        for (int32_t i = 0; i < n; ++i) { at[i] = i + first; pos[i + first] = i; }
    }
    FA_HD void heapify() {
        for (int32_t i = size >> 1; i > 0;) { --i; sift_down(i); }
    }
    FA_HD int32_t argmin() const { return at[0]; }        // :872-875
    FA_HD void remove(const int32_t node) {               // :884-895: the last entry takes the place of `node`
        --size;
        pos[at[size]] = pos[node];
        at[pos[node]] = at[size];
        if (at_key(size) <= key[node]) sift_up(pos[node]);
        else sift_down(pos[node]);
    }
    FA_HD void replace(const int32_t old_node, const int32_t new_node, const double val) {   // :897-905
        pos[new_node] = pos[old_node];
        at[pos[new_node]] = new_node;
        const bool not_more = val <= key[old_node];
        key[new_node] = val;
        if (not_more) sift_up(pos[new_node]);
        else sift_down(pos[new_node]);
    }
    FA_HD void raise(const int32_t node, const double val) {   // update_geq (:931-935): the new key is not less than the old one
        key[node] = val;
        sift_down(pos[node]);
    }
};

// The active nodes in index order (:299-349): every index 0 .. 2N-2 is a member from the start (the scans stop at the node they
// serve, so nodes that do not exist yet are never visited); next[i] == 0 marks a removed node.
struct ActiveList {
    int32_t *next;    // [2N + 1]
    int32_t *prev;    // [2N + 1]
    int32_t first;

    FA_HD void init(const int32_t count) {
        first = 0;
        for (int32_t i = 0; i < count; ++i) { prev[i + 1] = i; next[i] = i + 1; }
    }
    FA_HD void remove(const int32_t i) {
        if (i == first) first = next[i];
        else { next[prev[i]] = next[i]; prev[next[i]] = prev[i]; }
        next[i] = 0;
    }
    FA_HD bool gone(const int32_t i) const { return next[i] == 0; }
};

// What the single selecting thread does between two device-wide scans.  `Sel` carries the run; a scan is either the search of the
// nearest lower-indexed neighbour of the node created by the last merge (NEW_ROW) or the re-scan of a heap top whose recorded
// neighbour has been merged away (RESCAN).
enum : int32_t { RO_NEW_ROW = 0, RO_RESCAN = 1, RO_DONE = 2 };

struct Sel {
    Heap heap;
    ActiveList list;
    int32_t *nghbr;      // [2N]  recorded nearest lower-indexed neighbour by node
    int32_t n;           // points
    int32_t merges;      // rows of the dendrogram written so far
    int32_t op;          // scan requested next
    int32_t a, b;        // NEW_ROW: the pair merged last (a = the heap top, b = its neighbour), new node id = n + merges - 1
                         // RESCAN : a = the node to re-scan
    double *pair_a, *pair_b;   // [(n - 1)] dendrogram rows as the reference appends them: (idx1, idx2); heights are mindist[idx1]
    double *height_sq;         // [(n - 1)]

    // take the next pair(s) off the heap until a scan is needed (or the run is complete)
    FA_HD void advance() {
        for (;;) {
            const int32_t top = heap.argmin();
            if (list.gone(nghbr[top])) { op = RO_RESCAN; a = top; return; }      // :1705-1734
            const int32_t other = nghbr[top];
            list.remove(top);
            list.remove(other);
            pair_a[merges] = static_cast<double>(top);
            pair_b[merges] = static_cast<double>(other);
            height_sq[merges] = heap.key[top];
            ++merges;
            a = top; b = other;
            if (merges == n - 1) { op = RO_DONE; return; }                        // the last merge creates no row (:1745)
            op = RO_NEW_ROW;
            return;
        }
    }
    // the scan requested by `op` found (value, node): lowest node id among the minima over the active nodes below the scanned one
    FA_HD void scan_result(const double value, const int32_t node) {
        if (op == RO_NEW_ROW) {
            const int32_t created = n + merges - 1;
            nghbr[created] = node;
            if (b < list.first) heap.remove(list.first);                          // :1792-1797
            else heap.remove(b);
            heap.replace(a, created, value);
        } else {
            nghbr[a] = node;
            heap.raise(a, value);                                                  // :1731
        }
        advance();
    }
};

}  // namespace fa_ro
