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
//
// struct Heap follows fastcluster's binary_min_heap:
//   Copyright (c) 2011 Daniel Müllner <https://danifold.net>; changes from version 1.1.24 on (c) Google Inc.  All rights reserved.
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the following conditions
//   are met: redistributions of source code must retain the above copyright notice, this list of conditions and the following disclaimer;
//   redistributions in binary form must reproduce them in the documentation and/or other materials provided with the distribution.
//   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT NOT
//   LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE DISCLAIMED.  IN NO EVENT SHALL THE
//   COPYRIGHT HOLDER OR CONTRIBUTORS BE LIABLE FOR ANY DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES (INCLUDING,
//   BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED
//   AND ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT
//   OF THE USE OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
// (full text: ThirdPartyLicenses/fastcluster-LICENSE.md at the repository root)
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
    FA_HD double top_key() const { return key[at[0]]; }
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

// ---- The same heap, laid out for a wavefront (round 5).
// Heap above costs the selecting thread two dependent accesses per comparison (at[place] -> key[node]) and one memory round trip per level of a
// sift: ~100 dependent accesses per dendrogram row, which was what a row of the reference-order run waited for (profiles/r05_ties_probe.txt).
// Here an entry CARRIES its key (16 bytes: key, node), and a sift works on a block of entries fetched at once: a sift-up on the whole ancestor
// chain of its place (<= 31 entries), a sift-down on the eight levels below its place (510 entries), repeated from where it left off when the
// entry is still moving at the bottom of the block.  The walk on the block makes the comparisons of sift_up / sift_down above in the same
// order with the same strictness, so the array order — the tie order — is the same after every operation (tests/test_ahc_reforder_emul.py
// runs both forms side by side on tie-heavy key streams and through whole dendrograms).  HOW a block reaches the walker is the policy `Mem`:
// SerialMem (host, CPU tests) copies entry by entry; the device policy (ahc.hip: WaveMem) has the 64 lanes of the selecting wavefront load
// the block into LDS in one round trip.  On the device every lane executes every store of the walk (same address, same value): a lane's later
// loads — also the block loads, where lane l reads entries the walk wrote earlier — then see the latest values by its own program order.
struct alignas(16) Ent { double key; int32_t node, pad; };   // one 16-byte load / store

constexpr int32_t kTreeLevels = 8;                          // levels below the root of a sift-down block (16 levels = 43 200 entries: two blocks)
constexpr int32_t kTreeEnts = (1 << (kTreeLevels + 1)) - 1; // relative indices 0 (the root: not fetched) .. 510
FA_HD int32_t heap_depth(int32_t place) { int32_t L = 0; for (uint32_t v = static_cast<uint32_t>(place) + 1u; v > 1u; v >>= 1) ++L; return L; }   // ancestors of a place
FA_HD int32_t heap_ancestor(const int32_t place, const int32_t j) { return static_cast<int32_t>((static_cast<uint32_t>(place) + 1u) >> (j + 1)) - 1; }   // j = 0: the parent
// place of relative index t (children of t: 2 t + 1, 2 t + 2) in the subtree rooted at place r; 64-bit: beyond the heap for deep, wide blocks
FA_HD int64_t heap_tree_place(const int32_t r, const int32_t t) {
    int32_t lev = 0;
    for (uint32_t v = static_cast<uint32_t>(t) + 1u; v > 1u; v >>= 1) ++lev;
    return ((static_cast<int64_t>(r) + 1) << lev) - 1 + (static_cast<int64_t>(t) + 1 - (static_cast<int64_t>(1) << lev));
}

struct SerialMem {
    Ent buf[kTreeEnts + 1];
    FA_HD void fetch_chain(const Ent *ent, const int32_t place, const int32_t depth) { for (int32_t j = 0; j < depth; ++j) buf[j] = ent[heap_ancestor(place, j)]; }
    FA_HD void fetch_tree(const Ent *ent, const int32_t root, const int32_t size) {
        for (int32_t t = 1; t < kTreeEnts; ++t) { const int64_t p = heap_tree_place(root, t); if (p < size) buf[t] = ent[p]; }
    }
    FA_HD double key_at(const int32_t j) const { return buf[j].key; }
    FA_HD Ent ent_at(const int32_t j) const { return buf[j]; }
};

template <class Mem>
struct HeapK {
    Ent *ent;         // [N]    place -> (key, node)
    int32_t *pos;     // [2N]   node -> place
    int32_t size;
    Mem mem;

    FA_HD void put(const int64_t place, const Ent e) { ent[place] = e; pos[e.node] = static_cast<int32_t>(place); }
    FA_HD void sift_up(const int32_t i, const Ent e) {        // Heap::sift_up with the entry `e` arriving at place i
        const int32_t depth = heap_depth(i);
        mem.fetch_chain(ent, i, depth);
        int32_t c = 0;
        while (c < depth && e.key < mem.key_at(c)) ++c;       // strictly smaller than the ancestor: it comes down, the entry goes on
        int32_t child = i;
        for (int32_t j = 0; j < c; ++j) { put(child, mem.ent_at(j)); child = heap_ancestor(i, j); }
        put(child, e);
    }
    FA_HD void sift_down(int32_t i, const Ent e) {             // Heap::sift_down with the entry `e` arriving at place i
        for (;;) {
            mem.fetch_tree(ent, i, size);
            int32_t t = 0;                                      // relative index in the block and place in the heap of where the entry stands
            int64_t P = i;
            bool rest = false;
            while (t < (1 << kTreeLevels) - 1) {                // t above the lowest level of the block: both children are in the block
                const int32_t j = 2 * t + 1;
                const int64_t Pj = 2 * P + 1;
                if (Pj >= size) { rest = true; break; }
                const Ent l = mem.ent_at(j), r = mem.ent_at(j + 1);   // both children requested together; r means something only when Pj + 1 < size
                int32_t right;
                if (l.key >= e.key) {                           // left child not smaller: only a strictly smaller right child moves up
                    if (Pj + 1 >= size || r.key >= e.key) { rest = true; break; }
                    right = 1;
                } else right = (Pj + 1 < size && r.key < l.key) ? 1 : 0;   // both smaller: the right one only if strictly smaller than the left
                put(P, right ? r : l);
                t = j + right;
                P = Pj + right;
            }
            if (rest) { put(P, e); return; }
            i = static_cast<int32_t>(P);                        // still moving at the bottom of the block: the next levels
        }
    }
    FA_HD int32_t argmin() const { return ent[0].node; }
    FA_HD double top_key() const { return ent[0].key; }
    FA_HD void remove(const int32_t node) {                    // Heap::remove
        --size;
        const Ent last = ent[size];                             // requested next to pos[node], not behind it
        const int32_t p = pos[node];
        if (p == size) return;                                  // the last entry itself: above, a sift-up that never moves (its parent is not larger)
        if (last.key <= ent[p].key) sift_up(p, last);
        else sift_down(p, last);
    }
    FA_HD void replace(const int32_t old_node, const int32_t new_node, const double val) {   // Heap::replace
        const int32_t p = pos[old_node];
        Ent e; e.key = val; e.node = new_node; e.pad = 0;
        if (val <= ent[p].key) sift_up(p, e);
        else sift_down(p, e);
    }
    FA_HD void raise(const int32_t node, const double val) {   // Heap::raise
        Ent e; e.key = val; e.node = node; e.pad = 0;
        sift_down(pos[node], e);
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
    // remove(i) then remove(j) with the four list words (ni = next[i], pi = prev[i], nj = next[j], pj = prev[j]) READ BEFORE either removal — one
    // memory round trip instead of two dependent ones: what the first removal would have changed for the second (i and j adjacent) is patched in
    FA_HD void remove2(const int32_t i, const int32_t ni, const int32_t pi, const int32_t j, int32_t nj, int32_t pj) {
        const bool i_first = i == first;
        if (i_first) first = ni;
        else { next[pi] = ni; prev[ni] = pi; if (pi == j) nj = ni; if (ni == j) pj = pi; }
        next[i] = 0;
        if (j == first) first = nj;
        else { next[pj] = nj; prev[nj] = pj; }
        next[j] = 0;
    }
};

// What the single selecting thread does between two device-wide scans.  `Sel` carries the run; a scan is either the search of the
// nearest lower-indexed neighbour of the node created by the last merge (NEW_ROW) or the re-scan of a heap top whose recorded
// neighbour has been merged away (RESCAN).
enum : int32_t { RO_NEW_ROW = 0, RO_RESCAN = 1, RO_DONE = 2 };

template <class H>
struct SelT {
    H heap;
    ActiveList list;
    int32_t *nghbr;      // [2N]  recorded nearest lower-indexed neighbour by node
    int32_t n;           // points
    int32_t merges;      // rows of the dendrogram written so far
    int32_t op;          // scan requested next
    int32_t a, b;        // NEW_ROW: the pair merged last (a = the heap top, b = its neighbour), new node id = n + merges - 1
                         // RESCAN : a = the node to re-scan
    double *pair_a, *pair_b;   // [(n - 1)] dendrogram rows as the reference appends them: (idx1, idx2); heights are mindist[idx1]
    double *height_sq;         // [(n - 1)]

    // take the next pair off the heap unless a scan is needed first (or the run is complete).  Every word the step needs beyond the heap top is
    // requested as soon as its address is known (the neighbour's list words next to the top's), not where the reference's statements use it.
    FA_HD void advance() {
        const int32_t top = heap.argmin();
        const double top_key = heap.top_key();
        const int32_t other = nghbr[top];
        const int32_t nt = list.next[top], pt = list.prev[top];
        const int32_t no = list.next[other], po = list.prev[other];
        if (no == 0) { op = RO_RESCAN; a = top; return; }                         // the recorded neighbour is gone (:1705-1734)
        list.remove2(top, nt, pt, other, no, po);
        pair_a[merges] = static_cast<double>(top);
        pair_b[merges] = static_cast<double>(other);
        height_sq[merges] = top_key;
        ++merges;
        a = top; b = other;
        op = merges == n - 1 ? RO_DONE : RO_NEW_ROW;                              // the last merge creates no row (:1745)
    }
    // the scan requested by `op` found (value, node): lowest node id among the minima over the active nodes below the scanned one.  In two halves: what
    // does not depend on the scan's result (which heap entry goes after a merge, :1792-1797) may run while the result is still being computed.
    FA_HD void scan_begin() {
        if (op != RO_NEW_ROW) return;
        if (b < list.first) heap.remove(list.first);
        else heap.remove(b);
    }
    FA_HD void scan_finish(const double value, const int32_t node) {
        if (op == RO_NEW_ROW) {
            const int32_t created = n + merges - 1;
            nghbr[created] = node;
            heap.replace(a, created, value);
        } else {
            nghbr[a] = node;
            heap.raise(a, value);                                                  // :1731
        }
        advance();
    }
    FA_HD void scan_result(const double value, const int32_t node) { scan_begin(); scan_finish(value, node); }
};

using Sel = SelT<Heap>;

}  // namespace fa_ro
