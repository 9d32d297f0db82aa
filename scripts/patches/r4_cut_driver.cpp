#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>
#include <dlfcn.h>
#include "fluidaudio_hip.h"
typedef fa_status (*cut_fn)(const double *, size_t, double, int32_t *);
int main() {
    void *h = dlopen("/root/repo/fluidaudio_amd/csrc/libfluidaudio_hip.so", RTLD_NOW | RTLD_LOCAL);   // the shipped (unpatched) cut, for valid trees
    cut_fn shipped = h ? reinterpret_cast<cut_fn>(dlsym(h, "fa_ahc_cut")) : nullptr;
    if (!shipped) { std::printf("no shipped library\n"); return 2; }
    std::mt19937_64 rng(3);
    long valid = 0, refused = 0;
    for (int it = 0; it < 200000; ++it) {
        const size_t n = 2 + rng() % 30;
        std::vector<double> z(4 * (n - 1));
        std::vector<size_t> alive(n);
        for (size_t i = 0; i < n; ++i) alive[i] = i;
        for (size_t r = 0; r + 1 < n; ++r) {                     // a valid random tree with non-monotone heights
            const size_t i = rng() % alive.size(); size_t a = alive[i]; alive.erase(alive.begin() + i);
            const size_t j = rng() % alive.size(); size_t b = alive[j]; alive.erase(alive.begin() + j);
            z[4 * r] = static_cast<double>(a < b ? a : b); z[4 * r + 1] = static_cast<double>(a < b ? b : a);
            z[4 * r + 2] = (rng() % 2000) / 1000.0; z[4 * r + 3] = 2;
            alive.push_back(n + r);
        }
        const double thr = (rng() % 2200) / 1000.0;
        std::vector<int32_t> la(n, -9), lb(n, -9);
        const bool mutate = it % 2;
        if (mutate) {
            const size_t r = rng() % (n - 1), c = rng() % 2;
            switch (rng() % 6) {
                case 0: z[4 * r + c] = static_cast<double>(n + r + rng() % 3); break;        // its own node or a later one
                case 1: z[4 * r + c] = -1.0; break;
                case 2: z[4 * r + c] = 1e18; break;
                case 3: z[4 * r + c] += 0.5; break;
                case 4: z[4 * r] = z[4 * r + 1]; break;
                default: z[4 * r + c] = z[4 * ((r + 1) % (n - 1)) + c]; break;               // (possibly) a node merged twice
            }
        }
        const fa_status sa = fa_ahc_cut(z.data(), n, thr, la.data());                       // the patched cut (linked in)
        if (sa == FA_INVALID_ARGUMENT) { ++refused; continue; }
        if (sa != FA_SUCCESS) { std::printf("status %d\n", sa); return 1; }
        // accepted => it is a walkable tree, and the shipped cut gives the same labels
        if (shipped(z.data(), n, thr, lb.data()) != FA_SUCCESS || la != lb) { std::printf("labels differ at %d\n", it); return 1; }
        ++valid;
    }
    std::printf("done: %ld accepted (equal to the shipped cut), %ld refused\n", valid, refused);
    return 0;
}
