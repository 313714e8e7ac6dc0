// Host-side shortest-path preprocessing: the fork-safe CPU half of the boundary (include/mobgt_cpu.h).
//
// Semantics follow graphormer/algos.pyx (floyd_warshall :9-54, get_all_edges :57-62, gen_edge_input :65-96); the
// evaluation is arranged for a CPU core rather than transliterated:
//  * Floyd-Warshall works on 16-bit distance / predecessor rows (every value is <= 510: an entry only ever
//    decreases from its initial 0 / 1 / 510), so a 814-node graph is 1.3 MB instead of 10.6 MB and stays in L2;
//    within one k, row k and column k cannot change (M[k][k] = 0), hence the j loop is a branch-free
//    compare/select the compiler vectorises, and rows with M[i][k] = 510 are skipped (510 + x can never be
//    smaller than an entry <= 510) -- same M and path, bit for bit;
//  * paths are expanded with an explicit stack of pending (from, to) segments instead of Python recursion and
//    list concatenation, writing each hop's features as soon as the segment is known to be a direct hop.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mobgt_cpu.h"

namespace {
constexpr int kUnreachable = 510;

#if defined(__x86_64__) && defined(__GNUC__)
#define MOBGT_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define MOBGT_CLONES
#endif

// one relaxation sweep of row i through pivot k: d_i[j] = min(d_i[j], via + d_k[j]), remembering k where it won
MOBGT_CLONES void relax_row(uint16_t* __restrict d_i, uint16_t* __restrict p_i, const uint16_t* __restrict d_k, uint16_t via,
                            uint16_t k, int n) {
    for (int j = 0; j < n; ++j) {
        const uint16_t cand = (uint16_t)(via + d_k[j]);
        const bool better = d_i[j] > cand;
        d_i[j] = better ? cand : d_i[j];
        p_i[j] = better ? k : p_i[j];
    }
}

// Expands the path a -> b described by `path` (k = path[a][b]; 0 = direct hop) in order.  on_hop(u, v) is called
// for every direct hop; returns false when on_hop asks to stop or the matrix does not terminate (*bad set).
template <class OnHop>
bool walk_path(const int64_t* path, int n, int a, int b, std::vector<int32_t>& stack, OnHop&& on_hop, bool* bad) {
    stack.clear();
    stack.push_back(a);
    stack.push_back(b);
    long budget = 8L * n + 16;                       // a terminating expansion over n nodes needs < 2n segments
    while (!stack.empty()) {
        if (--budget < 0) { *bad = true; return false; }
        const int v = stack.back(); stack.pop_back();
        const int u = stack.back(); stack.pop_back();
        const int64_t k64 = path[(size_t)u * n + v];
        const uint32_t k = (uint32_t)k64;            // `cdef unsigned int k = path[i][j]`
        if (k == 0) {
            if (!on_hop(u, v)) return false;
            continue;
        }
        if (k >= (uint32_t)n) { *bad = true; return false; }     // the reference would index out of bounds
        stack.push_back((int32_t)k); stack.push_back(v);          // second half later ...
        stack.push_back(u); stack.push_back((int32_t)k);          // ... first half next
    }
    return true;
}
}  // namespace

extern "C" int mobgt_cpu_abi_version(void) { return 1; }

extern "C" int mobgt_floyd_warshall_cpu(const int64_t* adj, int n, int64_t* M, int64_t* path) {
    if (n < 0) return MOBGT_CPU_EBADDIM;
    if (n == 0) return 0;
    const size_t nn = (size_t)n * n;
    const int ld = (n + 31) & ~31;                   // padded rows: the vector loop never needs a tail in the hot case
    uint16_t* buf = (uint16_t*)std::aligned_alloc(64, (((size_t)2 * n * ld * sizeof(uint16_t)) + 63) & ~(size_t)63);
    if (!buf) return MOBGT_CPU_ENOMEM;
    uint16_t* d = buf;
    uint16_t* p = buf + (size_t)n * ld;
    for (int i = 0; i < n; ++i) {
        uint16_t* di = d + (size_t)i * ld;
        const int64_t* ai = adj + (size_t)i * n;
        for (int j = 0; j < n; ++j) di[j] = (i == j) ? 0 : (ai[j] != 0 ? (uint16_t)1 : (uint16_t)kUnreachable);
        for (int j = n; j < ld; ++j) di[j] = 0;      // padding: never "better", never read back
        std::memset(p + (size_t)i * ld, 0, (size_t)ld * sizeof(uint16_t));
    }
    // adjacency values other than 0/1 keep the reference's arithmetic: M starts from the matrix itself
    bool plain = true;
    for (size_t t = 0; t < nn && plain; ++t) plain = (adj[t] == 0 || adj[t] == 1);
    if (!plain) {
        // general integer weights (the reference accepts any int matrix): 64-bit fallback, same loop order
        std::free(buf);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const int64_t v = adj[(size_t)i * n + j];
                M[(size_t)i * n + j] = (i == j) ? 0 : (v == 0 ? kUnreachable : v);
                path[(size_t)i * n + j] = 0;
            }
        for (int k = 0; k < n; ++k)
            for (int i = 0; i < n; ++i) {
                const int64_t via = M[(size_t)i * n + k];
                int64_t* mi = M + (size_t)i * n;
                const int64_t* mk = M + (size_t)k * n;
                int64_t* pi = path + (size_t)i * n;
                for (int j = 0; j < n; ++j) {
                    const int64_t cand = via + mk[j];
                    if (mi[j] > cand) { mi[j] = cand; pi[j] = k; }
                }
            }
        for (size_t t = 0; t < nn; ++t)
            if (M[t] >= kUnreachable) { M[t] = kUnreachable; path[t] = kUnreachable; }
        return 0;
    }
    for (int k = 0; k < n; ++k) {
        const uint16_t* dk = d + (size_t)k * ld;
        for (int i = 0; i < n; ++i) {
            uint16_t* di = d + (size_t)i * ld;
            const uint16_t via = di[k];
            if (via >= kUnreachable || i == k) continue;
            relax_row(di, p + (size_t)i * ld, dk, via, (uint16_t)k, n);
        }
    }
    for (int i = 0; i < n; ++i) {
        const uint16_t* di = d + (size_t)i * ld;
        const uint16_t* pi = p + (size_t)i * ld;
        for (int j = 0; j < n; ++j) {
            const bool far = di[j] >= kUnreachable;
            M[(size_t)i * n + j] = far ? kUnreachable : di[j];
            path[(size_t)i * n + j] = far ? kUnreachable : pi[j];
        }
    }
    std::free(buf);
    return 0;
}

extern "C" int mobgt_get_all_edges_cpu(const int64_t* path, int n, int i, int j, int32_t* out_nodes, int cap, int32_t* out_len) {
    if (n <= 0 || i < 0 || j < 0 || i >= n || j >= n) return MOBGT_CPU_EBADDIM;
    std::vector<int32_t> stack;
    stack.reserve(64);
    int len = 0;
    bool bad = false, overflow = false;
    // the intermediates are the end points of all hops but the last one
    int pending = -1;
    walk_path(path, n, i, j, stack, [&](int, int v) {
        if (pending >= 0) {
            if (len >= cap) { overflow = true; return false; }
            out_nodes[len++] = pending;
        }
        pending = v;
        return true;
    }, &bad);
    *out_len = len;
    if (bad) return MOBGT_CPU_ERECURSION;
    return overflow ? MOBGT_CPU_EINDEX : 0;
}

extern "C" int mobgt_gen_edge_input_cpu(int max_dist, const int64_t* path, const int64_t* edge_feat, int n, int F, float* out) {
    if (n < 0 || F < 1 || max_dist < 0) return MOBGT_CPU_EBADDIM;
    const size_t per_pair = (size_t)max_dist * F;
    const size_t total = (size_t)n * n * per_pair;
    for (size_t t = 0; t < total; ++t) out[t] = -1.0f;
    std::vector<int32_t> stack;
    stack.reserve(64);
    int rc = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (i == j || path[(size_t)i * n + j] == kUnreachable) continue;
            float* dst = out + ((size_t)i * n + j) * per_pair;
            int hop = 0;
            bool bad = false, overflow = false;
            walk_path(path, n, i, j, stack, [&](int u, int v) {
                if (hop >= max_dist) { overflow = true; return false; }
                const int64_t* src = edge_feat + ((size_t)u * n + v) * F;
                for (int f = 0; f < F; ++f) dst[(size_t)hop * F + f] = (float)(double)src[f];   // via float64, algos.pyx:76
                ++hop;
                return true;
            }, &bad);
            if (bad) return MOBGT_CPU_ERECURSION;
            if (overflow) rc = MOBGT_CPU_EINDEX;         // the reference raises at the first such pair
            if (rc) return rc;
        }
    return rc;
}
