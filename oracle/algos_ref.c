/*
 * ORACLE -- test infrastructure, not product code.
 *
 * Plain-C CPU restatement of the reference's only native component, graphormer/algos.pyx.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (mobgt_amd/) never does.
 *
 * Parity pinning: checked bit-for-bit against tests/golden/g1_algos.npz and g2_collator.npz,
 * which were produced by running the reference's Cython build of algos.pyx in the build
 * container (tests/golden/make_golden.py).
 *
 *   oracle_floyd_warshall   <- algos.pyx:9-54
 *   oracle_get_all_edges    <- algos.pyx:57-62
 *   oracle_gen_edge_input   <- algos.pyx:65-96
 */
#include <stdint.h>
#include <stdlib.h>

#define UNREACH 510

/* algos.pyx:9-54.  adj: n*n, non-zero = edge.  M, path: n*n int64 outputs. */
int oracle_floyd_warshall(const int64_t* adj, int n, int64_t* M, int64_t* path)
{
    /* algos.pyx:27-32: copy, diagonal 0, non-edges 510 */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            int64_t v = adj[(size_t)i * n + j];
            if (i == j) v = 0;
            else if (v == 0) v = UNREACH;
            M[(size_t)i * n + j] = v;
            path[(size_t)i * n + j] = 0;
        }
    /* algos.pyx:35-45: k outermost, strict '>', in place, M_ik read once per (k,i) */
    for (int k = 0; k < n; ++k) {
        const int64_t* Mk = M + (size_t)k * n;
        for (int i = 0; i < n; ++i) {
            int64_t* Mi = M + (size_t)i * n;
            int64_t Mik = Mi[k];
            for (int j = 0; j < n; ++j) {
                int64_t c = Mik + Mk[j];
                if (Mi[j] > c) {
                    Mi[j] = c;
                    path[(size_t)i * n + j] = k;
                }
            }
        }
    }
    /* algos.pyx:48-52 */
    for (size_t t = 0; t < (size_t)n * n; ++t)
        if (M[t] >= UNREACH) { path[t] = UNREACH; M[t] = UNREACH; }
    return 0;
}

/* algos.pyx:57-62: in-order expansion; k == 0 means "no intermediate" (so an intermediate node 0
 * truncates the path -- reference quirk, kept).  Appends to out, returns new length, or -1 if the
 * buffer would overflow (the reference would raise IndexError later, when writing hop >= max_dist). */
static int get_all_edges_rec(const int64_t* path, int n, int i, int j, int* out, int len, int cap, int depth)
{
    if (depth > 4 * n + 8) return -1;               /* malformed path matrix: the reference would recurse forever */
    int k = (int)path[(size_t)i * n + j];
    if (k == 0) return len;
    len = get_all_edges_rec(path, n, i, k, out, len, cap, depth + 1);
    if (len < 0) return -1;
    if (len >= cap) return -1;
    out[len++] = k;
    return get_all_edges_rec(path, n, k, j, out, len, cap, depth + 1);
}

int oracle_get_all_edges(const int64_t* path, int n, int i, int j, int* out, int cap)
{
    return get_all_edges_rec(path, n, i, j, out, 0, cap, 0);
}

/* algos.pyx:65-96.  feat: n*n*F int64.  out: n*n*max_dist*F float32, filled with -1.
 * Returns 0, or 1 if some path has more hops than max_dist (the reference raises IndexError). */
int oracle_gen_edge_input(int max_dist, const int64_t* path, const int64_t* feat, int n, int F, float* out)
{
    size_t total = (size_t)n * n * (size_t)max_dist * F;
    for (size_t t = 0; t < total; ++t) out[t] = -1.0f;
    int cap = n + 2;
    int* nodes = (int*)malloc(sizeof(int) * (size_t)(cap + 2));
    if (!nodes) return 2;
    int rc = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (i == j) continue;
            if (path[(size_t)i * n + j] == UNREACH) continue;
            nodes[0] = i;
            int len = get_all_edges_rec(path, n, i, j, nodes + 1, 0, cap - 1, 0);
            if (len < 0) { rc = 1; continue; }
            nodes[len + 1] = j;
            int num_path = len + 1;
            for (int k = 0; k < num_path; ++k) {
                if (k >= max_dist) { rc = 1; break; }
                const int64_t* src = feat + ((size_t)nodes[k] * n + nodes[k + 1]) * F;
                float* dst = out + (((size_t)i * n + j) * max_dist + k) * F;
                for (int f = 0; f < F; ++f) dst[f] = (float)(double)src[f];   /* float64 round trip, algos.pyx:75-76 */
            }
        }
    free(nodes);
    return rc;
}
