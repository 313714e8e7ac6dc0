/* C ABI of mobgt_amd/libmobgt_cpu.so -- the host-side, fork-safe half of the drop-in boundary.
 *
 * The reference calls its Cython module per sample from forked DataLoader workers
 * (graphormer/wrapper.py:55-60 under data.py:282-295, `--num_workers 8` in README.md:62), where HIP must never
 * be initialised.  This library is plain C++ (g++, no HIP, no threads, no global state), so
 * `mobgt_amd.algos` can serve those calls inside a worker; the batched device path for whole padded batches is
 * mobgt_spd_batched in include/mobgt_hip.h.
 *
 * All functions return 0 on success or one of the MOBGT_CPU_E* codes; buffers are caller-owned, C-contiguous.
 */
#ifndef MOBGT_CPU_H
#define MOBGT_CPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOBGT_CPU_EBADDIM (-1)   /* n < 0, F < 1, max_dist < 0                                    */
#define MOBGT_CPU_EINDEX 1       /* a path has more hops than max_dist (the reference: IndexError) */
#define MOBGT_CPU_ERECURSION 3   /* the path matrix does not terminate (the reference: RecursionError) */
#define MOBGT_CPU_ENOMEM 4

int mobgt_cpu_abi_version(void);

/* graphormer/algos.pyx:9-54  floyd_warshall(adjacency_matrix) -> (M, path).
 * adj [n,n] int64, non-zero = edge.  M, path [n,n] int64 out: M[i][i] = 0, unreachable = 510 in both;
 * path[i][j] = the LAST k that improved (i,j) under the reference's k-outermost, strict '>' order, 0 if none. */
int mobgt_floyd_warshall_cpu(const int64_t* adj, int n, int64_t* M, int64_t* path);

/* graphormer/algos.pyx:65-96  gen_edge_input(max_dist, path, edge_feat) -> float32 [n,n,max_dist,F], -1 fill;
 * hop k of the reconstructed path i -> j receives edge_feat[p_k, p_{k+1}, :].  `path` may be ANY matrix (the
 * reference takes what it is given); an intermediate node 0 reads as "no intermediate" (algos.pyx:58-59). */
int mobgt_gen_edge_input_cpu(int max_dist, const int64_t* path, const int64_t* edge_feat, int n, int F, float* out);

/* graphormer/algos.pyx:57-62  get_all_edges(path, i, j): intermediate nodes of the path i -> j into out_nodes
 * (capacity cap); *out_len receives their number. */
int mobgt_get_all_edges_cpu(const int64_t* path, int n, int i, int j, int32_t* out_nodes, int cap, int32_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
