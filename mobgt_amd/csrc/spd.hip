// Batched shortest-path preprocessing on gfx950: the device counterpart of graphormer/algos.pyx
// (floyd_warshall :9-54, get_all_edges :57-62, gen_edge_input :65-96) plus the index shifts that
// wrapper.py:55-61,97-98 and collator.py:11-18,76-93 apply afterwards, for a whole padded batch.
//
// Bit-exactness notes:
//   * k stays sequential; inside one k, row k and column k are invariant, so all (i,j) updates of
//     that step run in parallel with the reference's strict '>' and "last improving k" semantics;
//   * "unreachable" is the literal 510 sentinel, 510+510 arithmetic included (int16 holds 1020);
//   * the path walk reproduces the recursion of get_all_edges exactly, including the quirk that an
//     intermediate node 0 is indistinguishable from "no intermediate" (k == 0 ends the descent);
//   * only the first D hops are emitted (collator.py:323 drops the rest).  The in-order expansion is
//     run with a bounded LIFO of D pending targets per pair: an entry deeper than D can only be
//     reached after D hops have been emitted, so dropping it never changes the output.
#include "common.h"
#include "mobgt_hip.h"

namespace {

constexpr int UNREACH = 510;
constexpr int FW_THREADS = 1024;
constexpr int LDS_M_MAX_N = 272;          // 272*272*2 B = 144.5 KiB of the 160 KiB LDS
constexpr int MAXD = 32;

template <bool IN_LDS>
__global__ __launch_bounds__(FW_THREADS) void fw_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ n_nodes,
                                                        int16_t* __restrict__ spd, int16_t* __restrict__ path,
                                                        int16_t* __restrict__ in_degree, int16_t* __restrict__ out_degree,
                                                        int N, const int* __restrict__ only_if) {
    extern __shared__ __attribute__((aligned(16))) int16_t Ml[];
    const int g = blockIdx.x;
    if (only_if && only_if[g] == 0) return;             // redo pass after fw_split_kernel: only graphs that gave up
    const int n = n_nodes[g];
    const int32_t* C = counts + (int64_t)g * N * N;
    int16_t* Mg = spd + (int64_t)g * N * N;
    int16_t* Pg = path + (int64_t)g * N * N;
    const int tid = threadIdx.x;
    const int pitch = IN_LDS ? n : N;
    int16_t* M = IN_LDS ? Ml : Mg;

    // degrees (wrapper.py:97-98: "in" = row sum, "out" = column sum of the 0/1 adjacency), +1, pad 0
    for (int i = tid; i < N; i += FW_THREADS) {
        int rs = 0, cs = 0;
        if (i < n) {
            for (int j = 0; j < n; ++j) {
                rs += C[(int64_t)i * N + j] != 0;
                cs += C[(int64_t)j * N + i] != 0;
            }
            rs += 1; cs += 1;
        }
        in_degree[(int64_t)g * N + i] = (int16_t)rs;
        out_degree[(int64_t)g * N + i] = (int16_t)cs;
    }
    // init (algos.pyx:27-32)
    for (int e = tid; e < n * n; e += FW_THREADS) {
        const int i = e / n, j = e - i * n;
        const int v = i == j ? 0 : (C[(int64_t)i * N + j] != 0 ? 1 : UNREACH);
        M[i * pitch + j] = (int16_t)v;
        Pg[(int64_t)i * N + j] = 0;
    }
    __syncthreads();
    // algos.pyx:35-45
    for (int k = 0; k < n; ++k) {
        for (int e = tid; e < n * n; e += FW_THREADS) {
            const int i = e / n, j = e - i * n;
            const int c = (int)M[i * pitch + k] + (int)M[k * pitch + j];
            if ((int)M[i * pitch + j] > c) {
                M[i * pitch + j] = (int16_t)c;
                Pg[(int64_t)i * N + j] = (int16_t)k;
            }
        }
        __syncthreads();
    }
    // algos.pyx:48-52 + padding
    for (int e = tid; e < N * N; e += FW_THREADS) {
        const int i = e / N, j = e - i * N;
        if (i < n && j < n) {
            int v = M[i * pitch + j];
            if (v >= UNREACH) { v = UNREACH; Pg[e] = UNREACH; }
            Mg[e] = (int16_t)v;
        } else {
            Mg[e] = -1;
            Pg[e] = -1;
        }
    }
}

// ---- long graphs: one graph over SEVERAL workgroups -------------------------------------------------------------
// A graph with N > 272 nodes does not fit one CU's LDS, and one workgroup walking N^2 pairs N times out of global
// memory took ~120 ms for 16 graphs of 784 nodes (15 of 16 CUs idle per graph slot).  Here FW_PARTS workgroups
// share a graph: each keeps a block of ROWS of M in its LDS.  Step k of algos.pyx:35-45 needs, beyond the
// workgroup's own rows, only row k as it stands after step k-1 -- so the owner of row k+1 updates that row FIRST
// in step k, publishes it (into the spd output array, which nobody reads until the end) and raises flag[k+1];
// the other workgroups wait on that flag only.  No grid-wide barrier per step: workgroups run ahead as far as
// the rows they need have been published.  The update itself is the reference's, bit for bit: strict '>',
// "last improving k", literal 510 sentinel (rows whose M[i][k] is 510 cannot improve and are skipped).
// Visibility across CUs / XCDs (MI355X_MICROARCH "inter-workgroup visibility", the sc1 form): every store of a
// published row and of its flag is an agent-scope (sc1, write-through) store, drained (s_waitcnt vmcnt(0) in
// every storing wave + workgroup barrier) before the flag is raised; every load of the flag and of the row is
// an agent-scope (sc1) load.  A release/acquire pair per step (L2 write-back + invalidate) cost ~15 us x N steps.
constexpr int FW_PARTS = 16;
constexpr int FW_SPLIT_THREADS = 512;
constexpr int FW_SPLIT_MAX_N = 1088;     // ceil(N/16) rows x N x 2 B + one row <= 160 KiB
// A workgroup waits for rows other workgroups publish.  Launches are sized to the device's resident capacity, but
// nothing guarantees that capacity while other streams (RCCL, a second graph, another process) hold CUs or LDS: a
// waiter whose producer cannot be scheduled would spin forever.  Every wait is therefore bounded in WALL time
// (s_memrealtime, 100 MHz); a workgroup that runs out raises its graph's `gave_up` flag and leaves, every other
// workgroup of that graph follows at its next wait, and mobgt_spd_batched re-runs exactly those graphs with the
// single-workgroup kernel (slower, same result).  A whole 16 x 784 batch takes ~8 ms; 200 ms is far outside it.
__device__ long long g_fw_spin_ticks = 20000000;      // 200 ms; mobgt_spd_set_spin_limit (tests) overrides

__global__ __launch_bounds__(FW_SPLIT_THREADS) void fw_split_kernel(const int32_t* __restrict__ counts,
                                                                    const int32_t* __restrict__ n_nodes, int16_t* spd,
                                                                    int16_t* __restrict__ path, int16_t* __restrict__ in_degree,
                                                                    int16_t* __restrict__ out_degree, int* flags, int* gave_up,
                                                                    uint32_t* pub, int N, int g0) {
    extern __shared__ __attribute__((aligned(16))) int16_t lds16[];
    __shared__ int leave_s;
    // workgroups b and b+8 land on the same XCD (round-robin dispatch): when the graph count allows, give every
    // graph's FW_PARTS workgroups one XCD, so that published rows travel through that XCD's L2 only (speed only)
    const int gc = gridDim.x / FW_PARTS;
    int gl = blockIdx.x / FW_PARTS, part = blockIdx.x % FW_PARTS;
    if (gc % 8 == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        gl = xcd * (gc / 8) + slot / FW_PARTS;
        part = slot % FW_PARTS;
    }
    const int g = g0 + gl;
    const int n = n_nodes[g];
    const int tid = threadIdx.x;
    const int32_t* C = counts + (int64_t)g * N * N;
    int16_t* Mg = spd + (int64_t)g * N * N;
    int16_t* Pg = path + (int64_t)g * N * N;
    int* flag = flags + (int64_t)g * N;
    const int npw = ((N + 7) & ~7) / 2;                              // words per published row
    uint32_t* pubg = pub + (int64_t)g * N * npw;
    const int rpw = (n + FW_PARTS - 1) / FW_PARTS;                  // rows per workgroup
    const int r0 = part * rpw, r1 = min(n, r0 + rpw), rows = max(0, r1 - r0);
    const int np = (n + 7) & ~7;                                     // LDS row pitch
    int16_t* M = lds16;                                              // [rpw][np]
    int16_t* rowk = lds16 + (size_t)rpw * np;                        // [np]

    // padding rows / columns of the outputs and the degrees are split over the graph's workgroups by row
    for (int i = part; i < N; i += FW_PARTS) {
        if (i >= n) {
            for (int j = tid; j < N; j += FW_SPLIT_THREADS) { Mg[(int64_t)i * N + j] = -1; Pg[(int64_t)i * N + j] = -1; }
            if (tid == 0) { in_degree[(int64_t)g * N + i] = 0; out_degree[(int64_t)g * N + i] = 0; }
        }
    }
    if (rows == 0) return;                                            // (nobody waits for a workgroup without rows)
    for (int i = r0 + tid; i < r1; i += FW_SPLIT_THREADS) {           // wrapper.py:97-98, +1
        int rs = 0, cs = 0;
        for (int j = 0; j < n; ++j) { rs += C[(int64_t)i * N + j] != 0; cs += C[(int64_t)j * N + i] != 0; }
        in_degree[(int64_t)g * N + i] = (int16_t)(rs + 1);
        out_degree[(int64_t)g * N + i] = (int16_t)(cs + 1);
    }
    // init (algos.pyx:27-32)
    for (int e = tid; e < rows * np; e += FW_SPLIT_THREADS) {
        const int il = e / np, j = e - il * np, i = r0 + il;
        if (j < n) {
            M[il * np + j] = (int16_t)(i == j ? 0 : (C[(int64_t)i * N + j] != 0 ? 1 : UNREACH));
            Pg[(int64_t)i * N + j] = 0;
        } else {
            M[il * np + j] = 0;                                       // padding columns: the minimum, never relaxed
        }
    }
    __syncthreads();
    auto publish = [&](int r) {                                       // row r (owned here) -> pub row r, flag[r] = 1
        const uint32_t* src = reinterpret_cast<const uint32_t*>(M + (size_t)(r - r0) * np);
        for (int w = tid; w < np / 2; w += FW_SPLIT_THREADS)
            __hip_atomic_store(pubg + (int64_t)r * npw + w, src[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains its stores ...
        __syncthreads();                                              // ... before the flag goes up
        if (tid == 0) __hip_atomic_store(&flag[r], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // algos.pyx:38-45 for rows [il0, il1) of this workgroup except `skip`.  A thread owns one PAIR of columns (one
    // dword of every row) and walks the rows four at a time, so that the LDS reads of four independent rows are in
    // flight together: one row at a time, each row paid a full LDS round trip (measured ~15 us per step).
    uint32_t* M32 = reinterpret_cast<uint32_t*>(M);
    const uint32_t* rowk32 = reinterpret_cast<const uint32_t*>(rowk);
    const int npw2 = np / 2;
    auto relax_rows = [&](int k, int il0, int il1, int skip) {
        for (int w = tid; w < npw2; w += FW_SPLIT_THREADS) {
            const uint32_t rk = rowk32[w];
            const int lo = (int)(rk & 0xffffu), hi = (int)(rk >> 16);
            for (int il = il0; il < il1; il += 4) {
                int mk[4];
                uint32_t m2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = il + u;
                    const bool on = r < il1 && r != skip;
                    mk[u] = on ? (int)M[r * np + k] : UNREACH;
                    m2[u] = on ? M32[r * npw2 + w] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (mk[u] >= UNREACH) continue;               // 510 + x > anything stored: no improvement
                    const int r = il + u;
                    const int m0 = (int)(m2[u] & 0xffffu), m1 = (int)(m2[u] >> 16);
                    const int c0 = mk[u] + lo, c1 = mk[u] + hi;
                    const bool u0 = m0 > c0, u1 = m1 > c1;        // (padding columns hold 0: never improved)
                    if (u0 | u1) {
                        M32[r * npw2 + w] = (uint32_t)(u0 ? c0 : m0) | ((uint32_t)(u1 ? c1 : m1) << 16);
                        int16_t* prow = Pg + (int64_t)(r0 + r) * N + 2 * w;
                        if (u0) prow[0] = (int16_t)k;
                        if (u1) prow[1] = (int16_t)k;
                    }
                }
            }
        }
    };
    if (r0 == 0) publish(0);
    for (int k = 0; k < n; ++k) {
        if (k >= r0 && k < r1) {                                      // my own row, current through step k-1
            for (int j = tid; j < np; j += FW_SPLIT_THREADS) rowk[j] = M[(k - r0) * np + j];
        } else {
            if (tid == 0) {
                const long long t0 = wall_clock64(), limit = g_fw_spin_ticks;
                int leave = 0;
                while (__hip_atomic_load(&flag[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__hip_atomic_load(&gave_up[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                        wall_clock64() - t0 > limit) {
                        __hip_atomic_store(&gave_up[g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        leave = 1;
                        break;
                    }
                }
                leave_s = leave;
            }
            __syncthreads();
            if (leave_s) return;                                      // the redo pass recomputes this graph
            uint32_t* dst = reinterpret_cast<uint32_t*>(rowk);
            for (int w = tid; w < np / 2; w += FW_SPLIT_THREADS)
                dst[w] = __hip_atomic_load(pubg + (int64_t)k * npw + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int nxt = k + 1;
        const bool own_next = nxt < n && nxt >= r0 && nxt < r1;
        if (own_next) {                                               // the row everybody waits for next: first
            relax_rows(k, nxt - r0, nxt - r0 + 1, -1);
            __syncthreads();
            publish(nxt);
        }
        relax_rows(k, 0, rows, own_next ? nxt - r0 : -1);
        __syncthreads();                                              // rowk is rewritten in the next step
    }
    // algos.pyx:48-52 + padding columns
    for (int e = tid; e < rows * N; e += FW_SPLIT_THREADS) {
        const int il = e / N, j = e - il * N, i = r0 + il;
        const int64_t at = (int64_t)i * N + j;
        if (j < n) {
            int v = M[il * np + j];
            if (v >= UNREACH) { v = UNREACH; Pg[at] = UNREACH; }
            Mg[at] = (int16_t)v;
        } else {
            Mg[at] = -1;
            Pg[at] = -1;
        }
    }
}

// one thread per ordered pair: rel_pos and the first D hop features
__global__ __launch_bounds__(256) void edge_path_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ n_nodes,
                                                        const int16_t* __restrict__ spd, const int16_t* __restrict__ path,
                                                        int16_t* __restrict__ rel_pos, uint8_t* __restrict__ edge_input,
                                                        int N, int D) {
    __shared__ int16_t stack[MAXD][256];
    const int g = blockIdx.y;
    const int n = n_nodes[g];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N * N) return;
    const int i = e / N, j = e - i * N;
    const int64_t gbase = (int64_t)g * N * N;
    const int16_t* P = path + gbase;
    const int32_t* C = counts + gbase;
    uint8_t* out = edge_input + (gbase + e) * D;
    const bool real = i < n && j < n;
    rel_pos[gbase + e] = real ? (int16_t)(spd[gbase + e] + 1) : (int16_t)0;      // collator.py:76-83
    int emitted = 0;
    if (real && i != j && P[e] != UNREACH) {                                     // algos.pyx:86-90
        const int tid = threadIdx.x;
        int size = 1, top = 0;                  // ring LIFO of pending targets; stack[top] is the newest
        stack[0][tid] = (int16_t)j;
        int a = i;
        int guard = 4 * n + 2 * D + 8;
        while (emitted < D && size > 0 && guard-- > 0) {
            const int t = stack[top][tid];
            const int k = P[(int64_t)a * N + t];
            if (k == 0) {                       // algos.pyx:59-60: "no intermediate" (or node 0: quirk kept)
                const int c = C[(int64_t)a * N + t];
                // wrapper.py:52 (+2 on edges, 0 elsewhere) then collator.py:87 (+1): count+3 / 1
                out[emitted++] = (uint8_t)(c != 0 ? (c > 252 ? 255 : c + 3) : 1);
                a = t;
                top = top == 0 ? D - 1 : top - 1;
                --size;
            } else {                            // expand (a,t) into (a,k),(k,t): reach k first
                top = top == D - 1 ? 0 : top + 1;
                stack[top][tid] = (int16_t)k;
                if (size < D) ++size;           // full ring: the oldest (deepest) target is overwritten
            }
        }
    }
    for (int d = emitted; d < D; ++d) out[d] = 0;                                // -1 fill, +1 (collator.py:87)
}

}  // namespace

extern "C" int mobgt_spd_set_spin_limit(int64_t ticks_100mhz) {
    const long long v = ticks_100mhz < 0 ? 20000000 : (long long)ticks_100mhz;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fw_spin_ticks), &v, sizeof(v));
}

extern "C" int64_t mobgt_spd_workspace_bytes(int G, int N) {
    // long graphs (fw_split_kernel): one "row published" flag per (graph, row) + one arrival counter per graph
    if (G <= 0 || N <= LDS_M_MAX_N || N > FW_SPLIT_MAX_N) return 16;
    const int64_t npw = ((N + 7) & ~7) / 2;                           // + the published rows, [G][N][npw] words
    return ((int64_t)G * N + G) * (int64_t)sizeof(int) + (int64_t)G * N * npw * 4 + 16;
}

extern "C" int mobgt_spd_batched(const int32_t* counts, const int32_t* n_nodes, int16_t* spd, int16_t* path,
                                 int16_t* rel_pos, uint8_t* edge_input, int16_t* in_degree, int16_t* out_degree,
                                 void* work, int G, int N, int D, void* stream) {
    (void)work;
    if (G <= 0 || N <= 0 || D < 0 || D > MAXD || N > 32000) return MOBGT_EBADDIM;
    hipStream_t st = (hipStream_t)stream;
    if (N <= LDS_M_MAX_N) {
        const size_t shm = (size_t)N * N * sizeof(int16_t);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fw_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS_M_MAX_N * LDS_M_MAX_N * 2);
            attr_set = true;
        }
        hipLaunchKernelGGL(fw_kernel<true>, dim3(G), dim3(FW_THREADS), shm, st, counts, n_nodes, spd, path, in_degree,
                           out_degree, N, nullptr);
    } else if (N <= FW_SPLIT_MAX_N && work != nullptr) {
        int* flags = reinterpret_cast<int*>(work);
        int* gave_up = flags + (int64_t)G * N;
        uint32_t* pub = reinterpret_cast<uint32_t*>(gave_up + G);
        if (hipMemsetAsync(work, 0, ((size_t)G * N + G) * sizeof(int), st) != hipSuccess) return MOBGT_EBADDIM;
        const int rpw = (N + FW_PARTS - 1) / FW_PARTS, np = (N + 7) & ~7;
        const size_t shm = ((size_t)rpw * np + np) * sizeof(int16_t);
        static bool attr_set = false;
        if (!attr_set) {
            // (FW_SPLIT_MAX_N needs 150 144 B; the kernel also has a few bytes of static LDS, so not the full 160 KiB)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(fw_split_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
                (void)hipGetLastError();
            attr_set = true;
        }
        // The workgroups of a graph wait for each other: never launch more than THIS device can hold at once
        // (CU count and the occupancy the runtime computes for this LDS size, queried per call -- the process may
        // drive several devices).  Co-residency can still be lost to other streams; the bounded waits cover that.
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return MOBGT_EBADDIM;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fw_split_kernel),
                                                         FW_SPLIT_THREADS, shm) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();                                   // (do not let the query's status leak into the launches')
            per_cu = shm <= 72 * 1024 ? 2 : 1;                         // LDS-only estimate: 160 KiB per CU
        }
        if (per_cu > 2) per_cu = 2;                                    // (allocation granularity: no tight fits)
        int chunk = (cus * per_cu) / FW_PARTS;
        if (chunk < 1) return MOBGT_EBADDIM;
        for (int g0 = 0; g0 < G; g0 += chunk) {
            const int gc = G - g0 < chunk ? G - g0 : chunk;
            hipLaunchKernelGGL(fw_split_kernel, dim3(gc * FW_PARTS), dim3(FW_SPLIT_THREADS), shm, st, counts, n_nodes, spd,
                               path, in_degree, out_degree, flags, gave_up, pub, N, g0);
        }
        // redo pass: graphs whose workgroups gave up waiting (none in an undisturbed run: G early exits)
        hipLaunchKernelGGL(fw_kernel<false>, dim3(G), dim3(FW_THREADS), 0, st, counts, n_nodes, spd, path, in_degree,
                           out_degree, N, gave_up);
    } else {
        hipLaunchKernelGGL(fw_kernel<false>, dim3(G), dim3(FW_THREADS), 0, st, counts, n_nodes, spd, path, in_degree,
                           out_degree, N, nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (D > 0) {
        const dim3 grid((N * N + 255) / 256, G);
        hipLaunchKernelGGL(edge_path_kernel, grid, dim3(256), 0, st, counts, n_nodes, spd, path, rel_pos, edge_input, N, D);
    } else {
        return MOBGT_EBADDIM;
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The rest of the fq collators' device work in ONE launch (collator.py:57-64 padding mask of attn_bias, :354-358 the
// rel_pos_max cut, :428-437 poi_pos = distance bin of (x_i, x_j) on real pairs): ~10 elementwise / index launches of
// data.DeviceCollator.finish before round 3, which matters now that the collate runs inside every replayed step.
namespace {
__global__ __launch_bounds__(256) void collate_finish_kernel(const int32_t* __restrict__ x, const int32_t* __restrict__ n_nodes,
                                                             const int16_t* __restrict__ spd, const int16_t* __restrict__ bin_table,
                                                             int64_t ld_bin, int rel_pos_max, float* __restrict__ attn_bias,
                                                             int16_t* __restrict__ poi_pos, int G, int N) {
    const int T = N + 1, g = blockIdx.y;
    const int n = n_nodes[g];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)T * T; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e / T), j = (int)(e - (int64_t)i * T);
        float ab = j <= n ? 0.f : -INFINITY;                                   // token 0 + n real nodes are keys
        if (i >= 1 && j >= 1) {
            const int64_t pair = ((int64_t)g * N + (i - 1)) * N + (j - 1);
            if (rel_pos_max <= 510 && spd[pair] >= rel_pos_max) ab = -INFINITY;
            const int xi = x[(int64_t)g * N + (i - 1)], xj = x[(int64_t)g * N + (j - 1)];
            poi_pos[pair] = (bin_table && xi != 0 && xj != 0) ? bin_table[(int64_t)xi * ld_bin + xj] : (int16_t)0;
        }
        attn_bias[(int64_t)g * T * T + e] = ab;
    }
}
}  // namespace

extern "C" int mobgt_collate_finish(const int32_t* x, const int32_t* n_nodes, const int16_t* spd, const int16_t* bin_table,
                                    int64_t ld_bin, int rel_pos_max, float* attn_bias, int16_t* poi_pos, int G, int N,
                                    void* stream) {
    if (G <= 0 || N <= 0) return MOBGT_EBADDIM;
    const int64_t tt = (int64_t)(N + 1) * (N + 1);
    const unsigned bx = (unsigned)((tt + 1023) / 1024 < 1 ? 1 : ((tt + 1023) / 1024 > 1024 ? 1024 : (tt + 1023) / 1024));
    hipLaunchKernelGGL(collate_finish_kernel, dim3(bx, G), dim3(256), 0, (hipStream_t)stream, x, n_nodes, spd, bin_table, ld_bin,
                       rel_pos_max, attn_bias, poi_pos, G, N);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Single-graph entry points with the reference's own call signatures (algos.pyx:9, :65), for the
// per-item drop-in path (wrapper.py:55-60).  Unlike the batched kernels above they take an ARBITRARY
// path matrix / feature tensor and emit up to max_dist hops, as gen_edge_input does.
namespace {

__global__ __launch_bounds__(256) void fw_io_kernel(const int64_t* __restrict__ adj, int32_t* __restrict__ counts, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n * n) counts[e] = adj[e] != 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void widen_kernel(const int16_t* __restrict__ a, const int16_t* __restrict__ b,
                                                    int64_t* __restrict__ A, int64_t* __restrict__ B, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n * n) { A[e] = a[e]; B[e] = b[e]; }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, int64_t n, float v) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) p[e] = v;
}

// one thread per ordered pair; LIFO of pending targets in dynamic LDS, depth = cap (>= max_dist hops)
__global__ __launch_bounds__(64) void edge_paths_generic_kernel(const int64_t* __restrict__ path, const int64_t* __restrict__ feat,
                                                                float* __restrict__ out, int* __restrict__ err,
                                                                int n, int F, int max_dist, int cap) {
    extern __shared__ __attribute__((aligned(16))) int16_t gstack[];   // [cap][64]
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    if (i == j || path[e] == UNREACH) return;                          // algos.pyx:86-90
    const int tid = threadIdx.x;
    int size = 1, top = 0, emitted = 0, a = i;
    gstack[tid] = (int16_t)j;
    long guard = 4L * n + 2L * max_dist + 8;
    float* dst = out + (int64_t)e * max_dist * F;
    while (size > 0 && guard-- > 0) {
        const int t = gstack[top * 64 + tid];
        const int k = (int)path[(int64_t)a * n + t];
        if (k == 0) {
            if (emitted >= max_dist) { atomicExch(err, 1); return; }  // the reference raises IndexError here
            const int64_t* src = feat + ((int64_t)a * n + t) * F;
            for (int f = 0; f < F; ++f) dst[(int64_t)emitted * F + f] = (float)(double)src[f];
            ++emitted;
            a = t;
            top = top == 0 ? cap - 1 : top - 1;
            --size;
        } else {
            top = top == cap - 1 ? 0 : top + 1;
            gstack[top * 64 + tid] = (int16_t)k;
            if (size < cap) ++size; else { atomicExch(err, 2); return; }   // deeper than any valid path
        }
    }
    if (guard <= 0) atomicExch(err, 3);
}

}  // namespace

extern "C" int mobgt_floyd_warshall(const int64_t* adj, int n, int64_t* M, int64_t* path, void* work, void* stream) {
    // work: n*n int32 counts + 2*n*n int16 + n*n int16 + n*n*1 uint8 + 2*n int16 + 4 int32
    if (n <= 0 || n > 32000) return MOBGT_EBADDIM;
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)work;
    const size_t nn = (size_t)n * n;
    int32_t* counts = (int32_t*)w;                w += (nn * 4 + 15) / 16 * 16;
    int16_t* spd = (int16_t*)w;                   w += (nn * 2 + 15) / 16 * 16;
    int16_t* pth = (int16_t*)w;                   w += (nn * 2 + 15) / 16 * 16;
    int16_t* rel = (int16_t*)w;                   w += (nn * 2 + 15) / 16 * 16;
    uint8_t* ei = (uint8_t*)w;                    w += (nn + 15) / 16 * 16;
    int16_t* indeg = (int16_t*)w;                 w += ((size_t)n * 2 + 15) / 16 * 16;
    int16_t* outdeg = (int16_t*)w;                w += ((size_t)n * 2 + 15) / 16 * 16;
    int32_t* nn_dev = (int32_t*)w;
    hipError_t e = hipMemcpyAsync(nn_dev, &n, sizeof(int), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fw_io_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, adj, counts, n);
    int rc = mobgt_spd_batched(counts, nn_dev, spd, pth, rel, ei, indeg, outdeg, nullptr, 1, n, 1, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(widen_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, spd, pth, M, path, n);
    return (int)hipGetLastError();
}

extern "C" int64_t mobgt_floyd_warshall_workspace_bytes(int n) {
    const size_t nn = (size_t)n * n;
    return (int64_t)(nn * 4 + nn * 2 * 3 + nn + (size_t)n * 4 + 256 + 16 * 8);
}

extern "C" int mobgt_gen_edge_input(int max_dist, const int64_t* path, const int64_t* edge_feat, int n, int F,
                                    float* out, int* err_flag, void* stream) {
    if (n <= 0 || F <= 0 || max_dist < 0) return MOBGT_EBADDIM;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)n * n * max_dist * F;
    hipError_t e = hipMemsetAsync(err_flag, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    if (total == 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, total, -1.0f);
    int cap = max_dist + 1 < n + 1 ? max_dist + 1 : n + 1;      // a simple path has at most n-1 hops
    if (cap < 2) cap = 2;
    if (cap > 510) cap = 511;
    const size_t shm = (size_t)cap * 64 * sizeof(int16_t);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edge_paths_generic_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 511 * 64 * 2);
        attr_set = true;
    }
    hipLaunchKernelGGL(edge_paths_generic_kernel, dim3((unsigned)(((int64_t)n * n + 63) / 64)), dim3(64), shm, st, path,
                       edge_feat, out, err_flag, n, F, max_dist, cap);
    return (int)hipGetLastError();
}

namespace {
// algos.pyx:57-62 for ONE pair, iteratively (single lane; this is an API-parity helper, not a hot path)
__global__ void get_all_edges_kernel(const int64_t* __restrict__ path, int n, int i, int j, int32_t* __restrict__ out,
                                     int32_t* __restrict__ out_len, int32_t* __restrict__ stack) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int size = 1, len = 0, a = i;
    stack[0] = j;
    long guard = 8L * n + 16;
    while (size > 0 && guard-- > 0) {
        const int t = stack[size - 1];
        const int k = (int)path[(int64_t)a * n + t];
        if (k == 0) {
            if (size > 1) { if (len < n + 1) out[len] = t; ++len; }   // the bottom target is j itself
            a = t;
            --size;
        } else {
            if (size >= 2 * n + 2) { *out_len = -1; return; }
            stack[size++] = k;
        }
    }
    *out_len = guard <= 0 ? -1 : len;
}
}  // namespace

extern "C" int mobgt_get_all_edges(const int64_t* path, int n, int i, int j, int32_t* out_nodes, int32_t* out_len,
                                   int32_t* work, void* stream) {
    if (n <= 0 || i < 0 || j < 0 || i >= n || j >= n) return MOBGT_EBADDIM;
    hipLaunchKernelGGL(get_all_edges_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, path, n, i, j, out_nodes, out_len, work);
    return (int)hipGetLastError();
}
