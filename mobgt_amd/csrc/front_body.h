// Two tiny front-of-step kernels as device bodies: the index derivation of the node features (csrc/embed.hip:
// node_index_kernel) and the hop table's forward (csrc/hop.hip).  Each is 4.8 us as a launch of its own -- all ramp -- and both
// ride in the category GCN's forward launch as passenger workgroups (csrc/smallgcn.hip).
#pragma once
#include "common.h"
#include "mobgt_hip.h"

namespace mobgt_front {

struct NodeIndexParams {
    const void* x; int x_dtype; int64_t xs_g, xs_n; // POI ids [G,N] (0 = pad; int64 or int32), element strides
    const float* tn;  int64_t ts_g, ts_n;           // time_normal [G,N]
    const int64_t* poi2cat;                          // [P+1]
    const void *indeg, *outdeg;                      // [G*N] degrees (deg_dtype), or null
    int deg_dtype;
    int64_t* idx;                                    // [8][G*N]
    float* real;                                     // [G*N]
    int G, N, rows_only;
};

// one workgroup (256 threads) per graph g: count the real nodes (the positional rows stop there), then write all index rows and
// the mask for its N positions.  s_cnt: 4 ints of LDS.
__device__ __forceinline__ void node_index_body(const NodeIndexParams& p, const int g, int* s_cnt) {
    int cnt = 0;
    const bool x32 = p.x_dtype == MOBGT_I32;
    auto poi_at = [&](int n) -> int64_t {
        const int64_t o = g * p.xs_g + n * p.xs_n;
        return x32 ? (int64_t)reinterpret_cast<const int32_t*>(p.x)[o] : reinterpret_cast<const int64_t*>(p.x)[o];
    };
    for (int n = threadIdx.x; n < p.N; n += 256) cnt += poi_at(n) != 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const int64_t GN = (int64_t)p.G * p.N;
    for (int n = threadIdx.x; n < p.N; n += 256) {
        const int64_t r = (int64_t)g * p.N + n;
        const int64_t poi = poi_at(n);
        const bool real = poi != 0;
        const int64_t slot = (int64_t)(p.tn[g * p.ts_g + n * p.ts_n] * 48.f);
        p.idx[0 * GN + r] = real ? (p.rows_only ? r : poi - 1) : -1;
        p.idx[1 * GN + r] = real ? slot : -1;
        p.idx[2 * GN + r] = real ? p.poi2cat[poi] - 1 : -1;
        p.idx[3 * GN + r] = real && n + 1 <= cnt ? n + 1 : -1;
        p.idx[4 * GN + r] = poi > 0 ? poi - 1 : 0;
        p.idx[5 * GN + r] = 0;
        if (p.indeg) {
            int64_t a, b;
            if (p.deg_dtype == MOBGT_I16) { a = reinterpret_cast<const int16_t*>(p.indeg)[r]; b = reinterpret_cast<const int16_t*>(p.outdeg)[r]; }
            else if (p.deg_dtype == MOBGT_I32) { a = reinterpret_cast<const int32_t*>(p.indeg)[r]; b = reinterpret_cast<const int32_t*>(p.outdeg)[r]; }
            else { a = reinterpret_cast<const int64_t*>(p.indeg)[r]; b = reinterpret_cast<const int64_t*>(p.outdeg)[r]; }
            p.idx[6 * GN + r] = a;
            p.idx[7 * GN + r] = b;
        }
        p.real[r] = real ? 1.f : 0.f;
    }
}

struct HopFwd {
    const float *enc, *w;                    // edge_encoder [E, H], edge_dis_encoder [>= D, H, H]
    float* tab;                              // [D, E, H]
    int D, E, H, rt;
};
__device__ __forceinline__ float hop_r16(float v, bool on) { return on ? (float)(_Float16)v : v; }
// virtual block `bid` of 256 threads: entries bid * 256 .. of T[d, e, h] = sum_k enc[e, k] * w[d, k, h]
__device__ __forceinline__ void hop_table_fwd_body(const HopFwd& p, const int bid) {
    const int i = bid * 256 + threadIdx.x;
    if (threadIdx.x >= 256 || i >= p.D * p.E * p.H) return;
    const int H = p.H, h = i % H, e = (i / H) % p.E, d = i / (H * p.E);
    float acc = 0.f;
    for (int k = 0; k < H; ++k) acc += hop_r16(p.enc[e * H + k], p.rt) * hop_r16(p.w[(d * H + k) * H + h], p.rt);
    p.tab[i] = hop_r16(acc, p.rt);
}

}  // namespace mobgt_front
