// The hop table's backward for H = 8 as a device body (csrc/hop.hip launches it on its own; csrc/wgrad.hip carries it as extra
// workgroups of the step's grouped weight-gradient launch).
#pragma once
#include "common.h"

namespace mobgt_hop {

__device__ __forceinline__ float r16(float v, bool on) { return on ? (float)(_Float16)v : v; }

struct HopBwd {
    const float *dtab, *enc, *w;             // d(table) [D, E, 8], edge_encoder [E, 8], edge_dis_encoder [>= D, 8, 8]
    float *d_enc, *d_w;                      // written
    int D, E, rt;
};
inline __host__ __device__ int hop_bwd8_blocks(int D, int E) { return D + (E * 8 + 255) / 256; }

// H = 8 (every MobGT config).  Blocks [0, D): d_w of hop slot d -- g[d] and enc staged in LDS (sm: 2 E 8 floats), 4 threads
// per output each summing a quarter of the edge ids.  Blocks [D, ..): d_enc, one thread per (e, k), the 8 heads of g[d,e,:]
// and W[d,k,:] as two 16-byte loads each.  (The generic kernel's dependent scalar loads took 30 us.)  Threads >= 256 of a
// wider workgroup idle (they still reach the barrier).
__device__ __forceinline__ void hop_table_bwd8_body(const HopBwd& p, const int bid, float* __restrict__ sm) {
    constexpr int H = 8;
    const int D = p.D, E = p.E, rt = p.rt;
    const bool act = threadIdx.x < 256;
    if (bid < D) {
        const int d = bid;
        float* sg = sm;
        float* se = sm + E * H;
        if (act)
            for (int t = threadIdx.x; t < E * H; t += 256) { sg[t] = r16(p.dtab[(int64_t)d * E * H + t], rt); se[t] = r16(p.enc[t], rt); }
        __syncthreads();
        if (!act) return;
        const int o = threadIdx.x >> 2, part = threadIdx.x & 3, k = o >> 3, h = o & 7;       // 64 outputs x 4 parts
        const int e0 = (E * part) / 4, e1 = (E * (part + 1)) / 4;
        float acc = 0.f;
        for (int e = e0; e < e1; ++e) acc += se[e * H + k] * sg[e * H + h];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (part == 0) p.d_w[(d * H + k) * H + h] = r16(acc, rt);
        return;
    }
    if (!act) return;
    const int i = (bid - D) * 256 + threadIdx.x;
    if (i >= E * H) return;
    const int k = i & 7, e = i >> 3;
    float acc = 0.f;
    if (e != 0) {
        for (int d = 0; d < D; ++d) {
            const float4* g4 = reinterpret_cast<const float4*>(p.dtab + ((int64_t)d * E + e) * H);
            const float4* w4 = reinterpret_cast<const float4*>(p.w + ((int64_t)d * H + k) * H);
            const float4 ga = g4[0], gb = g4[1], wa = w4[0], wb = w4[1];
            acc += r16(ga.x, rt) * r16(wa.x, rt) + r16(ga.y, rt) * r16(wa.y, rt) + r16(ga.z, rt) * r16(wa.z, rt) +
                   r16(ga.w, rt) * r16(wa.w, rt) + r16(gb.x, rt) * r16(wb.x, rt) + r16(gb.y, rt) * r16(wb.y, rt) +
                   r16(gb.z, rt) * r16(wb.z, rt) + r16(gb.w, rt) * r16(wb.w, rt);
        }
    }
    p.d_enc[i] = r16(acc, rt);
}

}  // namespace mobgt_hop
