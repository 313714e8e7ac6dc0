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
// per output each summing every fourth edge id.  Blocks [D, ..): d_enc, one thread per (e, k), the 8 heads of g[d,e,:]
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
        if (act) {
            // (16-byte loads, four of each array in flight per thread before the first LDS store: written as one load and one
            //  store per element the loop was a chain of dependent round trips -- 31.6 us at E = 1 537, the stock variant)
            const float4* g4 = reinterpret_cast<const float4*>(p.dtab + (int64_t)d * E * H);
            const float4* e4 = reinterpret_cast<const float4*>(p.enc);
            const int n4 = E * H / 4;
            for (int t0 = 0; t0 < n4; t0 += 256 * 4) {
                float4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + 256 * u + (int)threadIdx.x;
                    a[u] = t < n4 ? g4[t] : make_float4(0.f, 0.f, 0.f, 0.f);
                    b[u] = t < n4 ? e4[t] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + 256 * u + (int)threadIdx.x;
                    if (t < n4) {
                        reinterpret_cast<float4*>(sg)[t] = make_float4(r16(a[u].x, rt), r16(a[u].y, rt), r16(a[u].z, rt), r16(a[u].w, rt));
                        reinterpret_cast<float4*>(se)[t] = make_float4(r16(b[u].x, rt), r16(b[u].y, rt), r16(b[u].z, rt), r16(b[u].w, rt));
                    }
                }
            }
        }
        __syncthreads();
        if (!act) return;
        const int o = threadIdx.x >> 2, part = threadIdx.x & 3, k = o >> 3, h = o & 7;       // 64 outputs x 4 parts
        // (the four parts take interleaved edge ids: their LDS reads of one iteration are 32 consecutive floats.  Contiguous
        //  quarters put all four on the same banks whenever E / 4 * 8 is a multiple of the bank count)
        // (eight iterations' LDS reads in flight: one read pair per dependent add was ~100 cycles x 385 iterations at E = 1 537)
        float acc = 0.f;
        int e = part;
        for (; e + 28 < E; e += 32) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[u] = se[(e + 4 * u) * H + k]; b[u] = sg[(e + 4 * u) * H + h]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += a[u] * b[u];
        }
        for (; e < E; e += 4) acc += se[e * H + k] * sg[e * H + h];
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
        // (four hop slots' loads in flight at a time: with a run-time D the plain loop waited for every slot's rows in turn --
        //  20 dependent round trips, most of the kernel's 19 us at E = 1 537; same order of additions)
        for (int d0 = 0; d0 < D; d0 += 4) {
            float4 ga[4], gb[4], wa[4], wb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int d = d0 + u < D ? d0 + u : D - 1;
                const float4* g4 = reinterpret_cast<const float4*>(p.dtab + ((int64_t)d * E + e) * H);
                const float4* w4 = reinterpret_cast<const float4*>(p.w + ((int64_t)d * H + k) * H);
                ga[u] = g4[0]; gb[u] = g4[1]; wa[u] = w4[0]; wb[u] = w4[1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (d0 + u < D)
                    acc += r16(ga[u].x, rt) * r16(wa[u].x, rt) + r16(ga[u].y, rt) * r16(wa[u].y, rt) + r16(ga[u].z, rt) * r16(wa[u].z, rt) +
                           r16(ga[u].w, rt) * r16(wa[u].w, rt) + r16(gb[u].x, rt) * r16(wb[u].x, rt) + r16(gb[u].y, rt) * r16(wb[u].y, rt) +
                           r16(gb[u].z, rt) * r16(wb[u].z, rt) + r16(gb[u].w, rt) * r16(wb[u].w, rt);
            }
        }
    }
    p.d_enc[i] = r16(acc, rt);
}

}  // namespace mobgt_hop
