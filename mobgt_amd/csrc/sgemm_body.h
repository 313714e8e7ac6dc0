// The small f32 GEMM's parameter block and its split-K tile body (see sgemm.hip), shared with bias.hip: the bias assembly's
// launch carries the distance GCN's first layer as extra workgroups (round 4).
#pragma once
#include "common.h"
#include "mobgt_hip.h"

namespace mobgt_sgemm {

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct SgemmParams {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    const float* bias;
    void* C; int64_t ldc;
    int c_bf16;                                  // store the result as bf16 (the adjacency product's operand) instead of f32
    int M, N, K;
    // optional epilogue: y = dropout(leaky_relu(acc + bias)) -- GraphConvolution / FuseEmbeddings' activation (modelGNN.py:
    // 66-72, model_fqandtoyo.py:452-455) without a launch of its own; the mask is mobgt_bias_act_fwd's
    int act;
    float slope;
    uint32_t thr;
    float inv_keep;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt;
    // optional prologue on A: A[r][k] *= m(amask[r][k]), m(y) = y > 0 ? mpos : (y < 0 ? mneg : mzero) -- the derivative
    // of that activation applied to an incoming gradient while it is loaded (amask = the activation's OUTPUT, ld = lda)
    const float* amask;
    float mpos, mneg, mzero;
    // optional: the result (times ct_scale[row]) ALSO / ONLY (C null) as bf16 transposed [N][ldt] -- the operand layout of the
    // bitmask adjacency product (csrc/maskgemm.hip), so that no transpose launch stands between the two
    bf16_t* ct;
    int64_t ldt;
    const float* ct_scale;
    int Kb;                                      // rows of a [K,N] B that exist (<= K: A zero-padded to a whole number of k-steps)
};

__device__ __forceinline__ float act_mask(float y, float pos, float neg, float zer) { return y > 0.f ? pos : (y < 0.f ? neg : zer); }

// Tall-and-narrow products with a long contraction (the distance GCN's first layer: 7 856 x 304 x 16): one wave per 16-row
// tile walks 19 dependent 16-deep steps (9.2 us in the S-FSQ step, a chain of load round trips).  Here KS waves share a tile,
// wave w taking the steps w, w + KS, ...; the partial tiles meet in LDS and wave 0 runs the epilogue.  [K,N] operand, N <= 16,
// rows 16-byte aligned, K % 16 == 0.
template <int KS>
__device__ __forceinline__ void sgemm_splitk_tile(const SgemmParams& p, const int tile, float (*part)[16 * 17]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int m0 = tile * 16;
    const float* arow = p.A + (int64_t)min(m0 + i, p.M - 1) * p.lda;
    const float* mrow = p.amask ? p.amask + (int64_t)min(m0 + i, p.M - 1) * p.lda : nullptr;
    const int col = min(i, p.N - 1);
    const int nsteps = p.K / 16;
    constexpr int MAXS = 8;                                   // steps per wave held in flight (K <= 16 * KS * MAXS)
    float4 a_r[MAXS], b_r[MAXS];
#pragma unroll
    for (int u = 0; u < MAXS; ++u) {
        const int s = wave + KS * u, k = 16 * s + 4 * kq;
        if (s < nsteps) {
            float4 v = *reinterpret_cast<const float4*>(arow + k);
            if (mrow) {
                const float4 y = *reinterpret_cast<const float4*>(mrow + k);
                v.x *= act_mask(y.x, p.mpos, p.mneg, p.mzero); v.y *= act_mask(y.y, p.mpos, p.mneg, p.mzero);
                v.z *= act_mask(y.z, p.mpos, p.mneg, p.mzero); v.w *= act_mask(y.w, p.mpos, p.mneg, p.mzero);
            }
            a_r[u] = v;
            const float* q = p.B + col;
            b_r[u].x = k < p.Kb ? q[(int64_t)k * p.ldb] : 0.f;
            b_r[u].y = k + 1 < p.Kb ? q[(int64_t)(k + 1) * p.ldb] : 0.f;
            b_r[u].z = k + 2 < p.Kb ? q[(int64_t)(k + 2) * p.ldb] : 0.f;
            b_r[u].w = k + 3 < p.Kb ? q[(int64_t)(k + 3) * p.ldb] : 0.f;
        } else {
            a_r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            b_r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < MAXS; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r[u].x, b_r[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r[u].y, b_r[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r[u].z, b_r[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r[u].w, b_r[u].w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) part[wave][(4 * kq + v) * 17 + i] = acc[v];
    __syncthreads();
    if (wave == 0) {
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    const int c = i;
    const float bv = (p.bias && c < p.N) ? p.bias[c] : 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int r = m0 + 4 * kq + v;
        if (r >= p.M || c >= p.N) continue;
        float o = bv;
#pragma unroll
        for (int w = 0; w < KS; ++w) o += part[w][(4 * kq + v) * 17 + i];
        if (p.act) {
            o = o > 0.f ? o : p.slope * o;
            if (p.thr) {
                const uint32_t rowh = dropout_row_hash(seed, (uint32_t)r ^ p.salt);
                o = dropout_bits16(seed, rowh, (uint32_t)c) >= p.thr ? o * p.inv_keep : 0.f;
            }
        }
        if (p.C) {
            if (p.c_bf16) reinterpret_cast<bf16_t*>(p.C)[(int64_t)r * p.ldc + c] = (bf16_t)o;
            else reinterpret_cast<float*>(p.C)[(int64_t)r * p.ldc + c] = o;
        }
        if (p.ct) p.ct[(int64_t)c * p.ldt + r] = (bf16_t)(p.ct_scale ? o * p.ct_scale[r] : o);
    }
    }
    __syncthreads();                                   // (the partial tiles are free again: a workgroup may run tile after tile)
}

}  // namespace mobgt_sgemm
