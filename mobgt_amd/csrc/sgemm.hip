// f32 GEMMs of the GCN / FuseEmbeddings / head path (graphormer/modelGNN.py:38-44 `torch.mm(input, weight)`,
// `torch.mm(adj, support)` for the 300-node category graph; model_fqandtoyo.py:444-456 FuseEmbeddings' Linear; and the
// data gradients autograd derives from them): a few hundred to 7856 rows, K and N between 16 and 320.  Two dozen such
// calls per step went through the library at 6-15 us each -- kernels tiled for large problems (256x128 tiles for a
// 300 x 64 output) whose time is all ramp-up.  Here ONE wave owns a 16-row x (16 NB)-column output tile and walks K
// with v_mfma_f32_16x16x4_f32 (full f32 products) straight from global memory: no LDS, no barrier, hundreds of
// independent single-wave workgroups.
//   C[M,N] = A[M,K] x B (+ bias[N])     B as [K,N] (x @ W, adj @ support, g @ W)  or as [N,K] (F.linear, g @ W^T)
// Lane (i, kq) supplies A[row i][4kq + s] to MFMA s of a 16-deep k-step (one float4 when rows are 16-byte aligned);
// the B operand uses the same k numbering, so the order of the four MFMAs inside a step is immaterial.
#include <cstdlib>
#include "common.h"
#include "mobgt_hip.h"
#include "sgemm_body.h"

namespace {
using mobgt_sgemm::SgemmParams;
using mobgt_sgemm::act_mask;

using mobgt_sgemm::f32x4;


// 4 consecutive k-values of one row, zero past `kmax`; VEC: the row is 16-byte aligned and fully inside
template <bool VEC>
__device__ __forceinline__ float4 load_k4(const float* row, int k, int kmax) {
    if (VEC) return *reinterpret_cast<const float4*>(row + k);
    float4 v;
    v.x = k < kmax ? row[k] : 0.f;
    v.y = k + 1 < kmax ? row[k + 1] : 0.f;
    v.z = k + 2 < kmax ? row[k + 2] : 0.f;
    v.w = k + 3 < kmax ? row[k + 3] : 0.f;
    return v;
}

// EXT: the activation epilogue / masked-A prologue are compiled in (the plain product keeps its register count and speed)
template <int NB, bool B_NK, bool VEC, bool EXT = false>
__global__ __launch_bounds__(64) void sgemm_kernel(const SgemmParams p) {
    const int lane = threadIdx.x;
    const int i = lane & 15, kq = lane >> 4;
    const int tiles_n = (p.N + 16 * NB - 1) / (16 * NB);
    const int m0 = (blockIdx.x / tiles_n) * 16, n0 = (blockIdx.x % tiles_n) * 16 * NB;
    const float* arow = p.A + (int64_t)min(m0 + i, p.M - 1) * p.lda;
    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int col[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) col[b] = min(n0 + 16 * b + i, p.N - 1);        // clamped: those products are never stored

    auto load_b = [&](int b, int k) -> float4 {
        if (B_NK) return load_k4<VEC>(p.B + (int64_t)col[b] * p.ldb, k, p.K);
        float4 v;
        const float* q = p.B + col[b];
        v.x = k < p.Kb ? q[(int64_t)k * p.ldb] : 0.f;
        v.y = k + 1 < p.Kb ? q[(int64_t)(k + 1) * p.ldb] : 0.f;
        v.z = k + 2 < p.Kb ? q[(int64_t)(k + 2) * p.ldb] : 0.f;
        v.w = k + 3 < p.Kb ? q[(int64_t)(k + 3) * p.ldb] : 0.f;
        return v;
    };

    const float* mrow = (EXT && p.amask) ? p.amask + (int64_t)min(m0 + i, p.M - 1) * p.lda : nullptr;
    auto load_a = [&](int k) -> float4 {
        float4 v = load_k4<VEC>(arow, k, p.K);
        if (EXT && mrow) {
            const float4 y = load_k4<VEC>(mrow, k, p.K);
            v.x *= act_mask(y.x, p.mpos, p.mneg, p.mzero); v.y *= act_mask(y.y, p.mpos, p.mneg, p.mzero);
            v.z *= act_mask(y.z, p.mpos, p.mneg, p.mzero); v.w *= act_mask(y.w, p.mpos, p.mneg, p.mzero);
        }
        return v;
    };
    // operands of TWO steps are in flight ahead of the one being multiplied (a ring of three register slots: a wave is alone
    // on its tile, and with one step of lookahead every 16-deep step paid most of a memory round trip -- 12-19 of them per
    // product at K = 192-304)
    constexpr int D = 3;
    float4 a_r[D], b_r[D][NB];
#pragma unroll
    for (int u = 0; u < D - 1; ++u) {
        const int k = 16 * u + 4 * kq;
        const bool live = 16 * u < p.K;
        a_r[u] = live ? load_a(k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int b = 0; b < NB; ++b) b_r[u][b] = live ? load_b(b, k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int k0 = 0; k0 < p.K; k0 += 16 * D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int kc = k0 + 16 * u;
            if (kc < p.K) {
                const int kp = kc + 16 * (D - 1);
                const int slot = (u + D - 1) % D;
                if (kp < p.K) {
                    a_r[slot] = load_a(kp + 4 * kq);
#pragma unroll
                    for (int b = 0; b < NB; ++b) b_r[slot][b] = load_b(b, kp + 4 * kq);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r[u].x, b_r[u][b].x, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r[u].y, b_r[u][b].y, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r[u].z, b_r[u][b].z, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r[u].w, b_r[u][b].w, acc[b], 0, 0, 0);
                }
            }
        }
    }
    // register v of lane (j = lane & 15, q = lane >> 4) is output row 4q + v, column j of its 16x16 block
    const uint64_t seed = (EXT && p.thr) ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int c = n0 + 16 * b + i;
        if (c >= p.N) continue;
        const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = m0 + 4 * kq + v;
            if (r < p.M) {
                float o = acc[b][v] + bv;
                if (EXT && p.act) {
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
                if (EXT && p.ct) p.ct[(int64_t)c * p.ldt + r] = (bf16_t)(p.ct_scale ? o * p.ct_scale[r] : o);
            }
        }
    }
}

// Tall-and-narrow products with a long contraction: split over KS waves per 16-row tile (sgemm_body.h)
template <int KS>
__global__ __launch_bounds__(64 * KS) void sgemm_splitk_kernel(const SgemmParams p) {
    __shared__ float part[KS][16 * 17];
    mobgt_sgemm::sgemm_splitk_tile<KS>(p, blockIdx.x, part);
}

template <int NB, bool EXT>
int launch_x(const SgemmParams& p, bool b_nk, bool vec, hipStream_t st) {
    const int tiles = ((p.M + 15) / 16) * ((p.N + 16 * NB - 1) / (16 * NB));
    const dim3 grid(tiles), block(64);
    if (b_nk) {
        if (vec) hipLaunchKernelGGL((sgemm_kernel<NB, true, true, EXT>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((sgemm_kernel<NB, true, false, EXT>), grid, block, 0, st, p);
    } else {
        if (vec) hipLaunchKernelGGL((sgemm_kernel<NB, false, true, EXT>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((sgemm_kernel<NB, false, false, EXT>), grid, block, 0, st, p);
    }
    return (int)hipGetLastError();
}

template <int NB>
int launch(const SgemmParams& p, bool b_nk, bool vec, hipStream_t st) {
    return (p.act || p.amask || p.ct) ? launch_x<NB, true>(p, b_nk, vec, st) : launch_x<NB, false>(p, b_nk, vec, st);
}

}  // namespace

namespace {
int run(SgemmParams& p, int b_is_nk, int c_dtype, void* stream) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return MOBGT_EBADDIM;
    if (p.Kb <= 0) p.Kb = p.K;
    if (p.Kb > p.K || (b_is_nk && p.Kb != p.K)) return MOBGT_EBADDIM;
    if (c_dtype != MOBGT_F32 && c_dtype != MOBGT_BF16) return MOBGT_EDTYPE;
    if ((((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.amask) & 3) || ((uintptr_t)p.C & (c_dtype == MOBGT_F32 ? 3 : 1))) return MOBGT_EALIGN;
    if (!p.C && !p.ct) return MOBGT_EBADDIM;
    p.c_bf16 = c_dtype == MOBGT_BF16;
    // float4 operand loads: every row 16-byte aligned and K a whole number of 16-deep steps
    const bool vec = (p.K % 16 == 0) && (p.lda % 4 == 0) && ((((uintptr_t)p.A | (uintptr_t)p.amask) & 15) == 0) &&
                     (!b_is_nk || ((p.ldb % 4 == 0) && (((uintptr_t)p.B & 15) == 0)));
    hipStream_t st = (hipStream_t)stream;
    // columns per wave: wide tiles reuse the A operand, narrow ones give more waves; aim at >= ~256 waves
    const int rows16 = (p.M + 15) / 16;
    constexpr bool no_splitk = false;
    if (!no_splitk && !b_is_nk && vec && p.N <= 16 && p.K >= 128 && p.K <= 16 * 4 * 8 && rows16 >= 64) {
        hipLaunchKernelGGL(sgemm_splitk_kernel<4>, dim3(rows16), dim3(256), 0, st, p);
        return (int)hipGetLastError();
    }
    int nb = 4;
    while (nb > 1 && (rows16 * ((p.N + 16 * nb - 1) / (16 * nb)) < 256 || p.N <= 16 * (nb / 2))) nb >>= 1;
    if (nb == 4) return launch<4>(p, b_is_nk != 0, vec, st);
    if (nb == 2) return launch<2>(p, b_is_nk != 0, vec, st);
    return launch<1>(p, b_is_nk != 0, vec, st);
}
}  // namespace

extern "C" int mobgt_small_gemm_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int b_is_nk, const float* bias,
                                    void* c, int64_t ldc, int c_dtype, int M, int N, int K, void* stream) {
    SgemmParams p = {};
    p.A = a; p.lda = lda; p.B = b; p.ldb = ldb; p.bias = bias; p.C = c; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    return run(p, b_is_nk, c_dtype, stream);
}

/* The same product with the activation that follows it (epilogue) and / or the derivative of the activation that precedes
 * it (prologue on A) -- see include/mobgt_hip.h. */
extern "C" int mobgt_small_gemm_f32_act(const float* a, int64_t lda, const float* a_mask, float m_pos, float m_neg, float m_zero,
                                        const float* b, int64_t ldb, int b_is_nk, const float* bias, int leaky, float slope,
                                        float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* c,
                                        int64_t ldc, int c_dtype, void* c_t_bf16, int64_t ld_t, const float* c_t_scale, int M,
                                        int N, int K, int k_b, void* stream) {
    SgemmParams p = {};
    p.Kb = k_b;
    if (c_t_bf16 && ld_t < M) return MOBGT_EBADDIM;
    p.ct = reinterpret_cast<bf16_t*>(c_t_bf16); p.ldt = ld_t; p.ct_scale = c_t_scale;
    p.A = a; p.lda = lda; p.B = b; p.ldb = ldb; p.bias = bias; p.C = c; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.amask = a_mask; p.mpos = m_pos; p.mneg = m_neg; p.mzero = m_zero;
    p.act = leaky; p.slope = slope;
    if (leaky && dropout_p > 0.f) {
        p.thr = dropout_threshold(dropout_p);
        p.inv_keep = 1.f / (1.f - (float)p.thr / 65536.f);
        p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
    }
    return run(p, b_is_nk, c_dtype, stream);
}
