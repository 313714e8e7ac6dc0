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
#include "common.h"
#include "mobgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct SgemmParams {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    const float* bias;
    void* C; int64_t ldc;
    int c_bf16;                                  // store the result as bf16 (the adjacency product's operand) instead of f32
    int M, N, K;
};

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

template <int NB, bool B_NK, bool VEC>
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
        v.x = k < p.K ? q[(int64_t)k * p.ldb] : 0.f;
        v.y = k + 1 < p.K ? q[(int64_t)(k + 1) * p.ldb] : 0.f;
        v.z = k + 2 < p.K ? q[(int64_t)(k + 2) * p.ldb] : 0.f;
        v.w = k + 3 < p.K ? q[(int64_t)(k + 3) * p.ldb] : 0.f;
        return v;
    };

    float4 a_cur = load_k4<VEC>(arow, 4 * kq, p.K), a_nxt = a_cur;
    float4 b_cur[NB], b_nxt[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { b_cur[b] = load_b(b, 4 * kq); b_nxt[b] = b_cur[b]; }
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        const int kn = k0 + 16 + 4 * kq;
        if (k0 + 16 < p.K) {
            a_nxt = load_k4<VEC>(arow, kn, p.K);
#pragma unroll
            for (int b = 0; b < NB; ++b) b_nxt[b] = load_b(b, kn);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.x, b_cur[b].x, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.y, b_cur[b].y, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.z, b_cur[b].z, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.w, b_cur[b].w, acc[b], 0, 0, 0);
        }
        a_cur = a_nxt;
#pragma unroll
        for (int b = 0; b < NB; ++b) b_cur[b] = b_nxt[b];
    }
    // register v of lane (j = lane & 15, q = lane >> 4) is output row 4q + v, column j of its 16x16 block
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int c = n0 + 16 * b + i;
        if (c >= p.N) continue;
        const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = m0 + 4 * kq + v;
            if (r < p.M) {
                if (p.c_bf16) reinterpret_cast<bf16_t*>(p.C)[(int64_t)r * p.ldc + c] = (bf16_t)(acc[b][v] + bv);
                else reinterpret_cast<float*>(p.C)[(int64_t)r * p.ldc + c] = acc[b][v] + bv;
            }
        }
    }
}

template <int NB>
int launch(const SgemmParams& p, bool b_nk, bool vec, hipStream_t st) {
    const int tiles = ((p.M + 15) / 16) * ((p.N + 16 * NB - 1) / (16 * NB));
    const dim3 grid(tiles), block(64);
    if (b_nk) {
        if (vec) hipLaunchKernelGGL((sgemm_kernel<NB, true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((sgemm_kernel<NB, true, false>), grid, block, 0, st, p);
    } else {
        if (vec) hipLaunchKernelGGL((sgemm_kernel<NB, false, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((sgemm_kernel<NB, false, false>), grid, block, 0, st, p);
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mobgt_small_gemm_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int b_is_nk, const float* bias,
                                    void* c, int64_t ldc, int c_dtype, int M, int N, int K, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return MOBGT_EBADDIM;
    if (c_dtype != MOBGT_F32 && c_dtype != MOBGT_BF16) return MOBGT_EDTYPE;
    if ((((uintptr_t)a | (uintptr_t)b) & 3) || ((uintptr_t)c & (c_dtype == MOBGT_F32 ? 3 : 1))) return MOBGT_EALIGN;
    SgemmParams p = {a, lda, b, ldb, bias, c, ldc, c_dtype == MOBGT_BF16, M, N, K};
    // float4 operand loads: every row 16-byte aligned and K a whole number of 16-deep steps
    const bool vec = (K % 16 == 0) && (lda % 4 == 0) && (((uintptr_t)a & 15) == 0) &&
                     (!b_is_nk || ((ldb % 4 == 0) && (((uintptr_t)b & 15) == 0)));
    hipStream_t st = (hipStream_t)stream;
    // columns per wave: wide tiles reuse the A operand, narrow ones give more waves; aim at >= ~256 waves
    const int rows16 = (M + 15) / 16;
    int nb = 4;
    while (nb > 1 && (rows16 * ((N + 16 * nb - 1) / (16 * nb)) < 256 || N <= 16 * (nb / 2))) nb >>= 1;
    if (nb == 4) return launch<4>(p, b_is_nk != 0, vec, st);
    if (nb == 2) return launch<2>(p, b_is_nk != 0, vec, st);
    return launch<1>(p, b_is_nk != 0, vec, st);
}
