// The encoder layer's eight small GEMMs (graphormer/model.py:388-403 FeedForwardNetwork, :406-463
// MultiHeadAttention's linear_q/k/v and output_layer; model_fqandtoyo.py:1641-1712) and their data gradients, bf16
// operands / f32 accumulate, with the elementwise step that follows each of them fused into the epilogue.
//
// Shape of the problem: M = G*T rows (a few hundred), N and K in {C, 3C, F} = {192, 576, 1024}.  134 MFLOP and
// ~1 MB per call: nothing but latency.  A library GEMM picks (per shape, by noisy timing) among kernels that walk the
// whole of K in one wave -- 5 us when the pick is good, 9-21 us when it is not (K = 1024, or a non-MFMA kernel for
// 608x192x192), and the elementwise step (GELU, its derivative, the residual add) is one more launch.  Here a workgroup
// owns a 32x64 output tile, its four waves SPLIT K (wave w takes k-steps w, w+4, ...), partial tiles meet in LDS
// and every thread finishes 8 adjacent outputs of one row: bias, GELU (u and h from one pass), GELU' * acc, or
// acc + addend, stored as one 16- or 32-byte vector.
//
// Operands come straight from global memory (everything is L2 / MALL resident), one 16-byte load per MFMA operand:
//   A [M,K] row-major: lane (i, kq) of v_mfma_f32_16x16x32_bf16 supplies A[row i][8kq .. 8kq+7].
//   B as [N,K] (nn.Linear weight, forward):  the same for column j.
//   B as [K,N] (the weight seen from the backward pass, dX = dY W): the 8 k-values of a lane live in 8 rows; the lane
//     reads one dword = columns (2j, 2j+1) of each row and splits low / high halves into an even-column and an
//     odd-column operand (v_perm_b32) -- no transposed copy of the weights, no LDS transpose.
#include "common.h"
#include "mobgt_hip.h"
#include "gemm_body.h"

namespace {

using namespace mobgt_gemm;

template <bool BKN, int EPI, int NB, int NW>
__global__ __launch_bounds__(NW * 64) void layer_gemm_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float lds[gemm_lds_floats<NB, NW>()];
    layer_gemm_body<BKN, EPI, NB, NW>(p, blockIdx.x, lds);
}


template <bool BKN, int NB, int NW>
int launch(const GemmParams& p, int epilogue, hipStream_t st) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + 16 * NB - 1) / (16 * NB));
    const dim3 grid(tiles), block(NW * 64);
    switch (epilogue) {
        case EPI_BIAS: hipLaunchKernelGGL((layer_gemm_kernel<BKN, EPI_BIAS, NB, NW>), grid, block, 0, st, p); break;
        case EPI_GELU: hipLaunchKernelGGL((layer_gemm_kernel<BKN, EPI_GELU, NB, NW>), grid, block, 0, st, p); break;
        case EPI_GELU_BWD: hipLaunchKernelGGL((layer_gemm_kernel<BKN, EPI_GELU_BWD, NB, NW>), grid, block, 0, st, p); break;
        case EPI_ADD: hipLaunchKernelGGL((layer_gemm_kernel<BKN, EPI_ADD, NB, NW>), grid, block, 0, st, p); break;
        default: return MOBGT_EBADDIM;
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mobgt_layer_gemm(const void* a, int64_t lda, const void* b, int64_t ldb, int b_is_kn, const void* bias,
                                void* c, int64_t ldc, int epilogue, const void* aux_in, void* aux_out, int M, int N, int K,
                                void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % KSTEP) || (N & 7) || (lda & 7) || (ldc & 7)) return MOBGT_EBADDIM;
    if (b_is_kn ? (ldb & 1) : (ldb & 7)) return MOBGT_EBADDIM;
    if (((uintptr_t)a | (uintptr_t)c | (uintptr_t)bias | (uintptr_t)aux_in | (uintptr_t)aux_out) & 15) return MOBGT_EALIGN;
    if ((uintptr_t)b & (b_is_kn ? 3 : 15)) return MOBGT_EALIGN;
    if (epilogue == EPI_GELU && !aux_out) return MOBGT_EBADDIM;
    if ((epilogue == EPI_GELU_BWD || epilogue == EPI_ADD) && !aux_in) return MOBGT_EBADDIM;
    GemmParams p = {reinterpret_cast<const uint16_t*>(a), lda, reinterpret_cast<const uint16_t*>(b), ldb,
                    reinterpret_cast<const uint16_t*>(bias), c, ldc, aux_in, reinterpret_cast<uint16_t*>(aux_out), M, N, K};
    // few 32x64 tiles AND a long K: narrower tiles, twice the waves on K
    const int tiles64 = ((M + BM - 1) / BM) * ((N + 63) / 64);
    const bool narrow = K >= 512 && tiles64 < 128;
    hipStream_t st = (hipStream_t)stream;
    if (narrow) return b_is_kn ? launch<true, 2, 8>(p, epilogue, st) : launch<false, 2, 8>(p, epilogue, st);
    if (tiles64 < 128) return b_is_kn ? launch<true, 2, 4>(p, epilogue, st) : launch<false, 2, 4>(p, epilogue, st);
    return b_is_kn ? launch<true, 4, 4>(p, epilogue, st) : launch<false, 4, 4>(p, epilogue, st);
}
