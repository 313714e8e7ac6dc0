// Body of the encoder layer's split-K MFMA GEMM (see gemm.hip for the design notes), shared by gemm.hip and by the
// grouped weight-gradient launch of wgrad.hip (which runs the layer's last data-gradient GEMM in the same launch).
#pragma once
#include "common.h"

namespace mobgt_gemm {

constexpr int BM = 32, KSTEP = 32;
// Tilings: 32x64 outputs / 4 waves; when that gives < 128 workgroups (N = C: output projection, FFN layer 2 and their
// data gradients), 32x32 outputs with 4 waves (K < 512) or 8 waves (K >= 512: 8 serial k-steps per wave otherwise).
// NB = number of 16-column MFMA operands.
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_GELU_BWD = 2, EPI_ADD = 3 };

struct GemmParams {
    const uint16_t* A; int64_t lda;
    const uint16_t* B; int64_t ldb;
    const uint16_t* bias;                        // [N] bf16 or null
    void* C; int64_t ldc;                        // bf16, or f32 for EPI_ADD
    const void* aux_in;                          // EPI_GELU_BWD: u [M,N] bf16 (ld = ldc); EPI_ADD: addend [M,N] f32 (ld = ldc)
    uint16_t* aux_out;                           // EPI_GELU: h = gelu(u) [M,N] bf16 (ld = ldc)
    int M, N, K;
};

__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float u) {
    const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * u * u);
    return cdf + u * pdf;
}

__device__ __forceinline__ void split_pairs(const uint32_t (&d)[8], bf16x8& even, bf16x8& odd) {
    uint32_t e[4], o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        e[j] = __builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], 0x05040100u);
        o[j] = __builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], 0x07060302u);
    }
    even = __builtin_bit_cast(bf16x8, e);
    odd = __builtin_bit_cast(bf16x8, o);
}

// one k-step's operands of one lane
template <bool BKN, int NB> struct Frags;
template <int NB> struct Frags<false, NB> {
    uint4 a[2], b[NB];
    __device__ __forceinline__ void load(const GemmParams& p, const uint16_t* const (&ap)[2], const uint16_t* const (&bp)[NB], int k) {
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const uint4*>(ap[t] + k);
#pragma unroll
        for (int t = 0; t < NB; ++t) b[t] = *reinterpret_cast<const uint4*>(bp[t] + k);
    }
    __device__ __forceinline__ bf16x8 B(int t) const { return __builtin_bit_cast(bf16x8, b[t]); }
};
template <int NB> struct Frags<true, NB> {
    uint4 a[2];
    uint32_t b[NB / 2][8];
    // bp[bb] points at B[8kq][n0 + 32bb + 2j]
    __device__ __forceinline__ void load(const GemmParams& p, const uint16_t* const (&ap)[2], const uint16_t* const (&bp)[NB], int k) {
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const uint4*>(ap[t] + k);
#pragma unroll
        for (int bb = 0; bb < NB / 2; ++bb)
#pragma unroll
            for (int r = 0; r < 8; ++r) b[bb][r] = *reinterpret_cast<const uint32_t*>(bp[bb] + (int64_t)(k + r) * p.ldb);
    }
};

// LDS floats the body needs (the caller owns the buffer, so that kernels hosting several bodies can overlay them)
template <int NB, int NW> constexpr int gemm_lds_floats() { return NW * BM * (16 * NB + 4); }

template <bool BKN, int EPI, int NB, int NW>
__device__ __forceinline__ void layer_gemm_body(const GemmParams& p, const int bid, float* __restrict__ lds) {
    constexpr int BN = 16 * NB;
    constexpr int LDP = BN + 4;                  // LDS row stride (floats): the lanes of one store spread over all banks
    float (*part)[BM * LDP] = reinterpret_cast<float (*)[BM * LDP]>(lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

    // rows / columns past the edge are clamped for the loads (their products are never stored)
    const uint16_t* ap[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) ap[t] = p.A + (int64_t)min(m0 + 16 * t + i, p.M - 1) * p.lda + 8 * kq;
    const uint16_t* bp[NB];
    if (BKN) {
#pragma unroll
        for (int bb = 0; bb < NB; ++bb)
            bp[bb] = bb < NB / 2 ? p.B + (int64_t)(8 * kq) * p.ldb + min(n0 + 32 * bb + 2 * i, p.N - 2) : nullptr;
    } else {
#pragma unroll
        for (int t = 0; t < NB; ++t) bp[t] = p.B + (int64_t)min(n0 + 16 * t + i, p.N - 1) * p.ldb + 8 * kq;
    }

    f32x4 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    int k = wave * KSTEP;
    Frags<BKN, NB> cur, nxt;
    if (k < p.K) cur.load(p, ap, bp, k);
    for (; k < p.K; k += NW * KSTEP) {
        const int kn = k + NW * KSTEP;
        if (kn < p.K) nxt.load(p, ap, bp, kn);
        bf16x8 bf[NB];
        if constexpr (BKN) {
#pragma unroll
            for (int bb = 0; bb < NB / 2; ++bb) split_pairs(cur.b[bb], bf[2 * bb], bf[2 * bb + 1]);
        } else {
#pragma unroll
            for (int t = 0; t < NB; ++t) bf[t] = cur.B(t);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, cur.a[a]);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[b], acc[a][b], 0, 0, 0);
        }
        if (kn < p.K) cur = nxt;
    }

    // register v of lane (j = lane & 15, q = lane >> 4) is MFMA row 4q + v, column j.  B as [N,K]: operand b covers
    // columns 16b + j.  B as [K,N]: operands (2bb, 2bb+1) are the even / odd columns 32bb + 2j (+1).
    float* mine = part[wave];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int col = BKN ? 32 * (b >> 1) + 2 * i + (b & 1) : 16 * b + i;
#pragma unroll
            for (int v = 0; v < 4; ++v) mine[(16 * a + 4 * kq + v) * LDP + col] = acc[a][b][v];
        }
    __syncthreads();

    constexpr int TPR = BN / 8;                  // threads per output row, 8 adjacent columns each
    if (threadIdx.x >= BM * TPR) return;
    const int r = threadIdx.x / TPR, c = (threadIdx.x % TPR) * 8;
    const int row = m0 + r, col = n0 + c;
    if (row >= p.M || col >= p.N) return;
    float s[8];
    {
        const float4 x0 = *reinterpret_cast<const float4*>(&part[0][r * LDP + c]);
        const float4 x1 = *reinterpret_cast<const float4*>(&part[0][r * LDP + c + 4]);
        s[0] = x0.x; s[1] = x0.y; s[2] = x0.z; s[3] = x0.w; s[4] = x1.x; s[5] = x1.y; s[6] = x1.z; s[7] = x1.w;
    }
    const int nw_used = min(NW, (p.K + KSTEP - 1) / KSTEP);
    for (int w = 1; w < nw_used; ++w) {
        const float4 x0 = *reinterpret_cast<const float4*>(&part[w][r * LDP + c]);
        const float4 x1 = *reinterpret_cast<const float4*>(&part[w][r * LDP + c + 4]);
        s[0] += x0.x; s[1] += x0.y; s[2] += x0.z; s[3] += x0.w; s[4] += x1.x; s[5] += x1.y; s[6] += x1.z; s[7] += x1.w;
    }
    if (p.bias) {
        float bv[8];
        load8(reinterpret_cast<const bf16_t*>(p.bias) + col, bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += bv[e];
    }
    const int64_t o = (int64_t)row * p.ldc + col;
    if constexpr (EPI == EPI_BIAS) {
        store8(reinterpret_cast<bf16_t*>(p.C) + o, s);
    } else if constexpr (EPI == EPI_GELU) {
        // h from the ROUNDED pre-activation, as a separate GELU launch reading the bf16 u would compute it
        const bf16x8 u8 = pack8(s);
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.C) + o) = u8;
        float h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = gelu_f((float)u8[e]);
        store8(reinterpret_cast<bf16_t*>(p.aux_out) + o, h);
    } else if constexpr (EPI == EPI_GELU_BWD) {
        float u[8];
        load8(reinterpret_cast<const bf16_t*>(p.aux_in) + o, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] *= gelu_grad(u[e]);
        store8(reinterpret_cast<bf16_t*>(p.C) + o, s);
    } else {
        float t[8];
        load8(reinterpret_cast<const float*>(p.aux_in) + o, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += t[e];
        store8(reinterpret_cast<float*>(p.C) + o, s);
    }
}


}  // namespace mobgt_gemm
