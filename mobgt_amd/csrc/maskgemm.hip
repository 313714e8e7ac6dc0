// GraphConvolution's adjacency product (graphormer/modelGNN.py:38-44 `torch.spmm(adj, support)`) for the DENSE-ish POI
// graph of the Foursquare / Gowalla universes, from a BITMASK of the adjacency.
//
// The reference's adjacency is (D+I)^-1 (A+I) with a 0/1 "within 3 km" matrix A (model_fqandtoyo.py:481-486): every
// non-zero of row i equals 1/(deg_i + 1).  Streaming it as a dense bf16 matrix costs 123 MB per product at P = 7 856
// (39-44 us at ~3 TB/s, twice per step = 7.6 % of the S-FSQ step); as CSR it costs one gathered row of the operand per
// non-zero (443 MB of L2 traffic at 2.8 % density, measured slower in round 1).  The structure says: ONE BIT per entry
// (7.7 MB, L2 / MALL resident) and one f32 per row:
//     out[i, :] = rscale[i] * sum_k bit(i, k) * (bscale[k] *) X[k, :]  (+ bias)
// -- `rscale` = 1/(deg+1) for the forward product A X;  for the transposed product A^T G the mask of the transpose and
// `bscale` = 1/(deg+1) on the operand's rows.  The product runs on v_mfma_f32_16x16x32_bf16: a lane's A operand (8
// consecutive k of one row) is ONE BYTE of the mask, expanded to 8 bf16 ones / zeros by a 256-entry LDS table (one
// ds_read_b128, no VALU); the B operand is X rounded to bf16 on the fly (what the dense path's bf16 `support` was), f32
// accumulate, split-K over the waves of a workgroup, partial tiles meet in LDS.  With the GCN evaluated as (A X) W
// instead of A (X W) the operand is 16 wide (modelGNN.GCN), i.e. 120 k MFMAs per product in all.
#include <algorithm>
#include "common.h"
#include "mobgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct MaskGemmParams {
    const uint32_t* mask; int64_t ldm;      // [M, ldm] words (ldm % 4 == 0, ldm * 32 >= roundup(K, 128)), bit (k & 31) of word k >> 5 = entry (row, k); bits >= K zero
    const float* X; int64_t ldx;            // [K, N] f32
    const uint16_t* Xt; int64_t ldxt;       // [N, ldxt] bf16: X transposed (and scaled by bscale), zero beyond K -- or null
    const float* bscale;                    // [K] or null
    const float* rscale;                    // [M] or null
    const float* bias;                      // [N] or null
    float* out; int64_t ldo;                // [M, N] f32
    int M, K, N;
};

// xt[n][k] = bf16(x[k][n] * bscale[k]), zero for K <= k < ldxt: 64 k x N per workgroup through an LDS tile, so that both the
// reads (rows of x) and the writes (128-byte runs of xt) are contiguous
__global__ __launch_bounds__(256) void xt_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ bscale,
                                                 uint16_t* __restrict__ xt, int64_t ldxt, int K, int N) {
    __shared__ float tile[64][65];
    const int k0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * N; e += 256) {
        const int kk = e / N, n = e % N, k = k0 + kk;
        float v = 0.f;
        if (k < K) v = x[(int64_t)k * ldx + n] * (bscale ? bscale[k] : 1.f);
        tile[kk][n] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * 64; e += 256) {
        const int n = e / 64, kk = e % 64;
        if (k0 + kk < ldxt) xt[(int64_t)n * ldxt + k0 + kk] = __builtin_bit_cast(uint16_t, (bf16_t)tile[kk][n]);
    }
}

// NB 16-column operands, NWAVE waves splitting K, RT 16-row tiles per workgroup
template <int NB, int NWAVE, int RT>
__global__ __launch_bounds__(NWAVE * 64) void mask_gemm_kernel(const MaskGemmParams p) {
    constexpr int BM = 16 * RT, BN = 16 * NB, LDP = BN + 4;
    __shared__ __attribute__((aligned(16))) uint4 lut[256];
    __shared__ __attribute__((aligned(16))) float part[NWAVE][BM * LDP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    for (int b = threadIdx.x; b < 256; b += NWAVE * 64) {
        uint32_t d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = ((b >> (2 * j)) & 1 ? 0x3F80u : 0u) | ((b >> (2 * j + 1)) & 1 ? 0x3F800000u : 0u);
        lut[b] = make_uint4(d[0], d[1], d[2], d[3]);
    }
    __syncthreads();

    const uint32_t* mrow[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a) mrow[a] = p.mask + (int64_t)min(m0 + 16 * a + i, p.M - 1) * p.ldm;
    f32x4 acc[RT][NB];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = (p.K + 31) >> 5;
    // Loads come in GROUPS of four k-steps: one 16-byte load of four mask words per row tile (a workgroup's 64 mask rows
    // are 63 KB, twice the CU's L1: word-by-word, every step paid an L2 round trip) and four 16-byte B operands -- 8
    // consecutive k of one column from the transposed bf16 copy (xt_kernel).  Each wave owns a contiguous, 4-aligned block
    // of k-steps and keeps the next group in flight.
    struct Group {
        uint4 w[RT];
        uint4 b[4][NB];
    };
    const int per = ((nsteps + NWAVE - 1) / NWAVE + 3) & ~3;
    const int s0 = wave * per, s1 = min(nsteps, s0 + per);
    auto load = [&](Group& g, const int ks) {                     // ks % 4 == 0; rows are padded to whole groups
#pragma unroll
        for (int a = 0; a < RT; ++a) g.w[a] = *reinterpret_cast<const uint4*>(mrow[a] + ks);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                g.b[u][b] = *reinterpret_cast<const uint4*>(p.Xt + (int64_t)(16 * b + i) * p.ldxt + 32 * (ks + u) + 8 * kq);
    };
    auto compute = [&](const Group& g, const int ks) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ks + u >= s1) break;
#pragma unroll
            for (int a = 0; a < RT; ++a) {
                const uint32_t w = u == 0 ? g.w[a].x : (u == 1 ? g.w[a].y : (u == 2 ? g.w[a].z : g.w[a].w));
                const bf16x8 af = __builtin_bit_cast(bf16x8, lut[(w >> (8 * kq)) & 0xffu]);
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, g.b[u][b]), acc[a][b], 0, 0, 0);
            }
        }
    };
    Group cur, nxt;
    if (s0 < s1) load(cur, s0);
    for (int ks = s0; ks < s1; ks += 4) {
        if (ks + 4 < s1) load(nxt, ks + 4);
        compute(cur, ks);
        if (ks + 4 < s1) cur = nxt;
    }

    // register v of lane (j = lane & 15, q = lane >> 4) is MFMA row 4q + v, column j
    float* mine = part[wave];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) mine[(16 * a + 4 * kq + v) * LDP + 16 * b + i] = acc[a][b][v];
    __syncthreads();
    for (int e = threadIdx.x; e < BM * (BN / 4); e += NWAVE * 64) {
        const int r = e / (BN / 4), c = (e % (BN / 4)) * 4;
        const int row = m0 + r;
        if (row >= p.M || c >= p.N) continue;
        float4 s = *reinterpret_cast<const float4*>(&part[0][r * LDP + c]);
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) {
            const float4 t = *reinterpret_cast<const float4*>(&part[w][r * LDP + c]);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const float rs = p.rscale ? p.rscale[row] : 1.f;
        float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s.x = fmaf(s.x, rs, b4.x); s.y = fmaf(s.y, rs, b4.y); s.z = fmaf(s.z, rs, b4.z); s.w = fmaf(s.w, rs, b4.w);
        *reinterpret_cast<float4*>(p.out + (int64_t)row * p.ldo + c) = s;
    }
}


// ---- Round 4: the distance GCN's hidden and last layer around two bitmask products (modelGNN.GCN, rows-only form) ----------------
// graphormer/modelGNN.py:38-44 (GraphConvolution: adj @ (x @ W) + b), :66-72 (GCN: LeakyReLU after every hidden layer, dropout in
// front of the last one); model_fqandtoyo.py:1236 / :1264 (the [P, hidden] table is read at the batch's POI rows only).
// With h0 = 16 and h1 = 64 the step ran  mask product -> small GEMM (+act) -> small GEMM (y1 W2, P rows) -> row gather of the
// dense bf16 adjacency (+ its transpose, 19 MB) -> skinny product  forward and four launches backward.  Here:
//   mask_gemm_l1_kernel   t1 = rs * (A y0) [P,16]  and, in its epilogue, y1 = dropout(leaky(t1 W1 + b1)) [P,64], also transposed
//                         in bf16 (the next product's operand);
//   mask_rows_kernel      u = rs[rows] * (A[rows] y1) [R,64] straight from the BITMASK rows (1 KB each instead of 16 KB of
//                         bf16) and out = u W2 + b2 [R,NO] as four partial tables the consumer adds;
//   mask_rows_bwd_kernel  dy1 = A[rows]^T (rs[rows] * (g W2^T)) [P,64]: the restricted transposed product, its A operand
//                         bit (k, r) = bit rows[r] of mask_t[k] gathered into an LDS bit tile first; epilogue: the hidden
//                         layer's data gradient ((dy1 * m(y1)) W1^T * bscale)^T in bf16, the operand of A^T (.).
constexpr int GH = 64;                  // width of the last hidden layer
constexpr int GI = 16;                  // width of the first hidden layer
#ifndef MASK_ROWS_ALL_IN_FLIGHT
#define MASK_ROWS_ALL_IN_FLIGHT 0
#endif
constexpr int MASK_ROWS_MAX_NO = 192;   // widest last layer mask_rows_kernel stages in LDS

struct MaskL1Params {
    const uint32_t* mask; int64_t ldm;
    const uint16_t* Xt; int64_t ldxt;       // y0^T bf16 [16][ldxt], zero beyond K
    const float* rscale;                    // [M]
    float* t; int64_t ldt;                  // [M,16] f32
    const float* W1; const float* b1;       // [16,64] row-major, [64]
    float* y; int64_t ldy;                  // [M,64] f32
    uint16_t* yt; int64_t ldyt;             // [64][ldyt] bf16 (unscaled)
    float slope; uint32_t thr; float inv_keep; uint64_t seed; const uint64_t* seed_dev; uint32_t salt;
    float4* zero; int64_t n_zero4;          // side job: n_zero4 float4 of zeros (the next launch's atomic destinations)
    int M, K;
};

__device__ __forceinline__ void lut_fill(uint4* lut, int nthreads) {
    for (int b = threadIdx.x; b < 256; b += nthreads) {
        uint32_t d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = ((b >> (2 * j)) & 1 ? 0x3F80u : 0u) | ((b >> (2 * j + 1)) & 1 ? 0x3F800000u : 0u);
        lut[b] = make_uint4(d[0], d[1], d[2], d[3]);
    }
}

template <int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void mask_gemm_l1_kernel(const MaskL1Params p) {
    constexpr int RT = 2, BM = 32, LDP = GI + 4, NT = NWAVE * 64;
    __shared__ __attribute__((aligned(16))) uint4 lut[256];
    __shared__ __attribute__((aligned(16))) float part[NWAVE][BM * LDP];
    __shared__ float tt[BM][GI + 1];
    __shared__ float ys[BM][GH + 1];
    __shared__ float w1s[GI][GH];
    __shared__ float b1s[GH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    lut_fill(lut, NT);
    for (int e = threadIdx.x; e < GI * GH; e += NT) w1s[e / GH][e % GH] = p.W1[e];
    if (threadIdx.x < GH) b1s[threadIdx.x] = p.b1[threadIdx.x];
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < p.n_zero4; e += (int64_t)gridDim.x * NT)
        p.zero[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const uint32_t* mrow[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a) mrow[a] = p.mask + (int64_t)min(m0 + 16 * a + i, p.M - 1) * p.ldm;
    f32x4 acc[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = (p.K + 31) >> 5;
    struct Group { uint4 w[RT]; uint4 b[4]; };
    const int per = ((nsteps + NWAVE - 1) / NWAVE + 3) & ~3;
    const int s0 = wave * per, s1 = min(nsteps, s0 + per);
    auto load = [&](Group& g, const int ks) {
#pragma unroll
        for (int a = 0; a < RT; ++a) g.w[a] = *reinterpret_cast<const uint4*>(mrow[a] + ks);
#pragma unroll
        for (int u = 0; u < 4; ++u) g.b[u] = *reinterpret_cast<const uint4*>(p.Xt + (int64_t)i * p.ldxt + 32 * (ks + u) + 8 * kq);
    };
    auto compute = [&](const Group& g, const int ks) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ks + u >= s1) break;
#pragma unroll
            for (int a = 0; a < RT; ++a) {
                const uint32_t w = u == 0 ? g.w[a].x : (u == 1 ? g.w[a].y : (u == 2 ? g.w[a].z : g.w[a].w));
                const bf16x8 af = __builtin_bit_cast(bf16x8, lut[(w >> (8 * kq)) & 0xffu]);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, g.b[u]), acc[a], 0, 0, 0);
            }
        }
    };
    Group cur, nxt;
    if (s0 < s1) load(cur, s0);
    for (int ks = s0; ks < s1; ks += 4) {
        if (ks + 4 < s1) load(nxt, ks + 4);
        compute(cur, ks);
        if (ks + 4 < s1) cur = nxt;
    }
    float* mine = part[wave];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int v = 0; v < 4; ++v) mine[(16 * a + 4 * kq + v) * LDP + i] = acc[a][v];
    __syncthreads();
    if (threadIdx.x < BM * GI) {                                    // t1 = rs * sum over the waves' K-slices
        const int r = threadIdx.x / GI, c = threadIdx.x % GI, row = m0 + r;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) s += part[w][r * LDP + c];
        s *= p.rscale[min(row, p.M - 1)];
        tt[r][c] = s;
        if (row < p.M) p.t[(int64_t)row * p.ldt + c] = s;
    }
    __syncthreads();
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    for (int e = threadIdx.x; e < BM * (GH / 2); e += NT) {          // y1 = dropout(leaky(t1 W1 + b1)): two columns per thread
        const int r = e / (GH / 2), c = (e % (GH / 2)) * 2, row = m0 + r;
        float a0 = b1s[c], a1 = b1s[c + 1];
#pragma unroll
        for (int k = 0; k < GI; ++k) {
            const float tv = tt[r][k];
            a0 = fmaf(tv, w1s[k][c], a0);
            a1 = fmaf(tv, w1s[k][c + 1], a1);
        }
        a0 = a0 > 0.f ? a0 : p.slope * a0;
        a1 = a1 > 0.f ? a1 : p.slope * a1;
        if (p.thr) {                                                 // (sgemm.hip's epilogue rule: the host replay is unchanged)
            const uint32_t rowh = dropout_row_hash(seed, (uint32_t)row ^ p.salt);
            a0 = dropout_bits16(seed, rowh, (uint32_t)c) >= p.thr ? a0 * p.inv_keep : 0.f;
            a1 = dropout_bits16(seed, rowh, (uint32_t)(c + 1)) >= p.thr ? a1 * p.inv_keep : 0.f;
        }
        ys[r][c] = a0; ys[r][c + 1] = a1;
        if (row < p.M) *reinterpret_cast<float2*>(p.y + (int64_t)row * p.ldy + c) = make_float2(a0, a1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < GH * (BM / 2); e += NT) {          // ... and transposed, bf16: 64-byte runs per column
        const int c = e / (BM / 2), r = (e % (BM / 2)) * 2, row = m0 + r;
        const uint16_t lo = __builtin_bit_cast(uint16_t, (bf16_t)ys[r][c]), hi = __builtin_bit_cast(uint16_t, (bf16_t)ys[r + 1][c]);
        uint16_t* q = p.yt + (int64_t)c * p.ldyt + row;
        if (row + 1 < p.M) *reinterpret_cast<uint32_t*>(q) = (uint32_t)lo | ((uint32_t)hi << 16);
        else if (row < p.M) *q = lo;
    }
}

struct MaskRowsParams {
    const uint32_t* mask; int64_t ldm;
    const int64_t* rows;                    // [R] row of the adjacency for output row r
    const uint16_t* Xt; int64_t ldxt;       // y1^T bf16 [64][ldxt], zero beyond K
    const float* rscale;                    // [P]
    const float* W2; const float* b2;       // [64,NO] row-major, [NO] or null
    float* u;                               // [R,64] f32 (written)
    float* out; int64_t ldo;                // parts [4][R,NO] f32 (written): out = parts[0] + parts[1] + parts[2] + parts[3]
    float* rs_rows;                         // [R]: rscale[rows[r]] (for the backward pass), or null
    int R, K, NO, KS;
};

// FOUR workgroups per 16-row tile, one per 16 of y1's 64 columns; each walks all of K with 16 waves (a wave's share is four
// 128-deep groups at P = 7 856, the next group requested while the current one is multiplied: with all four in flight at once
// -- MASK_ROWS_ALL_IN_FLIGHT=1 -- the launch measured 13.1 us against 12.0; the same change made mask_gemm_l1_kernel,
// mask_gemm_kernel and mask_rows_bwd_kernel 1.6-2.1 us SLOWER each).  u is written once, in a fixed summation order.
// out = u W2 + b2 is linear in u's column blocks: workgroup q writes ITS product u[:, 16q : 16q + 16] W2[16q : 16q + 16, :]
// (+ b2 for q = 0) to parts[q] with plain stores, and the consumer adds the four tables in a fixed order
// (mobgt_embed_gather_multi, which gathers these rows anyway): no atomics, no zero-fill, bit-reproducible.
// (Measured before this form, S-FSQ, R = 608: K split over six workgroups per tile with f32 atomics into u and out 13.4 us --
// and eval-mode logits that differed from run to run; two workgroups per tile with a two-term atomic sum 20.3 us.)
__global__ __launch_bounds__(1024) void mask_rows_kernel(const MaskRowsParams p) {
    constexpr int NWAVE = 16, BN = 16, LDP = BN + 4, NT = NWAVE * 64;
    __shared__ __attribute__((aligned(16))) uint4 lut[256];
    __shared__ __attribute__((aligned(16))) float part[NWAVE][16 * LDP];
    __shared__ float us[16][BN + 1];
    __shared__ float rs_s[16];
    __shared__ __attribute__((aligned(16))) float w2s[BN * MASK_ROWS_MAX_NO];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int q = blockIdx.x & 3, m0 = (blockIdx.x >> 2) * 16;
    const int64_t arow = p.rows[min(m0 + i, p.R - 1)];
    const uint32_t* mrow = p.mask + arow * p.ldm;
    // what the epilogue reads (this block's 16 rows of W2, the row scales) is requested now
    const int n_w2v = BN * p.NO / 4;                                   // <= 768 float4: at most one per thread
    const float4 w2r = (int)threadIdx.x < n_w2v ? reinterpret_cast<const float4*>(p.W2 + (int64_t)BN * q * p.NO)[threadIdx.x]
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
    const float my_rs = threadIdx.x < 16 ? p.rscale[p.rows[min(m0 + (int)threadIdx.x, p.R - 1)]] : 0.f;
    lut_fill(lut, NT);
    __syncthreads();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = (p.K + 31) >> 5;
    const int per = ((nsteps + NWAVE - 1) / NWAVE + 3) & ~3;
    const int s0 = wave * per, s1 = min(nsteps, s0 + per);
    const uint16_t* xrow = p.Xt + (int64_t)(BN * q + i) * p.ldxt + 8 * kq;
    struct Group { uint4 w; uint4 b[4]; };
#if MASK_ROWS_ALL_IN_FLIGHT
    for (int base = s0; base < s1; base += 16) {
        Group g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ks = min(base + 4 * j, ((nsteps + 3) & ~3) - 4);  // (clamped: a group past the end is loaded, never multiplied)
            g[j].w = *reinterpret_cast<const uint4*>(mrow + ks);
#pragma unroll
            for (int u = 0; u < 4; ++u) g[j].b[u] = *reinterpret_cast<const uint4*>(xrow + 32 * (ks + u));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (base + 4 * j + u >= s1) break;
                const uint32_t w = u == 0 ? g[j].w.x : (u == 1 ? g[j].w.y : (u == 2 ? g[j].w.z : g[j].w.w));
                const bf16x8 af = __builtin_bit_cast(bf16x8, lut[(w >> (8 * kq)) & 0xffu]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, g[j].b[u]), acc, 0, 0, 0);
            }
    }
#else
    auto load = [&](Group& g, const int ks) {
        g.w = *reinterpret_cast<const uint4*>(mrow + ks);
#pragma unroll
        for (int u = 0; u < 4; ++u) g.b[u] = *reinterpret_cast<const uint4*>(xrow + 32 * (ks + u));
    };
    Group cur, nxt;
    if (s0 < s1) load(cur, s0);
    for (int ks = s0; ks < s1; ks += 4) {
        if (ks + 4 < s1) load(nxt, ks + 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ks + u >= s1) break;
            const uint32_t w = u == 0 ? cur.w.x : (u == 1 ? cur.w.y : (u == 2 ? cur.w.z : cur.w.w));
            const bf16x8 af = __builtin_bit_cast(bf16x8, lut[(w >> (8 * kq)) & 0xffu]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, cur.b[u]), acc, 0, 0, 0);
        }
        if (ks + 4 < s1) cur = nxt;
    }
#endif
    float* mine = part[wave];
#pragma unroll
    for (int v = 0; v < 4; ++v) mine[(4 * kq + v) * LDP + i] = acc[v];
    if ((int)threadIdx.x < n_w2v) reinterpret_cast<float4*>(w2s)[threadIdx.x] = w2r;
    if (threadIdx.x < 16) rs_s[threadIdx.x] = my_rs;
    __syncthreads();
    if (threadIdx.x < 16 * BN) {
        const int r = threadIdx.x / BN, c = threadIdx.x % BN, row = m0 + r;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) s += part[w][r * LDP + c];
        const float rs = rs_s[r];
        s *= rs;
        us[r][c] = s;
        if (row < p.R) {
            p.u[(int64_t)row * GH + BN * q + c] = s;
            if (p.rs_rows && q == 0 && c == 0) p.rs_rows[row] = rs;
        }
    }
    __syncthreads();
    float* dst = p.out + (int64_t)q * p.R * p.ldo;
    for (int e = threadIdx.x; e < 16 * p.NO; e += NT) {               // this column block's share of out = u W2 (+ b2)
        const int r = e / p.NO, c = e % p.NO, row = m0 + r;
        float o = (q == 0 && p.b2) ? p.b2[c] : 0.f;
#pragma unroll
        for (int k = 0; k < BN; ++k) o = fmaf(us[r][k], w2s[k * p.NO + c], o);
        if (row < p.R) dst[(int64_t)row * p.ldo + c] = o;
    }
}

struct MaskRowsBwdParams {
    const uint32_t* mask_t; int64_t ldm;    // bit i of row k = A[i][k]
    const int64_t* rows; int R, P;
    const uint16_t* GuT; int64_t ldg;       // [64][ldg] bf16: (rs[rows[r]] * (g W2^T)[r, :])^T, zero beyond R; ldg % 128 == 0
    const float* y1; int64_t ldy;           // [P,64]: the hidden layer's output (its sign / zero pattern is the derivative)
    float mpos, mneg, mzero;
    const float* W1;                        // [16,64]
    const float* bscale;                    // [P]
    float* dy1; int64_t lddy;               // [P,64] f32, the gradient at y1 (unmasked)
    uint16_t* dtt; int64_t lddt;            // [16][lddt] bf16: (((dy1 * m(y1)) W1^T) * bscale)^T
};

__global__ __launch_bounds__(512) void mask_rows_bwd_kernel(const MaskRowsBwdParams p) {
    constexpr int NWAVE = 8, BM = 32, NB = GH / 16, LDP = GH + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ __attribute__((aligned(16))) uint4 lut[256];
    __shared__ float w1s[GI][GH + 1];
    __shared__ float gs[BM][GH + 1];
    __shared__ float dts[GI][BM + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int k0 = blockIdx.x * BM;
    const int nw = (int)(p.ldg / 32);                        // words of r-bits per k-row
    const int ldmt = (int)p.ldm + 1;                         // LDS row stride of the mask_t rows (odd: the lookups spread over banks)
    // LDS: [mask_t rows | partial tiles] share one region (the rows are dead once the bit tile exists), then rows[], then bits
    const size_t region = max((size_t)BM * ldmt * 4, (size_t)NWAVE * BM * LDP * 4);
    uint32_t* mt = reinterpret_cast<uint32_t*>(dyn);
    float* part = reinterpret_cast<float*>(dyn);
    int* rows_s = reinterpret_cast<int*>(dyn + region);
    uint32_t* abits = reinterpret_cast<uint32_t*>(dyn + region + (size_t)p.ldg * 4);     // [BM][nw + 1]
    // what the epilogue reads (this thread's four y1 values, its row's scale) and the first product step's operand are
    // requested before the bit tile is built
    const int er = threadIdx.x / (GH / 4), ec = (threadIdx.x % (GH / 4)) * 4;
    const float4 yv_pre = *reinterpret_cast<const float4*>(p.y1 + (int64_t)min(k0 + er, p.P - 1) * p.ldy + ec);
    const float bs_pre = p.bscale[min(k0 + (int)threadIdx.x / GI, p.P - 1)];
    uint4 bq0[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
        bq0[b] = *reinterpret_cast<const uint4*>(p.GuT + (int64_t)(16 * b + i) * p.ldg + 32 * min(wave, nw - 1) + 8 * kq);
    lut_fill(lut, 512);
    for (int e = threadIdx.x; e < GI * GH; e += 512) w1s[e / GH][e % GH] = p.W1[e];
    for (int e = threadIdx.x; e < BM * (int)p.ldm; e += 512) {
        const int kr = e / (int)p.ldm, w = e % (int)p.ldm;
        mt[kr * ldmt + w] = p.mask_t[(int64_t)min(k0 + kr, p.P - 1) * p.ldm + w];
    }
    for (int e = threadIdx.x; e < (int)p.ldg; e += 512) rows_s[e] = e < p.R ? (int)p.rows[e] : -1;
    __syncthreads();
    for (int e = threadIdx.x; e < nw * BM; e += 512) {
        const int j = e / BM, kr = e % BM;
        uint32_t w = 0u;
#pragma unroll 8
        for (int b = 0; b < 32; ++b) {
            const int rr = rows_s[32 * j + b];
            if (rr >= 0) w |= ((mt[kr * ldmt + (rr >> 5)] >> (rr & 31)) & 1u) << b;
        }
        abits[kr * (nw + 1) + j] = w;
    }
    __syncthreads();                                           // (mt is dead from here on: `part` reuses it after the products)
    f32x4 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = wave; ks < nw; ks += NWAVE) {
        uint4 bq[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
            bq[b] = ks == wave ? bq0[b] : *reinterpret_cast<const uint4*>(p.GuT + (int64_t)(16 * b + i) * p.ldg + 32 * ks + 8 * kq);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const uint32_t w = abits[(16 * a + i) * (nw + 1) + ks];
            const bf16x8 af = __builtin_bit_cast(bf16x8, lut[(w >> (8 * kq)) & 0xffu]);
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, bq[b]), acc[a][b], 0, 0, 0);
        }
    }
    float* mine = part + (size_t)wave * BM * LDP;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) mine[(16 * a + 4 * kq + v) * LDP + 16 * b + i] = acc[a][b][v];
    __syncthreads();
    for (int e = threadIdx.x; e < BM * (GH / 4); e += 512) {
        const int r = e / (GH / 4), c = (e % (GH / 4)) * 4, row = k0 + r;
        float4 s = *reinterpret_cast<const float4*>(part + r * LDP + c);
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) {
            const float4 t = *reinterpret_cast<const float4*>(part + (size_t)w * BM * LDP + r * LDP + c);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const float4 yv = yv_pre;                              // (e == threadIdx.x: BM * GH / 4 == 512 threads, one round)
        if (row < p.P) *reinterpret_cast<float4*>(p.dy1 + (int64_t)row * p.lddy + c) = s;
        auto m = [&](float y) { return y > 0.f ? p.mpos : (y < 0.f ? p.mneg : p.mzero); };
        gs[r][c] = s.x * m(yv.x); gs[r][c + 1] = s.y * m(yv.y); gs[r][c + 2] = s.z * m(yv.z); gs[r][c + 3] = s.w * m(yv.w);
    }
    __syncthreads();
    {                                                          // dt = (g1 W1^T) * bscale: 32 x 16 outputs, one per thread
        const int r = threadIdx.x / GI, j = threadIdx.x % GI;
        float s = 0.f;
#pragma unroll 16
        for (int c = 0; c < GH; ++c) s = fmaf(gs[r][c], w1s[j][c], s);
        dts[j][r] = s * bs_pre;
    }
    __syncthreads();
    {
        const int j = threadIdx.x / BM, r = threadIdx.x % BM, row = k0 + r;
        if (row < p.P) p.dtt[(int64_t)j * p.lddt + row] = __builtin_bit_cast(uint16_t, (bf16_t)dts[j][r]);
    }
}

}  // namespace

extern "C" int64_t mobgt_mask_gemm_workspace_bytes(int K, int N) {
    return (int64_t)N * (((int64_t)K + 127) / 128 * 128) * 2;
}

extern "C" int mobgt_mask_gemm(const uint32_t* mask, int64_t ld_mask_words, const float* x, int64_t ldx, const float* bscale,
                               const float* rscale, const float* bias, float* out, int64_t ld_out, void* work, int M, int K,
                               int N, void* stream) {
    if (M <= 0 || K <= 0) return 0;
    if (N <= 0 || (N & 15) || N > 64 || (ld_out & 3) || ld_mask_words * 32 < K || !work) return MOBGT_EBADDIM;
    if (((uintptr_t)out | (uintptr_t)bias | (uintptr_t)work) & 15) return MOBGT_EALIGN;
    const int64_t ldxt = ((int64_t)K + 127) / 128 * 128;          // whole groups of four k-steps
    if (ld_mask_words % 4 || ld_mask_words * 32 < ldxt) return MOBGT_EBADDIM;
    MaskGemmParams p = {mask, ld_mask_words, x, ldx, reinterpret_cast<const uint16_t*>(work), ldxt, bscale, rscale, bias, out,
                        ld_out, M, K, N};
    hipStream_t st = (hipStream_t)stream;
    // x null: `work` already holds the operand transposed (bf16 [N][ldxt], scaled, zero beyond K) -- written by its producer
    // (mobgt_bias_act_fwd_t / mobgt_small_gemm_f32_act), no transpose launch
    if (x) hipLaunchKernelGGL(xt_kernel, dim3((unsigned)(ldxt / 64)), dim3(256), 0, st, x, ldx, bscale, reinterpret_cast<uint16_t*>(work),
                              ldxt, K, N);
    // every workgroup walks ALL of X: row blocks of 32 (two row tiles share each operand load) and 16 waves on K keep both the
    // L2 -> CU traffic (P/32 x |X|) and the per-wave chain of k-steps short
    // (N = 16, measured at M = K = 7856: 64 rows x 16 waves 14.5 us, 32 rows x 16 waves 11.9, 32 x 8 waves 12.1, 16 x 16 19.4)
    if (N == 16) hipLaunchKernelGGL((mask_gemm_kernel<1, 16, 2>), dim3((M + 31) / 32), dim3(1024), 0, st, p);
    else if (N == 32) hipLaunchKernelGGL((mask_gemm_kernel<2, 16, 2>), dim3((M + 31) / 32), dim3(1024), 0, st, p);
    else hipLaunchKernelGGL((mask_gemm_kernel<4, 8, 2>), dim3((M + 31) / 32), dim3(512), 0, st, p);
    return (int)hipGetLastError();
}

/* Round 4 -- the distance GCN's hidden + last layer around two bitmask products (see the kernels' header comment). */
extern "C" int mobgt_mask_gemm_l1_fwd(const uint32_t* mask, int64_t ld_mask_words, const float* rscale, const void* y0t_bf16,
                                      int64_t ld_y0t, float* t, const float* w1, const float* b1, float slope, float dropout_p,
                                      uint64_t seed, const uint64_t* seed_dev, uint32_t salt, float* y, void* yt_bf16, int64_t ld_yt,
                                      void* zero, int64_t zero_floats, int M, int K, void* stream) {
    if (M <= 0 || K <= 0) return 0;
    const int64_t ldxt = ((int64_t)K + 127) / 128 * 128;
    if (ld_mask_words % 4 || ld_mask_words * 32 < ldxt || ld_y0t < ldxt || ld_yt < M || (ld_yt & 1) || (zero_floats & 3)) return MOBGT_EBADDIM;
    if (!mask || !rscale || !y0t_bf16 || !t || !w1 || !b1 || !y || !yt_bf16) return MOBGT_EBADDIM;
    if (((uintptr_t)y0t_bf16 | (uintptr_t)zero | (uintptr_t)mask) & 15 || ((uintptr_t)y & 7) || ((uintptr_t)yt_bf16 & 3)) return MOBGT_EALIGN;
    MaskL1Params p = {};
    p.mask = mask; p.ldm = ld_mask_words; p.Xt = reinterpret_cast<const uint16_t*>(y0t_bf16); p.ldxt = ld_y0t; p.rscale = rscale;
    p.t = t; p.ldt = GI; p.W1 = w1; p.b1 = b1; p.y = y; p.ldy = GH; p.yt = reinterpret_cast<uint16_t*>(yt_bf16); p.ldyt = ld_yt;
    p.slope = slope;
    if (dropout_p > 0.f) {
        p.thr = dropout_threshold(dropout_p);
        p.inv_keep = 1.f / (1.f - (float)p.thr / 65536.f);
        p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
    }
    p.zero = reinterpret_cast<float4*>(zero); p.n_zero4 = zero ? zero_floats / 4 : 0;
    p.M = M; p.K = K;
    hipLaunchKernelGGL((mask_gemm_l1_kernel<16>), dim3((M + 31) / 32), dim3(1024), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int mobgt_mask_rows_fwd(const uint32_t* mask, int64_t ld_mask_words, const int64_t* rows, const float* rscale,
                                   const void* y1t_bf16, int64_t ld_y1t, const float* w2, const float* b2, float* u, float* out,
                                   float* rs_rows, int R, int K, int NO, void* stream) {
    if (R <= 0 || K <= 0) return 0;
    const int64_t ldxt = ((int64_t)K + 127) / 128 * 128;
    if (ld_mask_words % 4 || ld_mask_words * 32 < ldxt || ld_y1t != ldxt || NO <= 0 || NO > MASK_ROWS_MAX_NO || (NO & 3)) return MOBGT_EBADDIM;
    if (!mask || !rows || !rscale || !y1t_bf16 || !w2 || !u || !out) return MOBGT_EBADDIM;
    if (((uintptr_t)y1t_bf16 | (uintptr_t)mask | (uintptr_t)w2) & 15) return MOBGT_EALIGN;
    MaskRowsParams p = {};
    p.mask = mask; p.ldm = ld_mask_words; p.rows = rows; p.Xt = reinterpret_cast<const uint16_t*>(y1t_bf16); p.ldxt = ld_y1t;
    p.rscale = rscale; p.W2 = w2; p.b2 = b2; p.u = u; p.out = out; p.ldo = NO; p.rs_rows = rs_rows; p.R = R; p.K = K; p.NO = NO;
    p.KS = 1;
    hipLaunchKernelGGL(mask_rows_kernel, dim3(4 * ((R + 15) / 16)), dim3(1024), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int64_t mobgt_mask_rows_bwd_lds_bytes(int64_t ld_mask_words, int R) {
    const int64_t ldg = ((int64_t)R + 127) / 128 * 128;
    const int64_t region = std::max<int64_t>(32 * (ld_mask_words + 1) * 4, (int64_t)8 * 32 * (GH + 4) * 4);
    return region + ldg * 4 + 32 * (ldg / 32 + 1) * 4;
}

extern "C" int mobgt_mask_rows_bwd(const uint32_t* mask_t, int64_t ld_mask_words, const int64_t* rows, const void* gut_bf16,
                                   int64_t ld_gut, const float* y1, float m_pos, float m_neg, float m_zero, const float* w1,
                                   const float* bscale, float* dy1, void* dtt_bf16, int64_t ld_dtt, int R, int P, void* stream) {
    if (R <= 0 || P <= 0) return 0;
    const int64_t ldg = ((int64_t)R + 127) / 128 * 128;
    if (ld_gut != ldg || ld_mask_words * 32 < P || ld_dtt < P) return MOBGT_EBADDIM;
    if (!mask_t || !rows || !gut_bf16 || !y1 || !w1 || !bscale || !dy1 || !dtt_bf16) return MOBGT_EBADDIM;
    if (((uintptr_t)gut_bf16 | (uintptr_t)y1 | (uintptr_t)dy1) & 15) return MOBGT_EALIGN;
    const int64_t lds = mobgt_mask_rows_bwd_lds_bytes(ld_mask_words, R);
    if (lds > 120 * 1024) return MOBGT_EBADDIM;                  // (static LDS of the kernel: ~19 KB beside it)
    MaskRowsBwdParams p = {};
    p.mask_t = mask_t; p.ldm = ld_mask_words; p.rows = rows; p.R = R; p.P = P;
    p.GuT = reinterpret_cast<const uint16_t*>(gut_bf16); p.ldg = ldg; p.y1 = y1; p.ldy = GH; p.mpos = m_pos; p.mneg = m_neg; p.mzero = m_zero;
    p.W1 = w1; p.bscale = bscale; p.dy1 = dy1; p.lddy = GH; p.dtt = reinterpret_cast<uint16_t*>(dtt_bf16); p.lddt = ld_dtt;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mask_rows_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(mask_rows_bwd_kernel, dim3((P + 31) / 32), dim3(512), (size_t)lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
