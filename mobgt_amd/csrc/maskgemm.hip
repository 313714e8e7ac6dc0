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
