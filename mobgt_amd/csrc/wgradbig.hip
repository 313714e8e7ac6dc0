// Weight gradients of one encoder layer for LONG batches (past 4 096 rows), all of them in ONE launch (round 6).
//
// Replaces, at S-BIG sizes (R = 12 560 rows, C = 256, F = 1 024), three library split-K GEMMs (24-26 us each whatever their size),
// the grouped 32 x 32-tile kernel of wgrad.hip for dWo (30 us) and a column-sum launch: autograd of the layer's four F.linear
// calls (graphormer/model.py:436-438 linear_q/k/v, :455 output_layer, :393-405 layer1 / layer2):
//     dW[M, N] = G^T X,   G = d(output) [R, M],  X = the Linear's input [R, N],  both row-major bf16,  f32 result
//     dWqkv = dqkv^T xa (768 x 256) + column sums of dqkv;  dWo = dy^T a;  dW1 = du^T z (1024 x 256);  dW2 = df^T h (256 x 1024)
// 19.8 GFLOP and 103 MB of operands per layer: HBM-bound at 17 us, not MFMA-bound (8 us at peak).
//
// Shape of the work.  K = R is the long axis and the outputs are small (786 k elements per layer = 24 tiles of 128 x 256), so the
// rows are split: a workgroup owns one 128 (M) x 256 (N) output tile and one of S row ranges, and writes its f32 partial tile into
// slice s of the product's [S, M, N] partial buffer (S = 10 at S-BIG: 240 workgroups for 256 compute units, 31 MB of partials per
// layer).  The sum over slices is NOT done here: inside a train step all of a step's partial buffers are summed into the flat
// gradient buffer by ONE launch at the end of the backward pass (mobgt_partial_sum_multi, as for the library's split-K
// partials before); f32 atomics instead would cost 24 us per layer at the memory side's 1.3 TB/s of added bytes.
// Workgroups of one row range are neighbours in launch order and share an XCD (xcd_remap), so a range's operand rows are
// fetched from HBM about once and re-read by the other tiles from that XCD's L2 (231 MB of L2 reads per layer).
//
// Both operands are "K-major the wrong way" for the matrix cores: G[k][m] and X[k][n] are row-major in k, and an MFMA lane wants
// eight consecutive k of one m (or n).  The 64-row chunks are therefore staged in LDS exactly as they lie in memory (16-byte
// pieces, row pitch + 64 B so that the four rows x two column blocks a 32-lane half reads fall on 64 distinct banks) and read back with
// ds_read_b64_tr_b16: two of them give a lane its 8 k-values of v_mfma_f32_32x32x16_bf16's A (from G) or B (from X) operand.
// Eight waves as 2 (M) x 4 (N), 64 x 64 outputs each: per 16-row k-step a wave issues 8 transposing reads and 4 MFMAs.
// Chunk c + 1 sits in registers while chunk c is multiplied (buffer loads: rows beyond R and columns beyond M / N come back as
// zeros through out-of-range offsets, no branch in the loop) and goes into the other LDS image before the chunk's one barrier.
#include "common.h"
#include "mobgt_hip.h"

namespace {

typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
typedef short v4s_w __attribute__((ext_vector_type(4)));
typedef short v8s_w __attribute__((ext_vector_type(8)));

constexpr int WB_TM = 128, WB_TN = 256, WB_KC = 64, WB_NT = 512;
constexpr int WB_GP = WB_TM + 32;        // LDS row pitch of the G image in bf16: 320 B = 80 dwords (16 mod 64: the 4 rows x 2 column blocks of a 32-lane half fall on 64 distinct banks)
constexpr int WB_XP = WB_TN + 32;        // ... of the X image: 576 B = 144 dwords (16 mod 64)
constexpr int WB_MAXJ = 4;

struct WbJob {
    const bf16_t* g;
    const bf16_t* x;
    float* part;          // [S, M, N]
    float* colsum;        // [M] (accumulated by atomics) or null
    int64_t ldg, ldx;
    int M, N, nt, tile0;  // nt: N tiles; tile0: first tile id of this product
};
struct WbParams {
    WbJob job[WB_MAXJ];
    int njobs, ntiles, S, nchunk;
    int64_t R;
};

__device__ __forceinline__ v4s_w wb_tr16(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_w __attribute__((address_space(3)))*)(p));
}

__global__ __launch_bounds__(WB_NT) void layer_wgrad_big_kernel(const WbParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t Gs[2][WB_KC][WB_GP];
    __shared__ __attribute__((aligned(16))) bf16_t Xs[2][WB_KC][WB_XP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int s = lid / p.ntiles, t = lid % p.ntiles;
    int j = 0;
#pragma unroll
    for (int q = 1; q < WB_MAXJ; ++q)
        if (q < p.njobs && t >= p.job[q].tile0) j = q;
    const WbJob& jb = p.job[j];
    const int tt = t - jb.tile0;
    const int mi = tt / jb.nt, ni = tt % jb.nt;
    const int m0 = mi * WB_TM, n0 = ni * WB_TN;
    const int M = jb.M, N = jb.N;
    const int c0 = (int)((int64_t)s * p.nchunk / p.S), c1 = (int)((int64_t)(s + 1) * p.nchunk / p.S);

    // operands through buffer descriptors: an offset beyond the R x ld elements (a row >= R) reads zeros; a column beyond the
    // product's M / N gets such an offset by hand
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(jb.g), 0, (int)(p.R * jb.ldg * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(jb.x), 0, (int)(p.R * jb.ldx * 2), 0x00020000);
    // staging: G chunk = 64 rows x 16 pieces of 8 columns (2 per thread), X chunk = 64 rows x 32 pieces (4 per thread)
    const int gcg = tid & 15, grow = tid >> 4;          // + 32 rows for the second piece
    const int xcg = tid & 31, xrow = tid >> 5;          // + 16 rows per further piece
    const bool gcol_ok = m0 + gcg * 8 < M, xcol_ok = n0 + xcg * 8 < N;
    const uint32_t gbase = gcol_ok ? (uint32_t)((m0 + gcg * 8) * 2) : 0x80000000u;
    const uint32_t xbase = xcol_ok ? (uint32_t)((n0 + xcg * 8) * 2) : 0x80000000u;
    const uint32_t gstep = (uint32_t)(jb.ldg * 2), xstep = (uint32_t)(jb.ldx * 2);
    u32x4w greg[2], xreg[4];
    auto stage_load = [&](const int c) {
        const uint32_t r0 = (uint32_t)(c * WB_KC);
#pragma unroll
        for (int q = 0; q < 2; ++q) greg[q] = __builtin_amdgcn_raw_buffer_load_b128(grs, gbase + (r0 + grow + 32 * q) * gstep, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) xreg[q] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xbase + (r0 + xrow + 16 * q) * xstep, 0, 0);
    };
    const bool want_cs = jb.colsum != nullptr && ni == 0;          // (workgroup-uniform)
    float cs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cs[i] = 0.f;
    auto stage_store = [&](const int b) {
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4w*>(&Gs[b][grow + 32 * q][gcg * 8]) = greg[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4w*>(&Xs[b][xrow + 16 * q][xcg * 8]) = xreg[q];
        if (want_cs) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) { cs[2 * k] += bf16_lo(greg[q][k]); cs[2 * k + 1] += bf16_hi(greg[q][k]); }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int wm = wave >> 2, wn = wave & 3;
    // a lane's corner inside a transposing read: row (k) offset and column offset within a 32-wide block
    const int trk = 8 * (lane >> 5) + ((lane & 15) >> 2), trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    if (c0 < c1) {
        stage_load(c0);
        stage_store(0);
        stage_load(min(c0 + 1, c1 - 1));
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
        const int b = (c - c0) & 1;
#pragma unroll
        for (int ks = 0; ks < WB_KC / 16; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bf16_t* ga = &Gs[b][ks * 16 + trk][64 * wm + 32 * q + trc];
                const v4s_w r0 = wb_tr16(ga), r1 = wb_tr16(ga + 4 * WB_GP);
                const v8s_w av = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
                af[q] = __builtin_bit_cast(bf16x8, av);
                const bf16_t* xa = &Xs[b][ks * 16 + trk][64 * wn + 32 * q + trc];
                const v4s_w s0 = wb_tr16(xa), s1 = wb_tr16(xa + 4 * WB_XP);
                const v8s_w bv = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
                bfr[q] = __builtin_bit_cast(bf16x8, bv);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[a][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[q], acc[a][q], 0, 0, 0);
        }
        // chunk c + 1 (in registers since the last iteration) into the other image -- its readers finished before the last
        // barrier --, chunk c + 2 requested
        if (c + 1 < c1) {
            stage_store(b ^ 1);
            stage_load(min(c + 2, c1 - 1));
        }
        __syncthreads();
    }

    // ---- the partial tile: acc register r of lane l = output (m0 + 64 wm + 32 a + 8 (r / 4) + 4 (l / 32) + r % 4, n0 + 64 wn + 32 q + l % 32)
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(jb.part + (int64_t)s * M * N, 0, (int)((int64_t)M * N * 4), 0x00020000);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = n0 + 64 * wn + 32 * q + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 64 * wm + 32 * a + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                const uint32_t off = (m < M && n < N) ? (uint32_t)(m * N + n) * 4u : 0x80000000u;
                // (through a scalar copy: __builtin_bit_cast applied to the vector ELEMENT expression acc[a][q][r] made hipcc 7.2
                //  store element 0 sixteen times -- seen in the ISA: one data register for all 16 stores)
                const float val = acc[a][q][r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), prs, off, 0, 0);
            }
        }

    // ---- column sums of G (the Linear's bias gradient), from the pieces this thread staged: 32 threads share a column group
    if (want_cs) {
        float* red = reinterpret_cast<float*>(&Xs[0][0][0]);        // (every read of the images is behind the loop's last barrier)
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid * 8 + i] = cs[i];
        __syncthreads();
        if (tid < WB_TM) {
            const int cg = tid >> 3, e = tid & 7;
            float v = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) v += red[(cg + 16 * k) * 8 + e];
            if (m0 + tid < M) atomicAdd(jb.colsum + m0 + tid, v);
        }
    }
}

}  // namespace

// Recommended number of row ranges for a launch of `ntiles` output tiles over R rows: about one workgroup per compute unit, at
// least two 64-row chunks per range.
extern "C" int mobgt_layer_wgrad_big_splits(int64_t R, int ntiles) {
    if (R <= 0 || ntiles <= 0) return MOBGT_EBADDIM;
    const int nchunk = (int)((R + WB_KC - 1) / WB_KC);
    int S = (240 + ntiles / 2) / ntiles;
    if (S > nchunk / 2) S = nchunk / 2;
    if (S < 1) S = 1;
    if (S > 32) S = 32;
    return S;
}

extern "C" int mobgt_layer_wgrad_big_tiles(int M, int N) {
    if (M <= 0 || N <= 0) return MOBGT_EBADDIM;
    return ((M + WB_TM - 1) / WB_TM) * ((N + WB_TN - 1) / WB_TN);
}

extern "C" int mobgt_layer_wgrad_big(int n, const void* const* g, const int64_t* ldg, const void* const* x, const int64_t* ldx,
                                     float* const* part, float* const* colsum, const int* M, const int* N, int64_t R, int S,
                                     void* stream) {
    if (n <= 0 || n > WB_MAXJ || R <= 0 || S <= 0 || S > 64) return MOBGT_EBADDIM;
    WbParams p{};
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (M[i] <= 0 || N[i] <= 0 || (M[i] & 7) || (N[i] & 7) || ldg[i] < M[i] || ldx[i] < N[i]) return MOBGT_EBADDIM;
        if ((ldg[i] & 7) || (ldx[i] & 7) || (((uintptr_t)g[i] | (uintptr_t)x[i] | (uintptr_t)part[i]) & 15)) return MOBGT_EALIGN;
        if (R * ldg[i] * 2 >= (int64_t)1 << 31 || R * ldx[i] * 2 >= (int64_t)1 << 31 || (int64_t)M[i] * N[i] * 4 >= (int64_t)1 << 31)
            return MOBGT_EBADDIM;
        WbJob& jb = p.job[i];
        jb.g = reinterpret_cast<const bf16_t*>(g[i]);
        jb.x = reinterpret_cast<const bf16_t*>(x[i]);
        jb.part = part[i];
        jb.colsum = colsum ? colsum[i] : nullptr;
        jb.ldg = ldg[i];
        jb.ldx = ldx[i];
        jb.M = M[i];
        jb.N = N[i];
        jb.nt = (N[i] + WB_TN - 1) / WB_TN;
        jb.tile0 = tiles;
        tiles += ((M[i] + WB_TM - 1) / WB_TM) * jb.nt;
    }
    p.njobs = n;
    p.ntiles = tiles;
    p.S = S;
    p.R = R;
    p.nchunk = (int)((R + WB_KC - 1) / WB_KC);
    if (S > p.nchunk) return MOBGT_EBADDIM;
    hipLaunchKernelGGL(layer_wgrad_big_kernel, dim3(tiles * S), dim3(WB_NT), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
