// Weight gradient of a Linear layer: dW [M,N] (f32) += g^T x for row-major bf16 g [R,M] (grad of the layer's
// output) and x [R,N] (its input), plus optionally db [M] += column sums of g -- the `weight.grad` / `bias.grad`
// of every nn.Linear in the encoder layer (graphormer/model.py:388-403, 406-463; autograd of F.linear).
//
// Shape of the problem at MobGT's sizes: the output is tiny (192x192 ... 576x192) and the contraction runs over
// the R = G*T rows of the batch (2-13 k).  A library GEMM call maps that onto 9-27 workgroups of 4 waves, each
// walking the whole of R: pure latency.  Here a workgroup owns one 32x32 output tile and its SIXTEEN waves
// split R between them (wave w takes row slabs w, w+16, ...); partial tiles meet in LDS and one thread per
// output element sums the 16 partials: no partial-sum buffer, no second launch.  A first version split R over
// ~340 four-wave workgroups with f32 atomics on every partial tile (1.4 M atomic lanes at R = 2432) and took
// 30-50 us; cross-workgroup splitting is therefore limited to the 2-7 ways that fill the chip (see the launcher).
//
// Both operands are contracted over their ROW index, so the 8 consecutive k-values an MFMA lane supplies live
// in 8 different rows.  No LDS transpose: lane (i, kq) of v_mfma_f32_16x16x32_bf16 reads one dword = columns
// (2i, 2i+1) from each of its 8 rows, splits low / high halves into an "even-column" and an "odd-column"
// operand, and the 2x2 MFMAs produce the 32x32 tile with rows 2i+a and columns 2j+b.
#include "common.h"
#include "mobgt_hip.h"
#include "gemm_body.h"
#include <stdlib.h>

namespace {

constexpr int TILE = 32;
constexpr int KSTEP = 32;            // rows contracted by one 16x16x32 MFMA
constexpr int SHORT_R = 1024;        // up to here 8 waves per workgroup, 16 beyond
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct WgradParams {
    const uint16_t* g;  int64_t ldg;     // [R,M] bf16
    const uint16_t* x;  int64_t ldx;     // [R,N] bf16
    float* dw;  int64_t ldw;             // [M,N] f32, accumulated
    float* db;                           // [M] f32, accumulated, or null
    int R, M, N;
    int tiles_n;
    int k_per_wg;                        // multiple of NWAVE * KSTEP when gridDim.y > 1
    int in_f32;                          // 1: operands are f32 in memory (ldg / ldx in f32 elements), rounded to bf16 here;
                                         // 2 (single-problem launches): g bf16, x f32
    // f32 operands only: g (x) is multiplied by m(gmask) (m(xmask)) while loading, m(y) = y > 0 ? mpos : (y < 0 ? mneg :
    // mzero) -- the derivative of dropout(leaky_relu(.)) from its output, so that the gradient at the pre-activation never
    // exists as a tensor.  A mask has its operand's layout and leading dimension.
    const float* gmask;
    const float* xmask;
    float mpos, mneg, mzero;
    int db_x;                            // db [N] += column sums of (the masked) x instead of g
    const float* out_bias;               // [N] or null: added to every row of dw (the product used as  A^T B + bias)
    float* gm_out;                       // or null: the masked g (f32, g's layout) written out by the first tile column --
                                         // the data-gradient GEMM that follows then needs no elementwise launch either
};

__device__ __forceinline__ void split_pairs(const uint32_t (&d)[8], bf16x8& even, bf16x8& odd) {
    uint32_t e[4], o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        e[j] = __builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], 0x05040100u);      // lo16(d0) | lo16(d1) << 16
        o[j] = __builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], 0x07060302u);      // hi16(d0) | hi16(d1) << 16
    }
    even = __builtin_bit_cast(bf16x8, e);
    odd = __builtin_bit_cast(bf16x8, o);
}

// two adjacent f32 columns -> one dword of two bf16 (round to nearest even, like a cast kernel in front would)
__device__ __forceinline__ uint32_t pack_pair(const float* p) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const float2 v = *reinterpret_cast<const float2*>(p);
    bf16x2 o;
    o[0] = (bf16_t)v.x;
    o[1] = (bf16_t)v.y;
    return __builtin_bit_cast(uint32_t, o);
}

__device__ __forceinline__ uint32_t pack_pair_masked(const float* p, const float* m, float pos, float neg, float zer,
                                                     float* keep = nullptr) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    float2 v = *reinterpret_cast<const float2*>(p);
    const float2 y = *reinterpret_cast<const float2*>(m);
    v.x *= y.x > 0.f ? pos : (y.x < 0.f ? neg : zer);
    v.y *= y.y > 0.f ? pos : (y.y < 0.f ? neg : zer);
    if (keep) *reinterpret_cast<float2*>(keep) = v;
    bf16x2 o;
    o[0] = (bf16_t)v.x;
    o[1] = (bf16_t)v.y;
    return __builtin_bit_cast(uint32_t, o);
}

struct Slab {
    uint32_t g[8], x[8];
    template <bool F32>
    __device__ __forceinline__ void load_masked(const WgradParams& p, const void* gp, const void* xp, const float* gm, const float* xm,
                                                int r0, int k1, bool m_ok, bool n_ok, float* gkeep = nullptr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = r0 + j;
            const bool ok = r < k1;
            const float* gq = reinterpret_cast<const float*>(gp) + (int64_t)r * p.ldg;
            const float* xq = reinterpret_cast<const float*>(xp) + (int64_t)r * p.ldx;
            g[j] = !(ok && m_ok) ? 0u : (gm ? pack_pair_masked(gq, gm + (int64_t)r * p.ldg, p.mpos, p.mneg, p.mzero,
                                                               gkeep ? gkeep + (int64_t)r * p.ldg : nullptr) : pack_pair(gq));
            x[j] = !(ok && n_ok) ? 0u : (xm ? pack_pair_masked(xq, xm + (int64_t)r * p.ldx, p.mpos, p.mneg, p.mzero) : pack_pair(xq));
        }
    }
    // gp / xp point at this lane's column pair of row 0 (as bf16 elements, or -- F32 -- as f32 elements)
    template <bool GF32, bool XF32>
    __device__ __forceinline__ void load(const void* gp, const void* xp, int64_t ldg, int64_t ldx, int r0, int k1,
                                         bool m_ok, bool n_ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = r0 + j;
            const bool ok = r < k1;
            if (GF32) g[j] = ok && m_ok ? pack_pair(reinterpret_cast<const float*>(gp) + (int64_t)r * ldg) : 0u;
            else g[j] = ok && m_ok ? *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(gp) + (int64_t)r * ldg) : 0u;
            if (XF32) x[j] = ok && n_ok ? pack_pair(reinterpret_cast<const float*>(xp) + (int64_t)r * ldx) : 0u;
            else x[j] = ok && n_ok ? *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(xp) + (int64_t)r * ldx) : 0u;
        }
    }
};

// NWAVE waves per workgroup: 16, or 8 when R is short (R = 608: 19 k-steps -- sixteen waves would mostly idle, and a
// 1024-thread workgroup leaves room for only two per CU where the grouped launch wants 528 of them at once)
// LDS (owned by the kernel, so that the f32 / bf16 instantiations and a passenger GEMM body overlay one buffer): NWAVE/2
// partial tiles -- the upper half of the waves hand their tiles to the lower half first -- and NWAVE column partials
template <int NWAVE> constexpr int wgrad_lds_floats() { return (NWAVE / 2) * TILE * TILE + NWAVE * TILE; }

template <bool F32, int NWAVE, bool XF32 = F32>
__device__ __forceinline__ void wgrad_body(const WgradParams& p, const int tile, const int split, const int nsplit,
                                           float* __restrict__ lds) {
    float (*part)[TILE * TILE] = reinterpret_cast<float (*)[TILE * TILE]>(lds);
    float (*colpart)[TILE] = reinterpret_cast<float (*)[TILE]>(lds + (NWAVE / 2) * TILE * TILE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int m0 = (tile / p.tiles_n) * TILE, n0 = (tile % p.tiles_n) * TILE;
    const int k0 = split * p.k_per_wg;
    const int k1 = min(p.R, k0 + p.k_per_wg);
    const bool want_db = p.db != nullptr && (p.db_x ? m0 == 0 : n0 == 0);
    const bool m_ok = m0 + 2 * i < p.M, n_ok = n0 + 2 * i < p.N;      // M, N even: a pair is in or out together
    const void* gp = F32 ? (const void*)(reinterpret_cast<const float*>(p.g) + m0 + 2 * i) : (const void*)(p.g + m0 + 2 * i);
    const void* xp = XF32 ? (const void*)(reinterpret_cast<const float*>(p.x) + n0 + 2 * i) : (const void*)(p.x + n0 + 2 * i);

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float se = 0.f, so = 0.f;

    const bool masked = F32 && (p.gmask || p.xmask);
    const float* gm = masked && p.gmask ? p.gmask + m0 + 2 * i : nullptr;
    const float* xm = masked && p.xmask ? p.xmask + n0 + 2 * i : nullptr;
    float* gkeep = (masked && gm && p.gm_out && n0 == 0) ? p.gm_out + m0 + 2 * i : nullptr;
    auto fetch = [&](Slab& s_, int r0_) {
        if (masked) s_.template load_masked<F32>(p, gp, xp, gm, xm, r0_, k1, m_ok, n_ok, gkeep);
        else s_.template load<F32, XF32>(gp, xp, p.ldg, p.ldx, r0_, k1, m_ok, n_ok);
    };
    int kb = k0 + wave * KSTEP;
    Slab cur, nxt;
    if (kb < k1) fetch(cur, kb + 8 * kq);
    for (; kb < k1; kb += NWAVE * KSTEP) {
        const int kn = kb + NWAVE * KSTEP;
        if (kn < k1) fetch(nxt, kn + 8 * kq);      // in flight during the MFMAs
        bf16x8 ge, go, xe, xo;
        split_pairs(cur.g, ge, go);
        split_pairs(cur.x, xe, xo);
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ge, xe, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ge, xo, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(go, xe, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(go, xo, acc[1][1], 0, 0, 0);
        if (want_db) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t d = p.db_x ? cur.x[j] : cur.g[j];
                se += bf16_lo(d); so += bf16_hi(d);
            }
        }
        if (kn < k1) cur = nxt;
    }

    // register v of lane (j = lane & 15, q = lane >> 4) is MFMA row 4q + v, column j; operand row i / column j
    // stand for tile rows 2i+a and columns 2j+b
    if (wave >= NWAVE / 2) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) part[wave - NWAVE / 2][(2 * (4 * kq + v) + a) * TILE + 2 * i + b] = acc[a][b][v];
    }
    __syncthreads();
    if (wave < NWAVE / 2) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) part[wave][(2 * (4 * kq + v) + a) * TILE + 2 * i + b] += acc[a][b][v];
    }
    if (want_db) {
        se += __shfl_xor(se, 16, 64); se += __shfl_xor(se, 32, 64);
        so += __shfl_xor(so, 16, 64); so += __shfl_xor(so, 32, 64);
        if (kq == 0) { colpart[wave][2 * i] = se; colpart[wave][2 * i + 1] = so; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TILE * TILE; e += NWAVE * 64) {
        const int r = e >> 5, c = e & 31;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE / 2; ++w) s += part[w][e];
        if (m0 + r < p.M && n0 + c < p.N) {
            float* dst = p.dw + (int64_t)(m0 + r) * p.ldw + n0 + c;
            if (p.out_bias && split == 0) s += p.out_bias[n0 + c];
            // always the atomic form: nothing waits for its result, where `*dst += s` ends every workgroup on a load round trip
            atomicAdd(dst, s);
        }
    }
    {
        const int e = threadIdx.x;
        const int c0 = p.db_x ? n0 : m0;
        if (want_db && e < TILE && c0 + e < (p.db_x ? p.N : p.M)) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) t += colpart[w][e];
            atomicAdd(p.db + c0 + e, t);
        }
    }
}

template <int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void wgrad_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) float lds[wgrad_lds_floats<NWAVE>()];
    if (p.in_f32 == 2) wgrad_body<false, NWAVE, true>(p, blockIdx.x, blockIdx.y, gridDim.y, lds);      // g bf16, x f32
    else if (p.in_f32) wgrad_body<true, NWAVE>(p, blockIdx.x, blockIdx.y, gridDim.y, lds);
    else wgrad_body<false, NWAVE>(p, blockIdx.x, blockIdx.y, gridDim.y, lds);
}

// Up to 32 independent problems in ONE launch (the four Linear layers of every encoder layer: nothing depends on a
// weight gradient, so they are all issued together at the end of the backward pass).  A kernel inside a replayed graph
// costs ~4.5 us whatever it does; three fewer launches per layer are worth more than any tuning of the kernel.
constexpr int WG_GROUP = 32;
struct WgradGroup {
    WgradParams p[WG_GROUP];
    int first[WG_GROUP + 1];           // first workgroup of each problem; first[n] = total
    int tiles[WG_GROUP], splits[WG_GROUP];
    int n;
    // Optional passenger (8-wave launches only): the layer's last data-gradient GEMM dx = dx1 + dqkv Wqkv, which depends
    // on the same dqkv as the weight gradients and on nothing they produce -- its n_tail workgroups lead the grid.
    mobgt_gemm::GemmParams tail;
    int n_tail;
};

template <int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void wgrad_group_kernel(const WgradGroup grp) {
    constexpr int LDS_W = wgrad_lds_floats<NWAVE>(), LDS_G = NWAVE == 8 ? mobgt_gemm::gemm_lds_floats<2, 8>() : 0;
    __shared__ __attribute__((aligned(16))) float lds[LDS_W > LDS_G ? LDS_W : LDS_G];      // one body runs per workgroup
    if constexpr (NWAVE == 8) {
        if ((int)blockIdx.x < grp.n_tail) {
            mobgt_gemm::layer_gemm_body<true, mobgt_gemm::EPI_ADD, 2, 8>(grp.tail, blockIdx.x, lds);
            return;
        }
    }
    const int bid = (int)blockIdx.x - grp.n_tail;
    int q = 0;
#pragma unroll
    for (int t = 1; t < WG_GROUP; ++t)
        if (t < grp.n && bid >= grp.first[t]) q = t;
    const int local = bid - grp.first[q];
    if (grp.p[q].in_f32) wgrad_body<true, NWAVE>(grp.p[q], local % grp.tiles[q], local / grp.tiles[q], grp.splits[q], lds);
    else wgrad_body<false, NWAVE>(grp.p[q], local % grp.tiles[q], local / grp.tiles[q], grp.splits[q], lds);
}

int fill_problem(WgradParams& p, const void* g, int64_t ldg, const void* x, int64_t ldx, float* dw, int64_t ldw, float* db,
                 int64_t R, int M, int N, int target_wgs, int* tiles_out, int* splits_out, int in_f32, int nwave) {
    if (R <= 0 || M <= 0 || N <= 0 || (M & 1) || (N & 1) || (ldg & 1) || (ldx & 1) || R > 0x7fffffff) return MOBGT_EBADDIM;
    if (((uintptr_t)g & (in_f32 == 1 ? 7 : 3)) || ((uintptr_t)x & (in_f32 ? 7 : 3))) return MOBGT_EALIGN;
    p.in_f32 = in_f32;
    p.gmask = p.xmask = nullptr; p.mpos = p.mneg = p.mzero = 1.f; p.db_x = 0; p.out_bias = nullptr; p.gm_out = nullptr;
    p.g = reinterpret_cast<const uint16_t*>(g); p.ldg = ldg;
    p.x = reinterpret_cast<const uint16_t*>(x); p.ldx = ldx;
    p.dw = dw; p.ldw = ldw; p.db = db;
    p.R = (int)R; p.M = M; p.N = N;
    const int tiles_m = (M + TILE - 1) / TILE;
    p.tiles_n = (N + TILE - 1) / TILE;
    const int tiles = tiles_m * p.tiles_n;
    const int slab = nwave * KSTEP;
    int splits = target_wgs / tiles;
    const int max_splits = (int)((R + slab - 1) / slab);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.k_per_wg = (int)(((R + splits - 1) / splits + slab - 1) / slab) * slab;
    *splits_out = (int)((R + p.k_per_wg - 1) / p.k_per_wg);
    *tiles_out = tiles;
    return 0;
}

}  // namespace

namespace {
int launch_group(int n, const void* const* g, const int64_t* ldg, const void* const* x, const int64_t* ldx, float* const* dw,
                 const int64_t* ldw, float* const* db, int64_t R, const int* M, const int* N, int act_dtype,
                 const mobgt_gemm::GemmParams* tail, void* stream) {
    if (act_dtype != MOBGT_BF16 && act_dtype != MOBGT_F32) return MOBGT_EDTYPE;
    if (n < 1 || n > WG_GROUP) return MOBGT_EBADDIM;
    if (R == 0) return 0;
    WgradGroup grp;
    grp.n = n;
    grp.n_tail = 0;
    int total = 0;
    const int nwave = R <= SHORT_R ? 8 : 16;
    if (tail) {
        if (nwave != 8) return MOBGT_EBADDIM;
        grp.tail = *tail;
        grp.n_tail = ((tail->M + 31) / 32) * ((tail->N + 31) / 32);
    }
    for (int q = 0; q < n; ++q) {
        // the problems share the chip: aim at ~256 workgroups for all of them together
        const int rc = fill_problem(grp.p[q], g[q], ldg[q], x[q], ldx[q], dw[q], ldw[q], db ? db[q] : nullptr, R, M[q], N[q],
                                    256 / n, &grp.tiles[q], &grp.splits[q], act_dtype == MOBGT_F32, nwave);
        if (rc) return rc;
        grp.first[q] = total;
        total += grp.tiles[q] * grp.splits[q];
    }
    for (int q = n; q <= WG_GROUP; ++q) grp.first[q] = total;
    for (int q = n; q < WG_GROUP; ++q) { grp.tiles[q] = 1; grp.splits[q] = 1; grp.p[q] = grp.p[0]; }
    if (nwave == 8) hipLaunchKernelGGL(wgrad_group_kernel<8>, dim3(total + grp.n_tail), dim3(8 * 64), 0, (hipStream_t)stream, grp);
    else hipLaunchKernelGGL(wgrad_group_kernel<16>, dim3(total), dim3(16 * 64), 0, (hipStream_t)stream, grp);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int mobgt_linear_wgrad_group(int n, const void* const* g, const int64_t* ldg, const void* const* x,
                                        const int64_t* ldx, float* const* dw, const int64_t* ldw, float* const* db,
                                        int64_t R, const int* M, const int* N, int act_dtype, void* stream) {
    return launch_group(n, g, ldg, x, ldx, dw, ldw, db, R, M, N, act_dtype, nullptr, stream);
}

extern "C" int mobgt_layer_backward_tail(int n, const void* const* g, const int64_t* ldg, const void* const* x,
                                         const int64_t* ldx, float* const* dw, const int64_t* ldw, float* const* db,
                                         int64_t R, const int* M, const int* N, int act_dtype, const void* a, int64_t lda,
                                         const void* b_kn, int64_t ldb, float* c, int64_t ldc, int gM, int gN, int gK,
                                         void* stream) {
    if (gM <= 0 || gN <= 0 || gK <= 0 || (gK % 32) || (gN & 7) || (lda & 7) || (ldc & 7) || (ldb & 1)) return MOBGT_EBADDIM;
    if ((((uintptr_t)a | (uintptr_t)c) & 15) || ((uintptr_t)b_kn & 3)) return MOBGT_EALIGN;
    mobgt_gemm::GemmParams t = {reinterpret_cast<const uint16_t*>(a), lda, reinterpret_cast<const uint16_t*>(b_kn), ldb, nullptr,
                                c, ldc, c, nullptr, gM, gN, gK};
    return launch_group(n, g, ldg, x, ldx, dw, ldw, db, R, M, N, act_dtype, &t, stream);
}

extern "C" int mobgt_linear_wgrad(const void* g, int64_t ldg, const void* x, int64_t ldx, float* dw, int64_t ldw,
                                  float* db, int64_t R, int M, int N, int act_dtype, void* stream) {
    if (act_dtype != MOBGT_BF16 && act_dtype != MOBGT_F32) return MOBGT_EDTYPE;
    if (R == 0) return 0;
    // A 1024-thread workgroup fills a CU, so up to ~256 workgroups run at once: when the tiles alone do not
    // reach that, R is also split over gridDim.y (>= one 512-row slab per workgroup) and the few partial tiles
    // per output are combined with f32 atomics.  Measured (MI355X, R = 2432): 192x192 5 splits 7.5 us vs 12.9 us
    // unsplit; 576x192 2 splits 10.8 vs 13.3; R = 12560, 256x256 4 splits 21 vs 46 (incl. two zero fills).
    WgradParams p;
    int tiles = 0, splits = 0;
    const int nwave = R <= SHORT_R ? 8 : 16;
    const int rc = fill_problem(p, g, ldg, x, ldx, dw, ldw, db, R, M, N, nwave == 8 ? 512 : 256, &tiles, &splits,
                                act_dtype == MOBGT_F32, nwave);
    if (rc) return rc;
    if (nwave == 8) hipLaunchKernelGGL(wgrad_kernel<8>, dim3(tiles, splits), dim3(8 * 64), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wgrad_kernel<16>, dim3(tiles, splits), dim3(16 * 64), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

/* f32 operands with the derivative of dropout(leaky_relu(.)) applied to g and / or x while loading (see WgradParams). */
extern "C" int mobgt_linear_wgrad_masked(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* g_mask,
                                         const float* x_mask, float m_pos, float m_neg, float m_zero, float* g_masked_out,
                                         float* dw, int64_t ldw, float* db, int db_of_x, int64_t R, int M, int N, void* stream) {
    if (R == 0) return 0;
    if (((uintptr_t)g_mask | (uintptr_t)x_mask) & 7) return MOBGT_EALIGN;
    WgradParams p;
    int tiles = 0, splits = 0;
    const int nwave = R <= SHORT_R ? 8 : 16;
    const int rc = fill_problem(p, g, ldg, x, ldx, dw, ldw, db, R, M, N, nwave == 8 ? 512 : 256, &tiles, &splits, 1, nwave);
    if (rc) return rc;
    p.gmask = g_mask; p.xmask = x_mask; p.mpos = m_pos; p.mneg = m_neg; p.mzero = m_zero; p.db_x = db_of_x;
    p.gm_out = g_mask ? g_masked_out : nullptr;
    if (nwave == 8) hipLaunchKernelGGL(wgrad_kernel<8>, dim3(tiles, splits), dim3(8 * 64), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wgrad_kernel<16>, dim3(tiles, splits), dim3(16 * 64), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

/* dw [M,N] (zero on entry) = g^T x + out_bias[n] on every row: the split-K product used as a skinny "A^T B + bias"
 * (the last GraphConvolution on the batch's rows, modelGNN.py:38-44). */
extern "C" int mobgt_linear_wgrad_bias(const void* g, int64_t ldg, const void* x, int64_t ldx, const float* out_bias, float* dw,
                                       int64_t ldw, int64_t R, int M, int N, int act_dtype, void* stream) {
    if (act_dtype != MOBGT_BF16 && act_dtype != MOBGT_F32) return MOBGT_EDTYPE;
    if (R == 0) return 0;
    WgradParams p;
    int tiles = 0, splits = 0;
    const int nwave = R <= SHORT_R ? 8 : 16;
    const int rc = fill_problem(p, g, ldg, x, ldx, dw, ldw, nullptr, R, M, N, nwave == 8 ? 512 : 256, &tiles, &splits,
                                act_dtype == MOBGT_F32, nwave);
    if (rc) return rc;
    p.out_bias = out_bias;
    if (nwave == 8) hipLaunchKernelGGL(wgrad_kernel<8>, dim3(tiles, splits), dim3(8 * 64), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wgrad_kernel<16>, dim3(tiles, splits), dim3(16 * 64), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

/* g bf16 [R,M], x f32 [R,N] (rounded to bf16 while loading): dw (ZERO on entry) = g^T x, db [N] (zero on entry, or null)
 * += column sums of x -- the backward of the rows-only GraphConvolution: d(support) = adj[rows]^T dout and
 * b.grad = dout.sum(0) from the f32 gradient as it arrives (no cast, no column-sum launch). */
extern "C" int mobgt_linear_wgrad_mixed(const void* g_bf16, int64_t ldg, const float* x_f32, int64_t ldx, float* dw, int64_t ldw,
                                        float* db_x, int64_t R, int M, int N, void* stream) {
    if (R == 0) return 0;
    WgradParams p;
    int tiles = 0, splits = 0;
    const int nwave = R <= SHORT_R ? 8 : 16;
    const int rc = fill_problem(p, g_bf16, ldg, x_f32, ldx, dw, ldw, db_x, R, M, N, nwave == 8 ? 512 : 256, &tiles, &splits, 2, nwave);
    if (rc) return rc;
    p.db_x = 1;
    if (nwave == 8) hipLaunchKernelGGL(wgrad_kernel<8>, dim3(tiles, splits), dim3(8 * 64), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wgrad_kernel<16>, dim3(tiles, splits), dim3(16 * 64), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
