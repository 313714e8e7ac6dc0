// Body of the split-K weight-gradient kernel (see wgrad.hip for the design notes), shared by wgrad.hip and by the backward
// chain kernel of chain.hip, which hosts the previous layer's four weight-gradient problems on the compute units its own
// 16-row workgroups leave idle.
#pragma once
#include "common.h"
#include "mobgt_hip.h"

namespace mobgt_wgrad {

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

inline int fill_problem(WgradParams& p, const void* g, int64_t ldg, const void* x, int64_t ldx, float* dw, int64_t ldw, float* db,
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


}  // namespace mobgt_wgrad
