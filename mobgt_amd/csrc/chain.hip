// Everything of an encoder layer that is ROW-LOCAL, as one launch (graphormer/model.py:479-489 /
// model_fqandtoyo.py:1731-1743, the fq layer: post-LayerNorm after the attention AND after the FFN):
//
//     y   = a Wo^T + bo                      output_layer of MultiHeadAttention (model.py:455)
//     x1  = x + dropout(y);  z = ffn_norm1(x1)
//     u   = z W1^T + b1;  h = gelu(u)        FeedForwardNetwork (model.py:388-403)
//     f   = h W2^T + b2
//     x2  = x1 + dropout(f);  out = ffn_norm2(x2)
//     qkv' = out Wqkv'^T + bqkv'             the NEXT layer's linear_q/k/v (model.py:436-438), when there is one
//
// Only softmax(QK^T)V mixes rows; the rest of the layer is four GEMMs against weights plus per-row work.  As separate
// launches (out-proj GEMM, LN+FFN1, FFN2, dropout+LN, QKV GEMM) these were 30 us of the S-FSQ layer's 36 us forward
// for 134 MFLOP: each launch costs ~4.5 us of ramp + one memory round trip whatever it computes.  Here a workgroup owns
// 32 rows and walks the whole chain: the 32-row activations never leave LDS (bf16 MFMA operands; the f32 residual stream
// next to them), the weights stream from L2 (1.08 MB per layer and workgroup: the per-CU L2 bandwidth, ~8 us, is the
// bound), and what the backward needs (x1, z, u, h, x2, the LayerNorm statistics) is written on the way.  The layer is
// then TWO launches forward: attention, and this.
//
// GEMM inside the workgroup: 12 waves; a wave owns 16-column groups of the output (all 32 rows, the whole K), so there is
// no split-K and no partial-tile exchange; v_mfma_f32_16x16x32_bf16, A from LDS (16 bytes per lane), B straight from
// global (W is [N,K] row-major: 16 bytes per lane, 64-byte runs per weight row), the next chunk of <= 8 k-steps in
// flight while the current one is multiplied.  Rounding points are those of the separate launches (y, u, h, f, qkv
// rounded to bf16 exactly where a bf16 tensor used to be written), and the dropout masks are the same hashes
// (mobgt_dropout_add_ln_fwd's), so the existing backward applies unchanged.
#include <type_traits>
#include "common.h"
#include "mobgt_hip.h"
#include "wgrad_body.h"
#include "pack_body.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int NW = 12, NT = NW * 64;

struct ChainParams {
    const uint16_t* a;                       // [R,C] bf16 attention output
    const float* x;                          // [R,C] f32 layer input (residual stream)
    const uint16_t *wo, *bo, *w1, *b1, *w2, *b2, *wq, *bq;      // bf16: [C,C] [C] [F,C] [F] [C,F] [C] [3C,C] [3C]  (wq null: last layer)
    const float *n1w, *n1b, *nxw, *nxb;      // ffn_norm1, ffn_norm2
    float *x1, *x2, *out;                    // [R,C] f32
    uint16_t *z, *u, *h, *out_a, *qkv;       // bf16 [R,C] [R,F] [R,F] [R,C] [R,3C]
    float *mean1, *rstd1, *mean2, *rstd2;    // [R]
    int R;
    uint32_t thr;
    float inv_keep;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt1, salt2;
    uint32_t* ws_gen;                        // cluster form: launch generations and the exchange area (see cluster_put / cluster_get)
    uint64_t* ws_ll;
    int nx;                                  // ... and the number of XCDs that take clusters
};

// in-kernel timeline (-DCH_DEBUG; the reader script left the tree in round 5): stamps of workgroup 0 / thread 0, written at the end
#ifdef CH_DEBUG
__device__ int* g_chain_dbg = nullptr;
#define STAMP_DECL int st_[16] = {}
#define STAMP(i) st_[i] = (int)wall_clock64()
#define STAMP_DUMP() do { if (g_chain_dbg && threadIdx.x == 0) for (int q_ = 0; q_ < 16; ++q_) if (st_[q_]) g_chain_dbg[blockIdx.x * 16 + q_] = st_[q_]; } while (0)
// the FIRST time a helper passes slot i (thread 0; written straight to the buffer)
#define STAMP_ONCE(i) do { if (g_chain_dbg && threadIdx.x == 0 && g_chain_dbg[blockIdx.x * 16 + (i)] == 0) g_chain_dbg[blockIdx.x * 16 + (i)] = (int)wall_clock64(); } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_DUMP()
#define STAMP_ONCE(i)
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the 16 lanes of a DPP row (every lane gets it): two quad permutes, then the mirrors of 8 and of 16 lanes --
// four v_add with a DPP operand, where a __shfl_xor butterfly is four ds_bpermute round trips (the LayerNorm passes
// were 1.5 us each with those)
__device__ __forceinline__ float dpp_add(float v, const int ctrl_sel) {
    const int x = __builtin_bit_cast(int, v);
    int y;
    if (ctrl_sel == 0) y = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    else if (ctrl_sel == 1) y = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    else if (ctrl_sel == 2) y = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false);  // row_half_mirror
    else y = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false);                     // row_mirror
    return v + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add(v, 0);
    v = dpp_add(v, 1);
    v = dpp_add(v, 2);
    return dpp_add(v, 3);
}

// sum over the wave, every lane gets it: the DPP row sum, then the two row broadcasts of gfx9 (lane 63 ends up with the
// total) and a readlane
__device__ __forceinline__ float wave64_sum(float v) {
    v = row16_sum(v);
    int x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));      // row_bcast:15 -> rows 1, 3
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));      // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of h): erff() is ~40 instructions,
// and here ONE compute unit applies GELU to 16 x 1024 pre-activations per layer
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float y = fmaf(1.061405429f, t, -1.453152027f);
    y = fmaf(y, t, 1.421413741f);
    y = fmaf(y, t, -0.284496736f);
    y = fmaf(y, t, 0.254829592f);
    y = 1.f - y * t * __expf(-ax * ax);
    return copysignf(y, x);
}
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.f + erf_as(u * 0.70710678118654752f)); }
__device__ __forceinline__ float bf16_round(float v) { return (float)(bf16_t)v; }
__device__ __forceinline__ uint16_t bf16_bits(float v) { return __builtin_bit_cast(uint16_t, (bf16_t)v); }
__device__ __forceinline__ float bf16_val(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }

// out[BM x N] = A[BM x K] (LDS, bf16, row stride LDA elements) x W[N x K]^T (global bf16, PACKED in MFMA operand order by
// mobgt_pack_mfma_b: the 16 bytes lane l = j + 16 q feeds to k-step s of column group g -- W[16 g + j][32 s + 8 q ..] -- sit
// at ((g S + s) 64 + l) * 16 bytes, so a wave's operand load is ONE contiguous KB.  Read straight from the row-major
// weight, adjacent lanes hit 16 different rows and the load unit serialises them: measured 15 bytes/clk per CU, a quarter
// of the L1 rate, 37 us for the chain).  Wave w owns the
// 16-column groups w, w + NW, ...; `epi(g, acc)` receives a finished group: acc[t][v] = out[16 t + 4 q + v][16 g + j] for
// lane (j = lane & 15, q = lane >> 4).  TWO chunks of B operands are in flight ahead of the one being multiplied (the
// per-CU L1 bandwidth is the kernel's bound: ~100 KB must be outstanding per CU to reach it); for K <= 256 the A operands
// of the whole K live in registers for all of the wave's groups.
// k-steps per chunk of B operands: the whole K when it is <= 8 steps, else the largest of 8 .. 4 that divides it
constexpr int chunk_steps(int S) { return S <= 8 ? S : (S % 8 == 0 ? 8 : (S % 7 == 0 ? 7 : (S % 6 == 0 ? 6 : (S % 5 == 0 ? 5 : 4)))); }

// The first two weight chunks of a wave's share of a GEMM, requested EARLY: issue() before the LayerNorm pass / the
// staging barrier that precedes the product, so that their round trip overlaps it instead of opening the GEMM (four
// products per launch each started on an idle ~1 us wait).
template <int N, int K, int SF = K / 32, int CHO = 0, int NWV = NW>          // (NWV: the workgroup's waves -- 8 in the 64-row form)
struct WPre {
    static constexpr int S = K / 32, CH = CHO ? CHO : chunk_steps(S), CPG = S / CH, G = N / 16;     // (CHO: the 64-row form's shorter chunks)
    uint4 b0[CH], b1[CH];
    static __device__ __forceinline__ int nchunks() {
        const int wave = threadIdx.x >> 6;
        return (wave < G ? (G - wave + NWV - 1) / NWV : 0) * CPG;
    }
    // (g0, s0): the product multiplies a SLICE of the packed weight -- column groups [g0, g0 + G) and k-steps [s0, s0 + S) of
    // the SF steps a group has (the cluster kernels: a workgroup owns a column range or a K range of the layer's weight)
    static __device__ __forceinline__ void load(const uint16_t* __restrict__ W, uint4 (&b)[CH], int c, int g0 = 0, int s0 = 0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int g = g0 + wave + (c / CPG) * NWV, st = s0 + (c % CPG) * CH;   // W is PACKED: one contiguous KB per (group, k-step)
        const uint16_t* wp = W + ((int64_t)(g * SF + st) * 64 + lane) * 8;
#pragma unroll
        for (int s = 0; s < CH; ++s) b[s] = *reinterpret_cast<const uint4*>(wp + 512 * s);
    }
    __device__ __forceinline__ void issue(const uint16_t* __restrict__ W, int g0 = 0, int s0 = 0) {
        const int n = nchunks();
        if (n > 0) load(W, b0, 0, g0, s0);
        if (n > 1) load(W, b1, 1, g0, s0);
    }
};

template <int BM, int N, int K, int LDA, typename EPI, int SF = K / 32, int CHO = 0, int NWV = NW>
__device__ __forceinline__ void wg_gemm(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, EPI&& epi, WPre<N, K, SF, CHO, NWV>& pre,
                                        int g0 = 0, int s0 = 0) {
    constexpr int MT = BM / 16;
    constexpr int S = K / 32;                        // k-steps per group
    constexpr int CH = WPre<N, K, SF, CHO, NWV>::CH;      // k-steps per chunk
    constexpr int CPG = S / CH;                      // chunks per group
    constexpr bool AREG = S <= 8 && MT <= 2;         // A operands held in registers (four row tiles x 8 steps would be 128 of them)
    static_assert(K % 32 == 0 && S % CH == 0 && N % 16 == 0, "shape");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    constexpr int G = N / 16;
    const int ng = wave < G ? (G - wave + NWV - 1) / NWV : 0;       // this wave's groups
    const int nchunks = ng * CPG;
    if (nchunks == 0) return;
    const uint16_t* a0 = A + j * LDA + 8 * q;
    uint4 afr[AREG ? MT : 1][AREG ? S : 1];
    if constexpr (AREG) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int s = 0; s < S; ++s) afr[t][s] = *reinterpret_cast<const uint4*>(a0 + 16 * t * LDA + 32 * s);
    }
    auto load = [&](uint4 (&b)[CH], int c) { WPre<N, K, SF, CHO, NWV>::load(W, b, c, g0, s0); };
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const uint4 (&b)[CH], int c) {
        const int sc = (c % CPG) * CH;
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const bf16x8 bf = __builtin_bit_cast(bf16x8, b[s]);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                bf16x8 af;
                if constexpr (AREG) af = __builtin_bit_cast(bf16x8, afr[t][s]);
                else af = *reinterpret_cast<const bf16x8*>(a0 + 16 * t * LDA + 32 * (sc + s));
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[t], 0, 0, 0);
            }
        }
        if (c % CPG == CPG - 1) {
            epi(wave + (c / CPG) * NWV, acc);         // (the group's index INSIDE the slice)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    uint4 (&b0)[CH] = pre.b0;
    uint4 (&b1)[CH] = pre.b1;                        // chunks 0 and 1: requested by pre.issue()
    uint4 b2[CH];
    for (int c = 0; c < nchunks; c += 3) {
        if (c + 2 < nchunks) load(b2, c + 2);
        compute(b0, c);
        if (c + 1 >= nchunks) break;
        if (c + 3 < nchunks) load(b0, c + 3);
        compute(b1, c + 1);
        if (c + 2 >= nchunks) break;
        if (c + 4 < nchunks) load(b1, c + 4);
        compute(b2, c + 2);
    }
}

// xb[r][:] (f32, LDS) -> LayerNorm -> `dst_a` (bf16, LDS, the next GEMM's A operand) + global copies; also writes the
// pre-norm rows (x1 / x2) and the statistics.  One wave per row (rows w, w + NW, ...), 256-byte runs to global; two-pass
// mean / variance as mobgt_dropout_add_ln_fwd.
// a LayerNorm's weight and bias at this lane's columns, requested at kernel start (a global load inside the norm pass is one
// more exposed round trip per pass)
template <int C>
struct LnW {
    static constexpr int PER = (C + 63) / 64;
    float w[PER], b[PER];
    __device__ __forceinline__ void issue(const float* __restrict__ wp, const float* __restrict__ bp) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = min(lane + 64 * k, C - 1);
            w[k] = wp[c];
            b[k] = bp ? bp[c] : 0.f;
        }
    }
};

template <int BM, int C, int LDX, int LDA, int NWV = NW>
__device__ __forceinline__ void ln_rows(const float* __restrict__ xb, uint16_t* __restrict__ dst_a, const LnW<C>& lw,
                                        float* __restrict__ g_pre, uint16_t* __restrict__ g_bf,
                                        float* __restrict__ g_f32, float* __restrict__ g_mean, float* __restrict__ g_rstd,
                                        int r0, int R, int own_mod = 1, int own_rem = 0) {
    // (own_mod, own_rem): the cluster kernels run the norm on every member of a cluster (each needs all rows in LDS) and
    // member `own_rem` of `own_mod` writes the global copies of rows r = own_rem (mod own_mod)
    constexpr int PER = (C + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float (&wv)[PER] = lw.w;
    const float (&bv)[PER] = lw.b;
    for (int r = wave; r < BM; r += NWV) {
        const int64_t row = r0 + r;
        float v[PER], s = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = lane + 64 * k;
            v[k] = c < C ? xb[r * LDX + c] : 0.f;
            s += v[k];
        }
        const float mu = wave64_sum(s) * (1.f / C);
        float qq = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float d = lane + 64 * k < C ? v[k] - mu : 0.f;
            qq += d * d;
        }
        const float rs = rsqrtf(wave64_sum(qq) * (1.f / C) + 1e-5f);
        const bool on = row < R && (own_mod == 1 || r % own_mod == own_rem);
        if (on && lane == 0) { g_mean[row] = mu; g_rstd[row] = rs; }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                const float o = (v[k] - mu) * rs * wv[k] + bv[k];
                const uint16_t ob = bf16_bits(o);
                dst_a[r * LDA + c] = ob;
                if (on) {
                    if (g_pre) g_pre[row * C + c] = v[k];
                    g_bf[row * C + c] = ob;
                    if (g_f32) g_f32[row * C + c] = o;
                }
            }
        }
    }
}

// ---- the same two passes with HALF a wave per row (the cluster kernels): 16 rows = 8 waves x 2 halves, ONE pass where a wave
// per row takes two (rows 12..15 on waves 0..3: 0.9 us per norm).  A lane holds columns hl + 32 k; the sums are formed in the
// very order of ln_rows -- the even k are what lane hl of a whole wave holds, the odd k what lane hl + 32 holds; 16-lane DPP
// sums of both, then (row 1 + row 0) and (row 3 + row 2), then their sum -- so mean / rstd / z are bit-identical to it.
__device__ __forceinline__ float swz_xor16(float v) {      // the value of lane ^ 16 (ds_swizzle, bit-mask mode: and 0x1f, xor 0x10)
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ float half_sum_as_wave(float pa, float pb) {
    pa = row16_sum(pa);
    pb = row16_sum(pb);
    pa += swz_xor16(pa);                     // r1 + r0
    pb += swz_xor16(pb);                     // r3 + r2
    return pb + pa;
}
template <int C>
struct LnWH {
    static constexpr int PER = C / 32;
    float w[PER], b[PER];
    __device__ __forceinline__ void issue(const float* __restrict__ wp, const float* __restrict__ bp) {
        const int hl = threadIdx.x & 31;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            w[k] = wp[hl + 32 * k];
            b[k] = bp ? bp[hl + 32 * k] : 0.f;
        }
    }
};
template <int BM, int C, int LDX, int LDA>
__device__ __forceinline__ void ln_rows_hw(const float* __restrict__ xb, uint16_t* __restrict__ dst_a, const LnWH<C>& lw,
                                           float* __restrict__ g_pre, uint16_t* __restrict__ g_bf,
                                           float* __restrict__ g_f32, float* __restrict__ g_mean, float* __restrict__ g_rstd,
                                           int r0, int R, int own_mod, int own_rem) {
    static_assert(C % 64 == 0 && BM <= 2 * NW, "half a wave per row");
    constexpr int PER = C / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31;
    const int r = 2 * wave + (lane >> 5);
    if (r >= BM) return;
    const int64_t row = r0 + r;
    float v[PER], pa = 0.f, pb = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        v[k] = xb[r * LDX + hl + 32 * k];
        if (k & 1) pb += v[k]; else pa += v[k];
    }
    const float mu = half_sum_as_wave(pa, pb) * (1.f / C);
    pa = pb = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const float d = v[k] - mu;
        if (k & 1) pb += d * d; else pa += d * d;
    }
    const float rs = rsqrtf(half_sum_as_wave(pa, pb) * (1.f / C) + 1e-5f);
    const bool on = row < R && (own_mod == 1 || r % own_mod == own_rem);
    if (on && hl == 0) { g_mean[row] = mu; g_rstd[row] = rs; }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = hl + 32 * k;
        const float o = (v[k] - mu) * rs * lw.w[k] + lw.b[k];
        const uint16_t ob = bf16_bits(o);
        dst_a[r * LDA + c] = ob;
        if (on) {
            if (g_pre) g_pre[row * C + c] = v[k];
            g_bf[row * C + c] = ob;
            if (g_f32) g_f32[row * C + c] = o;
        }
    }
}

// rows of the f32 residual tile [BM][LDX] -> global [R][C] (pre-LN layers without a successor in the chain: x2 leaves as it is)
template <int BM, int C, int LDX, int NTV = NT>
__device__ __forceinline__ void store_f32_rows(const float* __restrict__ xb, float* __restrict__ dst, int r0, int R, int own_mod = 1,
                                               int own_rem = 0) {
    for (int e = threadIdx.x; e < BM * (C / 4); e += NTV) {
        const int r = e / (C / 4), c = (e % (C / 4)) * 4;
        if (r0 + r < R && (own_mod == 1 || r % own_mod == own_rem))
            *reinterpret_cast<float4*>(dst + (int64_t)(r0 + r) * C + c) = *reinterpret_cast<const float4*>(xb + r * LDX + c);
    }
}

// rows of an LDS bf16 tile [BM][LD] -> global [R][N], 16 bytes per thread
template <int BM, int N, int LD, int NTV = NT>
__device__ __forceinline__ void store_rows(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int r0, int R) {
    for (int e = threadIdx.x; e < BM * (N / 8); e += NTV) {
        const int r = e / (N / 8), c = (e % (N / 8)) * 8;
        if (r0 + r < R) *reinterpret_cast<uint4*>(dst + (int64_t)(r0 + r) * N + c) = *reinterpret_cast<const uint4*>(src + r * LD + c);
    }
}

template <int BM, int C, int F>
__global__ __launch_bounds__(NT) void layer_chain_fwd_kernel(const ChainParams p) {
    constexpr int LDA = C + 8, LDH = F + 8, LDX = C + 4, MT = BM / 16;
    static_assert(3 * C + 8 <= LDH, "the qkv tile reuses the u tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* xb = reinterpret_cast<float*>(smem_raw);                              // [BM][LDX] f32: x -> x1 -> x2
    uint16_t* ab = reinterpret_cast<uint16_t*>(xb + BM * LDX);                   // [BM][LDA] bf16: a -> z -> out_a
    uint16_t* hb = ab + BM * LDA;                                                // [BM][LDH] bf16: h
    uint16_t* ub = hb + BM * LDH;                                                // [BM][LDH] bf16: u, later the next layer's qkv
    const int r0 = blockIdx.x * BM;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    STAMP_DECL;
    STAMP(0);

    WPre<C, C> pre1;
    pre1.issue(p.wo);
    LnW<C> ln1, ln2;
    ln1.issue(p.n1w, p.n1b);
    if (p.nxw) ln2.issue(p.nxw, p.nxb);           // (null: a pre-LN layer whose successor is not part of this chain -- x2 is the output)
    // this block's rows of a (bf16) and x (f32) -> LDS; rows past R are clamped (their results are never stored)
    for (int e = threadIdx.x; e < BM * (C / 8); e += NT) {
        const int r = e / (C / 8), c = (e % (C / 8)) * 8;
        *reinterpret_cast<uint4*>(ab + r * LDA + c) = *reinterpret_cast<const uint4*>(p.a + (int64_t)min(r0 + r, p.R - 1) * C + c);
    }
    for (int e = threadIdx.x; e < BM * (C / 4); e += NT) {
        const int r = e / (C / 4), c = (e % (C / 4)) * 4;
        *reinterpret_cast<float4*>(xb + r * LDX + c) = *reinterpret_cast<const float4*>(p.x + (int64_t)min(r0 + r, p.R - 1) * C + c);
    }
    __syncthreads();
    STAMP(1);

    // ---- y = a Wo^T + bo;  x1 = x + dropout(y)
    WPre<F, C> pre2;
    wg_gemm<BM, C, C, LDA>(ab, p.wo, [&](int g, const f32x4 (&acc)[MT]) {
        const int col = 16 * g + j;
        const float bias = bf16_val(p.bo[col]);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = 16 * t + 4 * q + v;
                float y = bf16_round(acc[t][v] + bias);
                if (p.thr) {
                    const uint32_t rowh = dropout_row_hash(seed, (uint32_t)(r0 + r) ^ p.salt1);
                    y = dropout_bits16(seed, rowh, (uint32_t)col) >= p.thr ? y * p.inv_keep : 0.f;
                }
                xb[r * LDX + col] += y;
            }
    }, pre1);
    pre2.issue(p.w1);                 // (BEFORE the barrier: measured better than after it -- the request travels while the
    __syncthreads();                  // wave waits for the slower waves' epilogues)
    STAMP(2);
    // ---- z = ffn_norm1(x1)
    ln_rows<BM, C, LDX, LDA>(xb, ab, ln1, p.x1, p.z, nullptr, p.mean1, p.rstd1, r0, p.R);
    __syncthreads();
    STAMP(3);
    // ---- u = z W1^T + b1;  h = gelu(u)  (h from the ROUNDED pre-activation, as the separate launches computed it)
    WPre<C, F> pre3;
    wg_gemm<BM, F, C, LDA>(ab, p.w1, [&](int g, const f32x4 (&acc)[MT]) {
        const int col = 16 * g + j;
        const float bias = bf16_val(p.b1[col]);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = 16 * t + 4 * q + v;
                const uint16_t ubits = bf16_bits(acc[t][v] + bias);
                ub[r * LDH + col] = ubits;
                hb[r * LDH + col] = bf16_bits(gelu_f(bf16_val(ubits)));
            }
    }, pre2);
    pre3.issue(p.w2);
    __syncthreads();
    STAMP(4);
    store_rows<BM, F, LDH>(ub, p.u, r0, p.R);
    store_rows<BM, F, LDH>(hb, p.h, r0, p.R);
    STAMP(5);
    // ---- f = h W2^T + b2;  x2 = x1 + dropout(f)
    WPre<3 * C, C> pre4;
    wg_gemm<BM, C, F, LDH>(hb, p.w2, [&](int g, const f32x4 (&acc)[MT]) {
        const int col = 16 * g + j;
        const float bias = bf16_val(p.b2[col]);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = 16 * t + 4 * q + v;
                float f = bf16_round(acc[t][v] + bias);
                if (p.thr) {
                    const uint32_t rowh = dropout_row_hash(seed, (uint32_t)(r0 + r) ^ p.salt2);
                    f = dropout_bits16(seed, rowh, (uint32_t)col) >= p.thr ? f * p.inv_keep : 0.f;
                }
                xb[r * LDX + col] += f;
            }
    }, pre3);
    if (p.wq) pre4.issue(p.wq);
    __syncthreads();
    STAMP(6);
    // ---- out = ffn_norm2(x2)  (f32 residual stream + the bf16 copy the next QKV GEMM multiplies)
    //      pre-LN (model.py:479-489): the norm is the NEXT layer's self_attention_norm, the residual stream stays x2 (p.out null)
    if (!p.nxw) {
        store_f32_rows<BM, C, LDX>(xb, p.x2, r0, p.R);
        return;
    }
    ln_rows<BM, C, LDX, LDA>(xb, ab, ln2, p.x2, p.out_a, p.out, p.mean2, p.rstd2, r0, p.R);
    if (!p.wq) return;
    __syncthreads();
    STAMP(7);
    // ---- the next layer's qkv = out Wqkv^T + bqkv
    wg_gemm<BM, 3 * C, C, LDA>(ab, p.wq, [&](int g, const f32x4 (&acc)[MT]) {
        const int col = 16 * g + j;
        const float bias = bf16_val(p.bq[col]);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) ub[(16 * t + 4 * q + v) * LDH + col] = bf16_bits(acc[t][v] + bias);
    }, pre4);
    __syncthreads();
    STAMP(8);
    store_rows<BM, 3 * C, LDH>(ub, p.qkv, r0, p.R);
    STAMP(9);
    STAMP_DUMP();
}


// ---- the LONG-batch form of the forward chain (round 5): 64 rows per workgroup ------------------------------------------------
// Past 4 096 rows a 16-row workgroup re-streams the layer's 1.5 MB of weights once per 16 rows (S-BIG: 785 workgroups, 1.2 GB of
// L2 -> L1 traffic per layer) and lost against the library's GEMMs (round 4: 9.53 -> 10.93 ms).  Here a workgroup owns 64 rows --
// a B operand feeds FOUR MFMAs, 197 workgroups at S-BIG's 12 560 rows -- and the FFN is walked in chunks of 384 hidden columns
// (24 column groups: two per wave) so that the 1 024-wide intermediate never exists in LDS as a whole:
//     u_c = z W1[c]^T + b1[c];  h_c = gelu(u_c)   -> LDS (the A operand of the next product) and, 16 bytes per thread, global
//     f  += h_c W2[:, c]^T                         split-K over the chunks, the partial sums stay in the wave's registers
// LDS: the bf16 operand tile [64][C + 8] (a -> z -> out_a), the f32 residual tile [64][C + 4] (x -> x1 -> x2) and ONE chunk tile
// [64][392] bf16 (u, then h in place -- GELU runs in the coalesced pass that copies the chunk to global): 147 KB at C = 256; the
// next layer's qkv tile overlays the last two.  Rounding points, dropout hashes and what is
// saved for the backward are those of the 16-row form; FFN-2's f32 sum is formed in three parts (rounded to bf16 at the same
// point).  The backward of such a layer runs the separate launches (its 64-row form is the step that is left).
// The 64-row form's product: out[64 x N] = A[64 x K] (LDS) x W^T (packed as for wg_gemm; slice (g0, s0) likewise).  A wave owns the
// column groups wave, wave + NWV, ... as in wg_gemm, but multiplies up to THREE of them at once: an A fragment read from LDS feeds
// NG MFMAs instead of one (with four row tiles per B fragment the one-group loop is a chain of LDS round trips at two waves per
// SIMD: measured ~150 clocks per MFMA) and the 4 x NG accumulators are independent.  The B operands run three chunks deep.
// epi(g, acc, bias) as wg_gemm's, once per group; `bias` (bf16 [N]) is requested with the first chunks -- a load inside the
// epilogue is a memory round trip per group with nothing beside it (measured: ~1 us each).
// A pass is an object: issue() requests its bias values and first two weight chunks, run() multiplies, finish() runs the
// epilogues.  The caller issues a product's FIRST pass ahead of whatever precedes the product (a barrier, a norm, the GELU
// pass), and a product of several passes issues pass k + 1 between run(k) and finish(k): no product and no pass opens on an
// idle memory round trip (measured ~1 us each, 13 of them per workgroup).
template <int BM, int NG, int K, int LDA, int SF, int NWV, int G>
struct WidePass {
    static constexpr int MT = BM / 16, S = K / 32, CH = S % 2 == 0 ? 2 : (S % 3 == 0 ? 3 : 1), NCH = S / CH;
    uint4 b0[CH][NG], b1[CH][NG];
    uint16_t bv[NG];
    f32x4 acc[NG][MT];
    static __device__ __forceinline__ void load(const uint16_t* __restrict__ W, uint4 (&bb)[CH][NG], const int c, const int g0,
                                                const int s0, const int i0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int g = g0 + min(wave + (i0 + i) * NWV, G - 1);         // (a wave without an i-th group multiplies the last one again)
            const uint16_t* wp = W + ((int64_t)(g * SF + s0 + c * CH) * 64 + lane) * 8;
#pragma unroll
            for (int s = 0; s < CH; ++s) bb[s][i] = *reinterpret_cast<const uint4*>(wp + 512 * s);
        }
    }
    __device__ __forceinline__ void issue(const uint16_t* __restrict__ W, const uint16_t* __restrict__ bias, const int g0, const int s0,
                                          const int i0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NG; ++i) bv[i] = bias[16 * min(wave + (i0 + i) * NWV, G - 1) + (lane & 15)];
        load(W, b0, 0, g0, s0, i0);
        if (NCH > 1) load(W, b1, 1, g0, s0, i0);
    }
    __device__ __forceinline__ void run(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const int g0, const int s0,
                                        const int i0) {
        const int lane = threadIdx.x & 63;
        const uint16_t* a0 = A + (lane & 15) * LDA + 8 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < NG; ++i)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        uint4 b2[CH][NG];
        auto compute = [&](const uint4 (&bb)[CH][NG], const int c) {
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                bf16x8 af[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) af[t] = *reinterpret_cast<const bf16x8*>(a0 + 16 * t * LDA + 32 * (c * CH + s));
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, bb[s][i]);
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bf, acc[i][t], 0, 0, 0);
                }
            }
        };
        // (the scheduling barriers pin every chunk request in FRONT of the products of the chunk two ahead of it: left alone, the
        //  scheduler sinks the loads to their uses -- the kernel sits at its register limit -- and a wave then has two KB in
        //  flight instead of 3 x CH x NG: measured 19 bytes per clock and CU)
#pragma unroll 1
        for (int c = 0; c < NCH; c += 3) {
            if (c + 2 < NCH) load(W, b2, c + 2, g0, s0, i0);
            __builtin_amdgcn_sched_barrier(0);
            compute(b0, c);
            if (c + 1 >= NCH) break;
            __builtin_amdgcn_sched_barrier(0);
            if (c + 3 < NCH) load(W, b0, c + 3, g0, s0, i0);
            __builtin_amdgcn_sched_barrier(0);
            compute(b1, c + 1);
            if (c + 2 >= NCH) break;
            __builtin_amdgcn_sched_barrier(0);
            if (c + 4 < NCH) load(W, b1, c + 4, g0, s0, i0);
            __builtin_amdgcn_sched_barrier(0);
            compute(b2, c + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    template <typename EPI>
    __device__ __forceinline__ void finish(EPI&& epi, const int i0) {
        const int wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NG; ++i)
            if (wave + (i0 + i) * NWV < G) epi(wave + (i0 + i) * NWV, acc[i], bf16_val(bv[i]));
    }
};
// a product: passes of up to WIDE_NG groups per wave (PER groups per wave in all)
constexpr int WIDE_NG = 3;
template <int BM, int N, int K, int LDA, int SF, int NWV>
struct WideGemm {
    static constexpr int G = N / 16, PER = (G + NWV - 1) / NWV;
    static_assert(N % 16 == 0 && K % 32 == 0 && PER >= 1 && PER <= 3 * WIDE_NG, "shape");
    static constexpr int N0 = PER < WIDE_NG ? PER : WIDE_NG, R1 = PER - N0, N1 = R1 < WIDE_NG ? R1 : WIDE_NG, N2 = R1 - N1;
    WidePass<BM, N0, K, LDA, SF, NWV, G> p0;
    __device__ __forceinline__ void issue(const uint16_t* __restrict__ W, const uint16_t* __restrict__ bias, const int g0 = 0, const int s0 = 0) {
        p0.issue(W, bias, g0, s0, 0);
    }
    template <typename EPI>
    __device__ __forceinline__ void run(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const uint16_t* __restrict__ bias,
                                        EPI&& epi, const int g0 = 0, const int s0 = 0) {
        p0.run(A, W, g0, s0, 0);
        if constexpr (N1 > 0) {
            WidePass<BM, N1, K, LDA, SF, NWV, G> p1;
            p1.issue(W, bias, g0, s0, N0);
            p0.finish(epi, 0);
            p1.run(A, W, g0, s0, N0);
            if constexpr (N2 > 0) {
                WidePass<BM, N2, K, LDA, SF, NWV, G> p2;
                p2.issue(W, bias, g0, s0, N0 + N1);
                p1.finish(epi, N0);
                p2.run(A, W, g0, s0, N0 + N1);
                p2.finish(epi, N0 + N1);
            } else {
                p1.finish(epi, N0);
            }
        } else {
            p0.finish(epi, 0);
        }
    }
};

constexpr int BIG_NW = 8;
template <int C, int F>
__global__ __launch_bounds__(BIG_NW * 64) void layer_chain_fwd_big_kernel(const ChainParams p) {
    constexpr int NWB = BIG_NW, NTB = NWB * 64;              // 8 waves: 2 per SIMD, 256 VGPRs; 16 / 24 / 48 column groups deal evenly
    constexpr int BM = 64, MT = 4, LDA = C + 8, LDX = C + 4, FC = 384, LDC = FC + 8, LDQ = 3 * C + 8;
    static_assert(F > 2 * FC && F <= 3 * FC && (F - 2 * FC) % 32 == 0 && C % 64 == 0, "three FFN chunks");
    static_assert(BM * LDQ * 2 <= BM * LDX * 4 + BM * LDC * 2, "the qkv tile overlays the residual tile and the chunk tile");
    constexpr int FL = F - 2 * FC;                                               // the last chunk's width (256)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint16_t* ab = reinterpret_cast<uint16_t*>(smem_raw);                        // [BM][LDA] bf16: a -> z -> out_a
    float* xb = reinterpret_cast<float*>(ab + BM * LDA);                         // [BM][LDX] f32: x -> x1 -> x2
    uint16_t* hb = reinterpret_cast<uint16_t*>(xb + BM * LDX);                   // [BM][LDC] bf16: a chunk of u, then of h in place
    uint16_t* qb = reinterpret_cast<uint16_t*>(xb);                              // [BM][LDQ] bf16: the next layer's qkv (over xb | hb)
    const int r0 = blockIdx.x * BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    STAMP_DECL;
    STAMP(0);

    // the dropout masks' row hashes (two full-width multiplies each: 16 clocks apiece), once per row and residual add instead of
    // once per ELEMENT in the epilogues (64 rows x 16 columns per lane and group: measured 2.2 us per product)
    __shared__ uint32_t s_rowh[2][BM];
    if (threadIdx.x < 2 * BM)
        s_rowh[threadIdx.x / BM][threadIdx.x % BM] =
            dropout_row_hash(seed, (uint32_t)(r0 + (int)(threadIdx.x % BM)) ^ (threadIdx.x < BM ? p.salt1 : p.salt2));
    WideGemm<BM, C, C, LDA, C / 32, NWB> g_wo;
    g_wo.issue(p.wo, p.bo);
    // the norms: HALF a wave per row (ln_rows_hw, bit-identical to the wave-per-row form), 16 rows per pass and four passes whose
    // reductions the compiler interleaves -- a wave per row walked 8 rows one after the other: 4.4 us per norm
    LnWH<C> ln1;
    ln1.issue(p.n1w, p.n1b);
    auto norm_rows = [&](const LnWH<C>& lw, float* g_pre, uint16_t* g_bf, float* g_f32, float* g_mean, float* g_rstd) {
#pragma unroll
        for (int it = 0; it < BM / 16; ++it)
            ln_rows_hw<16, C, LDX, LDA>(xb + 16 * it * LDX, ab + 16 * it * LDA, lw, g_pre, g_bf, g_f32, g_mean, g_rstd, r0 + 16 * it, p.R, 1, 0);
    };
#pragma unroll
    for (int e = threadIdx.x; e < BM * (C / 8); e += NTB) {
        const int r = e / (C / 8), c = (e % (C / 8)) * 8;
        *reinterpret_cast<uint4*>(ab + r * LDA + c) = *reinterpret_cast<const uint4*>(p.a + (int64_t)min(r0 + r, p.R - 1) * C + c);
    }
#pragma unroll
    for (int e = threadIdx.x; e < BM * (C / 4); e += NTB) {
        const int r = e / (C / 4), c = (e % (C / 4)) * 4;
        *reinterpret_cast<float4*>(xb + r * LDX + c) = *reinterpret_cast<const float4*>(p.x + (int64_t)min(r0 + r, p.R - 1) * C + c);
    }
    __syncthreads();
    STAMP(1);

    // the residual tile += dropout(bf16(acc + bias)) at this lane's 4 x MT positions of column group g
    // (the tile's 16 values are READ first and written last: sixteen read-modify-writes of LDS addresses the compiler cannot tell
    //  apart run one after the other -- measured 2.5 us per product)
    auto residual_add = [&](const int g, const f32x4 (&acc)[MT], const float bias, const int which) {
        const int col = 16 * g + j;
        float xv[MT][4];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) xv[t][v] = xb[(16 * t + 4 * q + v) * LDX + col];
        if (p.thr) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = 16 * t + 4 * q + v;
                    const float y = bf16_round(acc[t][v] + bias);
                    xv[t][v] += dropout_bits16(seed, s_rowh[which][r], (uint32_t)col) >= p.thr ? y * p.inv_keep : 0.f;
                }
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) xv[t][v] += bf16_round(acc[t][v] + bias);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) xb[(16 * t + 4 * q + v) * LDX + col] = xv[t][v];
    };
    // ---- y = a Wo^T + bo;  x1 = x + dropout(y)
    g_wo.run(ab, p.wo, p.bo, [&](int g, const f32x4 (&acc)[MT], float bias) { residual_add(g, acc, bias, 0); });
    WideGemm<BM, FC, C, LDA, C / 32, NWB> g_w1a;
    g_w1a.issue(p.w1, p.b1, 0, 0);                      // (ahead of the barrier and the norm)
    __syncthreads();
    STAMP(2);
    // ---- z = ffn_norm1(x1)
    norm_rows(ln1, p.x1, p.z, nullptr, p.mean1, p.rstd1);
    __syncthreads();
    STAMP(3);

    // ---- the FFN in chunks.  f's partial sums (split-K over the chunks) live in the residual tile: x1 has left for global with
    //      the norm's pass and is read back for the second residual add (this workgroup wrote it; 64 x C f32 through the L2).
    //      (In registers -- 32 per lane across the three chunks -- they were spilled to scratch: 6 us for that add alone.)
    uint16_t b2v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b2v[i] = p.b2[16 * min(wave + NWB * i, C / 16 - 1) + j];
    // one chunk: `g1` (issued by the caller) multiplies z by W1's rows c0 .., the GELU pass, then h_c by W2's columns c0 ..
    auto ffn_chunk = [&](auto nc_c, const int c0, auto& g1) {
        constexpr int NC = decltype(nc_c)::value;
        g1.run(ab, p.w1, p.b1 + c0, [&](int g, const f32x4 (&acc)[MT], float bias) {
            const int col = 16 * g + j;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) hb[(16 * t + 4 * q + v) * LDC + col] = bf16_bits(acc[t][v] + bias);
        }, c0 / 16, 0);
        WideGemm<BM, C, NC, LDC, F / 32, NWB> g2;
        g2.issue(p.w2, p.b2 /* (ignored: added with x2) */, 0, c0 / 32);
        __syncthreads();
        if (c0 == 0) STAMP(4);
        // u (rounded) -> global; h = gelu(u) -> global and, IN PLACE, the tile: 16 bytes per thread and step (GELU in this pass
        // and not in the product's epilogue: all waves meet the vector pipe once, on coalesced 16-byte pieces)
        for (int e = threadIdx.x; e < BM * (NC / 8); e += NTB) {
            const int r = e / (NC / 8), c = (e % (NC / 8)) * 8;
            uint4* tp = reinterpret_cast<uint4*>(hb + r * LDC + c);
            const uint4 uv = *tp;
            const uint32_t w[4] = {uv.x, uv.y, uv.z, uv.w};
            uint32_t hw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t lo = bf16_bits(gelu_f(__builtin_bit_cast(float, w[i] << 16)));
                const uint32_t hi = bf16_bits(gelu_f(__builtin_bit_cast(float, w[i] & 0xffff0000u)));
                hw[i] = lo | (hi << 16);
            }
            const uint4 hv = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            *tp = hv;
            if (r0 + r < p.R) {
                *reinterpret_cast<uint4*>(p.u + (int64_t)(r0 + r) * F + c0 + c) = uv;
                *reinterpret_cast<uint4*>(p.h + (int64_t)(r0 + r) * F + c0 + c) = hv;
            }
        }
        __syncthreads();
        if (c0 == 0) STAMP(5);
        g2.run(hb, p.w2, p.b2, [&](int g, const f32x4 (&acc)[MT], float) {
            const int col = 16 * g + j;
            float fv[MT][4];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) fv[t][v] = c0 ? xb[(16 * t + 4 * q + v) * LDX + col] : 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) xb[(16 * t + 4 * q + v) * LDX + col] = fv[t][v] + acc[t][v];
        }, 0, c0 / 32);
    };
    ffn_chunk(std::integral_constant<int, FC>{}, 0, g_w1a);
    WideGemm<BM, FC, C, LDA, C / 32, NWB> g_w1b;
    g_w1b.issue(p.w1, p.b1 + FC, FC / 16, 0);
    __syncthreads();                                     // (the chunk's h has been multiplied: its tile may be overwritten)
    STAMP(7);
    ffn_chunk(std::integral_constant<int, FC>{}, FC, g_w1b);
    WideGemm<BM, FL, C, LDA, C / 32, NWB> g_w1c;
    g_w1c.issue(p.w1, p.b1 + 2 * FC, 2 * FC / 16, 0);
    __syncthreads();
    STAMP(8);
    ffn_chunk(std::integral_constant<int, FL>{}, 2 * FC, g_w1c);
    WideGemm<BM, 3 * C, C, LDA, C / 32, NWB> g_wq;
    if (p.wq) g_wq.issue(p.wq, p.bq);                   // (ahead of the second residual add and norm)
    LnWH<C> ln2;
    if (p.nxw) ln2.issue(p.nxw, p.nxb);
    __syncthreads();
    STAMP(9);
    // ---- x2 = x1 + dropout(f + b2)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int g = wave + NWB * i;
        if (g < C / 16) {
            const int col = 16 * g + j;
            const float bias = bf16_val(b2v[i]);
            float x1v[MT][4], fv[MT][4];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = 16 * t + 4 * q + v;
                    x1v[t][v] = p.x1[(int64_t)min(r0 + r, p.R - 1) * C + col];
                    fv[t][v] = xb[r * LDX + col];
                }
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = 16 * t + 4 * q + v;
                    float f = bf16_round(fv[t][v] + bias);
                    if (p.thr) f = dropout_bits16(seed, s_rowh[1][r], (uint32_t)col) >= p.thr ? f * p.inv_keep : 0.f;
                    xb[r * LDX + col] = x1v[t][v] + f;
                }
        }
    }
    __syncthreads();
    STAMP(10);
    if (!p.nxw) {
        store_f32_rows<BM, C, LDX, NTB>(xb, p.x2, r0, p.R);
        return;
    }
    norm_rows(ln2, p.x2, p.out_a, p.out, p.mean2, p.rstd2);
    if (!p.wq) return;
    __syncthreads();
    STAMP(11);
    // ---- the next layer's qkv = out Wqkv^T + bqkv
    g_wq.run(ab, p.wq, p.bq, [&](int g, const f32x4 (&acc)[MT], float bias) {
        const int col = 16 * g + j;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) qb[(16 * t + 4 * q + v) * LDQ + col] = bf16_bits(acc[t][v] + bias);
    });
    __syncthreads();
    STAMP(12);
    store_rows<BM, 3 * C, LDQ, NTB>(qb, p.qkv, r0, p.R);
    STAMP(13);
    STAMP_DUMP();
}

constexpr int64_t CHAIN_BIG_ROWS = 4096;         // rows past which the forward chain takes the 64-row form (the host packs for it: model.pack_layer_weights)
template <int C, int F>
int launch_big(const ChainParams& p, hipStream_t st) {
    constexpr int BM = 64, LDA = C + 8, LDX = C + 4, LDC = 384 + 8;
    constexpr size_t lds = BM * LDA * 2 + BM * LDX * 4 + BM * LDC * 2;
    static_assert(lds <= 152 * 1024, "LDS plan");
    int rc = (int)hipFuncSetAttribute((const void*)layer_chain_fwd_big_kernel<C, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    hipLaunchKernelGGL((layer_chain_fwd_big_kernel<C, F>), dim3((p.R + BM - 1) / BM), dim3(BIG_NW * 64), lds, st, p);
    return (int)hipGetLastError();
}

// ---- the CLUSTER form of the two chain kernels (short batches) -------------------------------------------------------------
// With 16 rows per workgroup a batch of R rows occupies ceil(R / 16) compute units -- 38 of 256 at S-FSQ's R ~ 600 -- and
// each of them streams the layer's whole 1.08 MB of weights through its own L1 (~8 us, the kernel's floor).  Here a row
// block belongs to a CLUSTER of NCL workgroups (2 or 4) that split the two big weights: member m owns the column range m of
// FFN-1 (so of u and h) and of the next QKV projection, and the K range m of FFN-2 (the columns of h it produced).  The small
// out-projection (73 KB) is computed by every member in full -- cheaper than handing x1 slices around.  Per workgroup a
// quarter of the weight stream; per launch 4x the compute units.  FFN-2's partial sums need adding, so the members meet ONCE
// per pass:
//     forward    FFN-2 partial sums (split-K over the members) -> every member adds them in member order, forms x2, norms it
//     backward   du W1 partial sums -> dz      (the layer above's tail product dqkv Wqkv is computed by every member in full)
// The hand-over carries its own flags ("LL": each f32 travels as an 8-byte word {value, tag}, written with ONE sc1 --
// agent-scope, write-through -- 8-byte store and polled with sc1 loads until the tag is the launch's): no store drain, no
// barrier, no counter, one memory round trip.  tag = gen[block] + 1, where gen[block] is read by every member when it
// starts and raised by member 0 once it holds everybody's words (so all members have read it; the next launch -- which
// cannot start before this one has ended -- sees the new value and never mistakes stale words for its own).  Agent-scope
// fences would write back / invalidate the XCD's whole L2 (see csrc/smallgcn.hip), and a counter hand-over costs three
// round trips (measured 2.4-3.2 us per hand-over against ~1.3 here).  Members of a cluster sit on ONE XCD (block ids that
// agree modulo 8; the sc1 accesses are correct on any placement).
// Co-residency: the host picks NCL so that the live workgroups fit the device's compute units at one per unit; the poll is
// bounded and GIVES UP into the fault word (WS_FAULT below; the host polls it: train.TrainStep.check_faults, called by
// EpochLoop.run_epoch, bench.py and the tests' data-parallel worker).  Rounding: as the one-workgroup form except FFN-2 / du W1, whose f32 sums are added in NCL parts (then
// rounded to bf16 at the same point).
constexpr int WS_GEN_INTS = 1024;                                       // gen[<= 256 row blocks] (+ spare)
// A member that does not see its partners' packets within the poll limit GIVES UP instead of trapping (a trap kills the
// process -- under data parallelism the rank, and with it the job; VERDICT r3 weak #7b): it counts itself in the
// workspace's fault word and carries on with what it has, so the launch ends, its results are garbage, and the host finds
// the count at its next check (`mobgt_chain_faults`; train.TrainStep.check_faults re-runs the step in the one-workgroup
// form).  Never seen in an undisturbed run: all members of a cluster are resident together by construction.
constexpr int WS_FAULT = WS_GEN_INTS - 1;                               // workgroups that gave up, since the last reset
constexpr int WS_LIMIT = WS_GEN_INTS - 2;                               // poll rounds before giving up; 0 = the default below
constexpr uint32_t WS_DEFAULT_ROUNDS = 1u << 21;                        // seconds
constexpr size_t WS_LL_WORDS = (size_t)256 * 16 * 256;                  // <= 256 live workgroups x 16 rows x C <= 256 words of 8 bytes

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifndef LL_STORE_POLICY
#define LL_STORE_POLICY LL_SC1
#endif
constexpr int LL_SC0 = 1, LL_SC1 = 16, LL_VOLATILE = (int)0x80000000;       // buffer-instruction cache policy bits (gfx940+)

// The exchange area of one cluster: NCL slots of BM x C / 2 packets of 16 bytes {v0, tag, v1, tag} -- two f32 of adjacent
// columns, each with the launch's tag beside it in the same naturally aligned 8 bytes (what a store makes visible together).
// 16-byte accesses: the load unit takes 16 clocks per wave instruction whatever its width, and 8-byte polls (24 KB of
// payload per member as 96 KB of words) were 2.2 us of pure issue.
// cluster_put: this member's partial sums pb [BM][LDX] -> its slot, sc1 (agent scope: written through).
// cluster_get: epi(k, r, c, sums of columns c and c + 1 over the members in member order) (own partial sums from LDS: the same bits for everybody).  A
// poll round requests EVERY packet still missing before it looks at any (one round trip per round; polled one by one,
// twelve missing packets cost twelve round trips: measured 8.9 us).  The first rounds read through the XCD's L2 (sc0: misses
// only the CU's L1) -- the members of a cluster share an L2 when block ids map to XCDs round-robin (MI300-class parts in SPX
// mode); on any other placement such a load may keep returning a stale line, so later rounds use sc1.
template <int BM, int C, int NCL>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cluster_area(uint64_t* ws_ll, int blk) {
    return __builtin_amdgcn_make_buffer_rsrc(ws_ll + (int64_t)blk * NCL * BM * C, 0, NCL * BM * C * 8, 0x00020000);
}
template <int BM, int C, int LDX>
__device__ __forceinline__ void cluster_put(__amdgpu_buffer_rsrc_t area, int m, const float* __restrict__ pb, uint32_t tag) {
    constexpr int NP = BM * C / 2, EPT = (NP + NT - 1) / NT;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = threadIdx.x + k * NT;
        if (e < NP) {
            const int r = e / (C / 2), c = (e % (C / 2)) * 2;
            const u32x4 v = {__float_as_uint(pb[r * LDX + c]), tag, __float_as_uint(pb[r * LDX + c + 1]), tag};
            __builtin_amdgcn_raw_buffer_store_b128(v, area, (m * NP + e) * 16, 0, LL_STORE_POLICY);
        }
    }
}
template <int BM, int C, int LDX, int NCL, typename EPI>
__device__ __forceinline__ void cluster_get(__amdgpu_buffer_rsrc_t area, int m, const float* __restrict__ pb, uint32_t tag, uint32_t* ws_gen,
                                            EPI&& epi, int* dbg_rounds = nullptr) {
    constexpr int NP = BM * C / 2, EPT = (NP + NT - 1) / NT;
#ifndef LL_NEAR_ROUNDS
#define LL_NEAR_ROUNDS 24
#endif
    constexpr int NEAR_ROUNDS = LL_NEAR_ROUNDS;
    u32x4 w[EPT][NCL];
#pragma unroll
    for (int k = 0; k < EPT; ++k)
#pragma unroll
        for (int mm = 0; mm < NCL; ++mm) w[k][mm] = u32x4{0u, 0u, 0u, 0u};         // (tag 0 is never a launch's)
    int rounds = 0;
    bool missing;
    const uint32_t lim_raw = __builtin_nontemporal_load(ws_gen + WS_LIMIT);          // (host-written, launch-invariant)
    const int limit = (int)(lim_raw ? lim_raw : WS_DEFAULT_ROUNDS);
    do {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = min(threadIdx.x + k * NT, NP - 1);
#pragma unroll
            for (int mm = 0; mm < NCL; ++mm)
                if (mm != m && (w[k][mm][1] != tag || w[k][mm][3] != tag)) {
                    if (rounds < NEAR_ROUNDS) w[k][mm] = __builtin_amdgcn_raw_buffer_load_b128(area, (mm * NP + e) * 16, 0, LL_SC0 | LL_VOLATILE);
                    else w[k][mm] = __builtin_amdgcn_raw_buffer_load_b128(area, (mm * NP + e) * 16, 0, LL_SC1 | LL_VOLATILE);
                }
        }
        asm volatile("" ::: "memory");
        missing = false;
#pragma unroll
        for (int k = 0; k < EPT; ++k)
#pragma unroll
            for (int mm = 0; mm < NCL; ++mm) missing |= mm != m && (w[k][mm][1] != tag || w[k][mm][3] != tag);
        if (++rounds > limit) {                                          // seconds: never in a sane run
            // (one count per lane still waiting: non-zero = fault; limit word 0xffffffff = the tests' fault injection)
            if (missing || lim_raw == 0xffffffffu) atomicAdd(ws_gen + WS_FAULT, 1u);
            break;
        }
    } while (missing);
    if (dbg_rounds) *dbg_rounds = rounds;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = threadIdx.x + k * NT;
        if (e < NP) {
            const int r = e / (C / 2), c = (e % (C / 2)) * 2;
            const float o0 = pb[r * LDX + c], o1 = pb[r * LDX + c + 1];
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int mm = 0; mm < NCL; ++mm) {
                s0 += mm == m ? o0 : __uint_as_float(w[k][mm][0]);
                s1 += mm == m ? o1 : __uint_as_float(w[k][mm][2]);
            }
            epi(k, r, c, s0, s1);                     // (k: which of this thread's packets -- indexes what it preloaded per packet)
        }
    }
}
// block id -> (row block, member): ids that agree modulo 8 share an XCD; a cluster = NCL consecutive slots of one XCD.  All 8
// XCDs take clusters (nx = 8).  (Round 3 measured, round 5 removed the switch: confining the 152 live workgroups of R = 608 to
// the 5 XCDs they need cuts the launch's fetches from 13.3 to 10.0 MB -- every XCD that takes part pulls the layer's weights
// into its own L2 once -- but the launch gets SLOWER: forward 11.8 -> 12.4 us, backward 15.0 -> 21.7 us, its weight-gradient
// passengers are left with 3 XCDs.)
template <int NCL>
__device__ __forceinline__ void cluster_ids(int bid, int nx, int& blk, int& m) {
    const int xcd = bid & 7, slot = bid >> 3;
    m = slot % NCL;
    blk = xcd < nx ? (slot / NCL) * nx + xcd : -1;
}
inline int cluster_xcds(int nblk, int ncl) {
    (void)nblk; (void)ncl;
    return 8;
}

// rows of an LDS bf16 tile [BM][LD] -> columns [c0, c0 + N) of global [R][LDG]
template <int BM, int N, int LD>
__device__ __forceinline__ void store_cols(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int ldg, int c0, int r0, int R) {
    for (int e = threadIdx.x; e < BM * (N / 8); e += NT) {
        const int r = e / (N / 8), c = (e % (N / 8)) * 8;
        if (r0 + r < R) *reinterpret_cast<uint4*>(dst + (int64_t)(r0 + r) * ldg + c0 + c) = *reinterpret_cast<const uint4*>(src + r * LD + c);
    }
}

template <int BM, int C, int F, int NCL>
__global__ __launch_bounds__(NT) void layer_chain_fwd_cl_kernel(const ChainParams p) {
    constexpr int FS = F / NCL, QS = 3 * C / NCL;
    constexpr int LDA = C + 8, LDHS = FS + 8, LDX = C + 4, MT = BM / 16;
    static_assert(BM == 16 && FS % 32 == 0 && QS % 16 == 0 && QS <= FS, "cluster split");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* xb = reinterpret_cast<float*>(smem_raw);                              // [BM][LDX] f32: x -> x1 -> x2
    float* pb = xb + BM * LDX;                                                   // [BM][LDX] f32: this member's FFN-2 partial sums
    uint16_t* ab = reinterpret_cast<uint16_t*>(pb + BM * LDX);                   // [BM][LDA] bf16: a -> z -> out_a
    uint16_t* hb = ab + BM * LDA;                                                // [BM][LDHS] bf16: this member's columns of h
    uint16_t* ub = hb + BM * LDHS;                                               // [BM][LDHS] bf16: ... of u, later of the next qkv
    int blk, m;
    cluster_ids<NCL>((int)blockIdx.x, p.nx, blk, m);
    const int r0 = blk * BM;
    if (blk < 0 || r0 >= p.R) return;                                            // (a whole cluster: nobody waits for it)
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    STAMP_DECL;
    STAMP(0);

    // ---- everything read from global memory besides the big weights is requested NOW, in the order of its use: loads return
    // in issue order, so a late request (a bias inside a GEMM epilogue, the slow sc1 load of `gen`) is waited for together with
    // whatever was requested before it
    constexpr int AE = (BM * (C / 8) + NT - 1) / NT, XE = (BM * (C / 4) + NT - 1) / NT;
    uint4 a_in[AE];
    float4 x_in[XE];
#pragma unroll
    for (int k = 0; k < AE; ++k) {
        const int e = min(threadIdx.x + k * NT, BM * (C / 8) - 1), r = e / (C / 8), c = (e % (C / 8)) * 8;
        a_in[k] = *reinterpret_cast<const uint4*>(p.a + (int64_t)min(r0 + r, p.R - 1) * C + c);
    }
#pragma unroll
    for (int k = 0; k < XE; ++k) {
        const int e = min(threadIdx.x + k * NT, BM * (C / 4) - 1), r = e / (C / 4), c = (e % (C / 4)) * 4;
        x_in[k] = *reinterpret_cast<const float4*>(p.x + (int64_t)min(r0 + r, p.R - 1) * C + c);
    }
    WPre<C, C> pre1;
    pre1.issue(p.wo);
    // every weight chunk is requested TWO phases ahead of its product where the registers allow it (a phase is 0.8-1.5 us, a
    // miss in the XCD's L2 more), else one phase ahead
    constexpr bool DEEP = NCL == 4 && C <= 192;
    WPre<FS, C> pre2;
    if constexpr (DEEP) pre2.issue(p.w1, m * (FS / 16));
    LnWH<C> ln1, ln2;
    ln1.issue(p.n1w, p.n1b);
    if (p.nxw) ln2.issue(p.nxw, p.nxb);           // (null: see layer_chain_fwd_kernel)
    // biases at this lane's columns: group g of a product belongs to wave g % NW and is that wave's (g / NW)-th
    const int wave = threadIdx.x >> 6;
    constexpr int NG1 = (C / 16 + NW - 1) / NW, NG2 = (FS / 16 + NW - 1) / NW, NG4 = (QS / 16 + NW - 1) / NW;
    constexpr int PE = (BM * C / 2 + NT - 1) / NT;
    uint16_t bo_l[NG1], b1_l[NG2], bq_l[NG4];
    uint32_t b2_l[PE];
#pragma unroll
    for (int i = 0; i < NG1; ++i) bo_l[i] = p.bo[min(16 * (wave + i * NW), C - 16) + j];
#pragma unroll
    for (int i = 0; i < NG2; ++i) b1_l[i] = p.b1[m * FS + min(16 * (wave + i * NW), FS - 16) + j];
#pragma unroll
    for (int i = 0; i < NG4; ++i) bq_l[i] = p.wq ? p.bq[m * QS + min(16 * (wave + i * NW), QS - 16) + j] : (uint16_t)0;
#pragma unroll
    for (int k = 0; k < PE; ++k) b2_l[k] = *reinterpret_cast<const uint32_t*>(p.b2 + (min((int)threadIdx.x + k * NT, BM * C / 2 - 1) % (C / 2)) * 2);
    const uint32_t gen = __hip_atomic_load(p.ws_gen + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (used at the hand-over)
#pragma unroll
    for (int k = 0; k < AE; ++k) {
        const int e = threadIdx.x + k * NT, r = e / (C / 8), c = (e % (C / 8)) * 8;
        if (e < BM * (C / 8)) *reinterpret_cast<uint4*>(ab + r * LDA + c) = a_in[k];
    }
#pragma unroll
    for (int k = 0; k < XE; ++k) {
        const int e = threadIdx.x + k * NT, r = e / (C / 4), c = (e % (C / 4)) * 4;
        if (e < BM * (C / 4)) *reinterpret_cast<float4*>(xb + r * LDX + c) = x_in[k];
    }
    __syncthreads();
    STAMP(1);

    // ---- y = a Wo^T + bo;  x1 = x + dropout(y): in full on every member
    WPre<C, FS, F / 32> pre3;
    wg_gemm<BM, C, C, LDA>(ab, p.wo, [&](int g, const f32x4 (&acc)[MT]) {
        const int col = 16 * g + j;
        float bias = bf16_val(bo_l[0]);
#pragma unroll
        for (int i = 1; i < NG1; ++i) bias = g / NW == i ? bf16_val(bo_l[i]) : bias;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = 4 * q + v;
            float y = bf16_round(acc[0][v] + bias);
            if (p.thr) {
                const uint32_t rowh = dropout_row_hash(seed, (uint32_t)(r0 + r) ^ p.salt1);
                y = dropout_bits16(seed, rowh, (uint32_t)col) >= p.thr ? y * p.inv_keep : 0.f;
            }
            xb[r * LDX + col] += y;
        }
    }, pre1);
    if constexpr (DEEP) pre3.issue(p.w2, 0, m * (FS / 32));
    else pre2.issue(p.w1, m * (FS / 16));
    __syncthreads();
    STAMP(2);
    // ---- z = ffn_norm1(x1): every member norms all rows, member m writes rows m (mod NCL)
    ln_rows_hw<BM, C, LDX, LDA>(xb, ab, ln1, p.x1, p.z, nullptr, p.mean1, p.rstd1, r0, p.R, NCL, m);
    __syncthreads();
    STAMP(3);
    // ---- columns [m FS, (m + 1) FS) of u = z W1^T + b1;  h = gelu(u)
    WPre<QS, C> pre4;
    wg_gemm<BM, FS, C, LDA>(ab, p.w1, [&](int g, const f32x4 (&acc)[MT]) {
        const int cl = 16 * g + j;
        float bias = bf16_val(b1_l[0]);
#pragma unroll
        for (int i = 1; i < NG2; ++i) bias = g / NW == i ? bf16_val(b1_l[i]) : bias;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = 4 * q + v;
            const uint16_t ubits = bf16_bits(acc[0][v] + bias);
            ub[r * LDHS + cl] = ubits;
            hb[r * LDHS + cl] = bf16_bits(gelu_f(bf16_val(ubits)));
        }
    }, pre2, m * (FS / 16));
    if constexpr (DEEP) {
        if (p.wq) pre4.issue(p.wq, m * (QS / 16));
    } else {
        pre3.issue(p.w2, 0, m * (FS / 32));
    }
    __syncthreads();
    STAMP(4);
    // (u, h leave NOW: stores and loads retire in issue order, so a store in front of the hand-over's polls is waited for there)
    store_cols<BM, FS, LDHS>(ub, p.u, F, m * FS, r0, p.R);
    store_cols<BM, FS, LDHS>(hb, p.h, F, m * FS, r0, p.R);
    // ---- FFN-2 over this member's K range: partial sums of f = h W2^T
    if constexpr (!DEEP) {
        if (p.wq) pre4.issue(p.wq, m * (QS / 16));    // (HERE: requested behind this product, the hand-over's polls would wait for it)
    }
    wg_gemm<BM, C, FS, LDHS>(hb, p.w2, [&](int g, const f32x4 (&acc)[MT]) {
        const int col = 16 * g + j;
#pragma unroll
        for (int v = 0; v < 4; ++v) pb[(4 * q + v) * LDX + col] = acc[0][v];
    }, pre3, 0, m * (FS / 32));
    __syncthreads();
    STAMP(5);
    // ---- the partial sums change hands;  f = sum in member order + b2;  x2 = x1 + dropout(f)
    const __amdgpu_buffer_rsrc_t area = cluster_area<BM, C, NCL>(p.ws_ll, blk);
    cluster_put<BM, C, LDX>(area, m, pb, gen + 1u);
    STAMP(6);
#ifdef CH_DEBUG_ACK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_[13] = (int)wall_clock64();
#endif
    cluster_get<BM, C, LDX, NCL>(area, m, pb, gen + 1u, p.ws_gen, [&](int k, int r, int c, float s0, float s1) {
        float f0 = bf16_round(s0 + bf16_val((uint16_t)b2_l[k])), f1 = bf16_round(s1 + bf16_val((uint16_t)(b2_l[k] >> 16)));
        if (p.thr) {
            const uint32_t rowh = dropout_row_hash(seed, (uint32_t)(r0 + r) ^ p.salt2);
            f0 = dropout_bits16(seed, rowh, (uint32_t)c) >= p.thr ? f0 * p.inv_keep : 0.f;
            f1 = dropout_bits16(seed, rowh, (uint32_t)(c + 1)) >= p.thr ? f1 * p.inv_keep : 0.f;
        }
        xb[r * LDX + c] += f0;
        xb[r * LDX + c + 1] += f1;
    }
#ifdef CH_DEBUG
    , &st_[11]
#endif
    );
#ifdef CH_DEBUG
    st_[12] = (int)wall_clock64();
#endif
    __syncthreads();
    // (every thread of this workgroup holds every member's words: all of them have read gen)
    if (m == 0 && threadIdx.x == 0) __hip_atomic_store(p.ws_gen + blk, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    STAMP(7);
    // ---- out = ffn_norm2(x2)   (pre-LN: see layer_chain_fwd_kernel)
    if (!p.nxw) {
        store_f32_rows<BM, C, LDX>(xb, p.x2, r0, p.R, NCL, m);
        return;
    }
    ln_rows_hw<BM, C, LDX, LDA>(xb, ab, ln2, p.x2, p.out_a, p.out, p.mean2, p.rstd2, r0, p.R, NCL, m);
    if (!p.wq) return;
    __syncthreads();
    STAMP(8);
    // ---- columns [m QS, (m + 1) QS) of the next layer's qkv = out Wqkv^T + bqkv
    wg_gemm<BM, QS, C, LDA>(ab, p.wq, [&](int g, const f32x4 (&acc)[MT]) {
        const int cl = 16 * g + j;
        float bias = bf16_val(bq_l[0]);
#pragma unroll
        for (int i = 1; i < NG4; ++i) bias = g / NW == i ? bf16_val(bq_l[i]) : bias;
#pragma unroll
        for (int v = 0; v < 4; ++v) ub[(4 * q + v) * LDHS + cl] = bf16_bits(acc[0][v] + bias);
    }, pre4, m * (QS / 16));
    __syncthreads();
    STAMP(9);
    store_cols<BM, QS, LDHS>(ub, p.qkv, 3 * C, m * QS, r0, p.R);
    STAMP(10);
    STAMP_DUMP();
}

int device_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}
// members per row block: as many (4, 2) as keeps every live workgroup on a compute unit of its own; 1 = the one-workgroup form
int pick_ncl(int64_t R, const void* ws) {
    if (!ws) return 1;
    const int nblk = (int)((R + 15) / 16), cus = device_cus() < 256 ? device_cus() : 256;
    return nblk * 4 <= cus ? 4 : (nblk * 2 <= cus ? 2 : 1);
}

template <int BM, int C, int F, int NCL>
int launch_cl(const ChainParams& p, hipStream_t st) {
    constexpr size_t lds = 2 * BM * (C + 4) * 4 + BM * (C + 8) * 2 + 2 * BM * (F / NCL + 8) * 2;
    int rc = (int)hipFuncSetAttribute((const void*)layer_chain_fwd_cl_kernel<BM, C, F, NCL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    const int nblk = (p.R + BM - 1) / BM;
    ChainParams q = p;
    q.nx = cluster_xcds(nblk, NCL);
    hipLaunchKernelGGL((layer_chain_fwd_cl_kernel<BM, C, F, NCL>), dim3(8 * NCL * ((nblk + q.nx - 1) / q.nx)), dim3(NT), lds, st, q);
    return (int)hipGetLastError();
}

using mobgt_pack::PackJobs;

__global__ __launch_bounds__(256) void pack_mfma_b_kernel(const PackJobs jobs, int njobs, int nvb) {
    mobgt_pack::pack_blocks<1>(jobs, njobs, (int)blockIdx.x, 0, nvb);
}

template <int BM, int C, int F>
int launch(const ChainParams& p, hipStream_t st) {
    constexpr size_t lds = BM * (C + 4) * 4 + BM * (C + 8) * 2 + 2 * BM * (F + 8) * 2;
    static_assert(lds <= 152 * 1024, "LDS plan");
    int rc = (int)hipFuncSetAttribute((const void*)layer_chain_fwd_kernel<BM, C, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    hipLaunchKernelGGL((layer_chain_fwd_kernel<BM, C, F>), dim3((p.R + BM - 1) / BM), dim3(NT), lds, st, p);
    return (int)hipGetLastError();
}


// ---- the encoder's input rows and the FIRST layer's QKV projection in one launch ---------------------------------------------
// mobgt_assemble_tokens_fwd (model_fqandtoyo.py:1287-1298, 348-358, 1338-1347: graph token + pe[0], node features * mask +
// additive rows, positional and input dropout) for 16 rows per workgroup, then qkv = rows Wqkv^T + bqkv from LDS -- what
// every later layer gets from the previous layer's chain launch.  Masks, salts and row numbering are assemble_tokens_kernel's.
struct AsmQkvParams {
    const float *nf, *real, *add, *token, *pe0;
    float* out;                              // [G*T, C] f32
    uint16_t* out16;                         // [G*T, C] bf16
    const uint16_t *wq, *bq;                 // packed [3C,C], [3C]
    uint16_t* qkv;                           // [G*T, 3C] bf16
    int G, N;
    uint32_t thr_pos, thr_in;
    float keep_pos, keep_in;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt_nf, salt_tok, salt_in;
};

template <int C>
__global__ __launch_bounds__(NT) void assemble_qkv_kernel(const AsmQkvParams p) {
    constexpr int BM = 16, LDA = C + 8, LDQ = 3 * C + 8;
    __shared__ __attribute__((aligned(16))) uint16_t ab[BM * LDA];
    __shared__ __attribute__((aligned(16))) uint16_t qb[BM * LDQ];
    const int T = p.N + 1;
    const int64_t rows = (int64_t)p.G * T;
    const int r0 = blockIdx.x * BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
    WPre<3 * C, C> pre;
    pre.issue(p.wq);
    for (int r = wave; r < BM; r += NW) {
        const int64_t row = r0 + r;
        const bool on = row < rows;
        const int g = on ? (int)(row / T) : 0, t = on ? (int)(row - (int64_t)g * T) : 0;
        const bool tok = t == 0;
        const uint32_t r1 = tok ? (uint32_t)g : (uint32_t)(g * p.N + (t - 1));
        const uint32_t h1 = p.thr_pos ? dropout_row_hash(seed, r1 ^ (tok ? p.salt_tok : p.salt_nf)) : 0u;
        const uint32_t h2 = p.thr_in ? dropout_row_hash(seed, (uint32_t)row ^ p.salt_in) : 0u;
        const int64_t src = ((int64_t)g * p.N + (t - 1)) * C;
        const float rl = (tok || !on) ? 1.f : p.real[(int64_t)g * p.N + (t - 1)];
        for (int c = lane; c < C; c += 64) {
            float o = 0.f;
            if (on) {
                float scale = 1.f;
                if (p.thr_pos) scale = dropout_bits16(seed, h1, (uint32_t)c) >= p.thr_pos ? p.keep_pos : 0.f;
                if (p.thr_in) scale *= dropout_bits16(seed, h2, (uint32_t)c) >= p.thr_in ? p.keep_in : 0.f;
                const float v = tok ? p.token[c] + p.pe0[c] : p.nf[src + c] * rl + p.add[src + c];
                o = v * scale;
                p.out[row * C + c] = o;
                p.out16[row * C + c] = bf16_bits(o);
            }
            ab[r * LDA + c] = bf16_bits(o);
        }
    }
    __syncthreads();
    const int j = lane & 15, q = lane >> 4;
    wg_gemm<BM, 3 * C, C, LDA>(ab, p.wq, [&](int gq, const f32x4 (&acc)[1]) {
        const int col = 16 * gq + j;
        const float bias = bf16_val(p.bq[col]);
#pragma unroll
        for (int v = 0; v < 4; ++v) qb[(4 * q + v) * LDQ + col] = bf16_bits(acc[0][v] + bias);
    }, pre);
    __syncthreads();
    store_rows<BM, 3 * C, LDQ>(qb, p.qkv, r0, (int)rows);
}

// ---- Round 4: everything between the GCN tables and the first layer's attention in ONE launch ---------------------------------
// model_fqandtoyo.py:1259-1298 (row gathers), :444-456 / :1268-1269 (FuseEmbeddings-2 and -4: Linear + LeakyReLU), :1287-1298,
// :348-358, :1338-1347 (token assembly, positional and input dropout) and the first encoder layer's QKV projection.  After round 3
// that was four launches (mobgt_embed_gather_multi -> two small f32 GEMMs -> mobgt_assemble_tokens_qkv, 24-27 us for 608 rows).
// Here a workgroup owns 16 encoder-input rows (g, t): it gathers its positions' table rows into LDS (the jobs are
// mobgt_embed_gather_multi's, results also written out: the backward pass and the weight gradients read pt / x4 / nf),
// runs pt -> f2 = leaky(pt W2^T + b2) -> x4 = [f2 | cat] -> nf = leaky(x4 W4^T + b4) on v_mfma_f32_16x16x4_f32 (FULL f32
// products, as the small-GEMM launches did: one 16-column block per wave, its whole slice of the f32 weight requested before
// the gather starts), then continues exactly as assemble_qkv_kernel.  Masks, salts and row numbering are unchanged.
struct TokGather {                           // mobgt_embed_gather_multi's jobs of the fq model, sorted by destination
    static constexpr int MAXFOLD = 3, NIDX = 7;
    const float* s0_tab[2];                  // -> pt: at most two tables side by side ([poi | time]), plain copies
    const void* s0_idx[2];
    int s0_w[2], s0_coff[2], n0;
    const float* fold[MAXFOLD];              // partial tables added to s0_tab[0]'s row (same width, same index)
    int n_fold;
    const float* s1_tab;                     // -> the trailing columns of x4 (the category row)
    const void* s1_idx;
    int s1_w, s1_coff;
    const float* s2_tab[4];                  // -> add: the SUM of up to four C-wide rows
    const void* s2_idx[4];
    int n2, idx64;
};

struct TokFwdParams {
    AsmQkvParams a;                          // a.nf / a.add: the buffers nf and add are WRITTEN to (this launch produces them)
    TokGather g;
    float *pt, *x4;                          // [G*N, W2], [G*N, C]
    const float *w2, *b2, *w4, *b4;          // F.linear layout [out, in], f32
    float slope2, slope4;
};

template <int C, int W2>
__global__ __launch_bounds__(NT) void token_fwd_chain_kernel(const TokFwdParams p) {
    constexpr int BM = 16, LDA = C + 8, LDQ = 3 * C + 8, LDP = W2 + 4, LDX = C + 4;
    static_assert(W2 % 16 == 0 && C % 16 == 0 && W2 / 16 <= NW && C / 16 <= NW && W2 < C, "one column block per wave");
    __shared__ __attribute__((aligned(16))) uint16_t ab[BM * LDA];
    __shared__ __attribute__((aligned(16))) uint16_t qb[BM * LDQ];
    __shared__ __attribute__((aligned(16))) float pts[BM * LDP];
    __shared__ __attribute__((aligned(16))) float x4s[BM * LDX];
    __shared__ __attribute__((aligned(16))) float adds[BM * LDX];
    __shared__ __attribute__((aligned(16))) float nfs[BM * LDX];
    __shared__ int pos_s[BM];
    __shared__ int idx_s[TokGather::NIDX * BM];
    const AsmQkvParams& a = p.a;
    const int T = a.N + 1;
    const int64_t rows = (int64_t)a.G * T;
    const int r0 = blockIdx.x * BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
    // ---- the gathers.  First EVERY index of the tile (16 rows x 7 index lists) in one round trip, into LDS ...
    if (threadIdx.x < BM * TokGather::NIDX) {
        const int r = threadIdx.x % BM, which = threadIdx.x / BM;
        const int64_t row = r0 + r;
        const bool on = row < rows;
        const int g = on ? (int)(row / T) : 0, t = on ? (int)(row - (int64_t)g * T) : 0;
        const bool live = on && t > 0;
        const int64_t pos = (int64_t)g * a.N + (t - 1);
        // (selected, not indexed: the argument struct stays in scalar registers)
        const void* ip = which == 0 ? p.g.s0_idx[0] : which == 1 ? p.g.s0_idx[1] : which == 2 ? p.g.s1_idx
                       : which == 3 ? p.g.s2_idx[0] : which == 4 ? p.g.s2_idx[1] : which == 5 ? p.g.s2_idx[2] : p.g.s2_idx[3];
        const bool have = which < 2 ? which < p.g.n0 : (which == 2 ? true : which - 3 < p.g.n2);
        int v = -1;
        if (live && have) v = p.g.idx64 ? (int)reinterpret_cast<const int64_t*>(ip)[pos] : reinterpret_cast<const int32_t*>(ip)[pos];
        idx_s[which * BM + r] = v;
        if (which == 0) pos_s[r] = live ? (int)pos : -1;
    }
    // what the token assembly reads from global memory, requested now: graph token + pe[0] at this lane's columns, `real` of the rows
    float tokpe[C / 64];
#pragma unroll
    for (int u = 0; u < C / 64; ++u) tokpe[u] = a.token[lane + 64 * u] + a.pe0[lane + 64 * u];
    float rl_pre[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t row = r0 + wave + NW * u;
        const bool on = row < rows && wave + NW * u < BM;
        const int g = on ? (int)(row / T) : 0, t = on ? (int)(row - (int64_t)g * T) : 0;
        rl_pre[u] = (on && t > 0) ? a.real[(int64_t)g * a.N + (t - 1)] : 1.f;
    }
    __syncthreads();
    // ... then ALL table reads of the tile in flight at once: waves 0-7 take [poi (+ partial tables) | time] and the category row
    // of two rows each, waves 8-11 the four additive rows of four rows each
    {
        const int c = lane * 4;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wave < 8) {
            float4 v0[2], v1[2], vc[2], fo[2][TokGather::MAXFOLD];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 2 * wave + u;
                const int t0 = idx_s[0 * BM + r], t1 = idx_s[1 * BM + r], tc = idx_s[2 * BM + r];
                v0[u] = (t0 >= 0 && c < p.g.s0_w[0]) ? *reinterpret_cast<const float4*>(p.g.s0_tab[0] + (int64_t)t0 * p.g.s0_w[0] + c) : z4;
#pragma unroll
                for (int f = 0; f < TokGather::MAXFOLD; ++f)
                    fo[u][f] = (f < p.g.n_fold && t0 >= 0 && c < p.g.s0_w[0]) ? *reinterpret_cast<const float4*>(p.g.fold[f] + (int64_t)t0 * p.g.s0_w[0] + c) : z4;
                v1[u] = (p.g.n0 > 1 && t1 >= 0 && c < p.g.s0_w[1]) ? *reinterpret_cast<const float4*>(p.g.s0_tab[1] + (int64_t)t1 * p.g.s0_w[1] + c) : z4;
                vc[u] = (tc >= 0 && c < p.g.s1_w) ? *reinterpret_cast<const float4*>(p.g.s1_tab + (int64_t)tc * p.g.s1_w + c) : z4;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 2 * wave + u, pos = pos_s[r];
                if (c < W2) *reinterpret_cast<float4*>(pts + r * LDP + c) = z4;      // (columns no table covers)
                if (c >= W2 && c < C) *reinterpret_cast<float4*>(x4s + r * LDX + c) = z4;
                float4 val = v0[u];
#pragma unroll
                for (int f = 0; f < TokGather::MAXFOLD; ++f) { val.x += fo[u][f].x; val.y += fo[u][f].y; val.z += fo[u][f].z; val.w += fo[u][f].w; }
                if (c < p.g.s0_w[0]) *reinterpret_cast<float4*>(pts + r * LDP + p.g.s0_coff[0] + c) = val;
                if (p.g.n0 > 1 && c < p.g.s0_w[1]) *reinterpret_cast<float4*>(pts + r * LDP + p.g.s0_coff[1] + c) = v1[u];
                if (c < p.g.s1_w) *reinterpret_cast<float4*>(x4s + r * LDX + p.g.s1_coff + c) = vc[u];
                if (pos >= 0) {                     // what the backward pass reads: pt and the gathered (trailing) columns of x4
                    if (c < W2) *reinterpret_cast<float4*>(p.pt + (int64_t)pos * W2 + c) = *reinterpret_cast<const float4*>(pts + r * LDP + c);
                    if (c >= W2 && c < C) *reinterpret_cast<float4*>(p.x4 + (int64_t)pos * C + c) = *reinterpret_cast<const float4*>(x4s + r * LDX + c);
                }
            }
        } else {
            float4 va[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * (wave - 8) + u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tq = idx_s[(3 + q) * BM + r];
                    va[u][q] = (q < p.g.n2 && tq >= 0 && c < C) ? *reinterpret_cast<const float4*>(p.g.s2_tab[q] + (int64_t)tq * C + c) : z4;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * (wave - 8) + u, pos = pos_s[r];
                float4 val = va[u][0];
#pragma unroll
                for (int q = 1; q < 4; ++q) { val.x += va[u][q].x; val.y += va[u][q].y; val.z += va[u][q].z; val.w += va[u][q].w; }
                if (c < C) {
                    *reinterpret_cast<float4*>(adds + r * LDX + c) = val;
                    if (pos >= 0) *reinterpret_cast<float4*>(const_cast<float*>(a.add) + (int64_t)pos * C + c) = val;
                }
            }
        }
    }
    // this wave's 16 output columns of W2 (F.linear: row = output column) and of W4 -- requested only now: issued in front of
    // the gathers their 88 registers pushed the table reads' destinations into scratch
    float4 wb2[W2 / 16];
    if (wave < W2 / 16) {
#pragma unroll
        for (int s = 0; s < W2 / 16; ++s) wb2[s] = *reinterpret_cast<const float4*>(p.w2 + (int64_t)(16 * wave + i) * W2 + 16 * s + 4 * kq);
    }
    float4 wb4[C / 16];
    if (wave < C / 16) {
#pragma unroll
        for (int s = 0; s < C / 16; ++s) wb4[s] = *reinterpret_cast<const float4*>(p.w4 + (int64_t)(16 * wave + i) * C + 16 * s + 4 * kq);
    }
    __syncthreads();
    // ---- f2 = leaky(pt W2^T + b2) -> the leading W2 columns of x4
    if (wave < W2 / 16) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < W2 / 16; ++s) {
            const float4 av = *reinterpret_cast<const float4*>(pts + i * LDP + 16 * s + 4 * kq);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wb2[s].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wb2[s].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wb2[s].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wb2[s].w, acc, 0, 0, 0);
        }
        const int col = 16 * wave + i;
        const float bv = p.b2[col];
#pragma unroll
        for (int v = 0; v < 4; ++v) {                       // register v of lane (i, kq): row 4 kq + v, column i of the block
            const int r = 4 * kq + v;
            float o = acc[v] + bv;
            o = o > 0.f ? o : p.slope2 * o;
            x4s[r * LDX + col] = o;
            const int pos = pos_s[r];
            if (pos >= 0) p.x4[(int64_t)pos * C + col] = o;
        }
    }
    WPre<3 * C, C> pre;
    pre.issue(a.wq);                                        // the QKV weight's first chunks: in flight during the second product
    __syncthreads();
    // ---- nf = leaky(x4 W4^T + b4)
    if (wave < C / 16) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < C / 16; ++s) {
            const float4 av = *reinterpret_cast<const float4*>(x4s + i * LDX + 16 * s + 4 * kq);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wb4[s].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wb4[s].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wb4[s].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wb4[s].w, acc, 0, 0, 0);
        }
        const int col = 16 * wave + i;
        const float bv = p.b4[col];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = 4 * kq + v;
            float o = acc[v] + bv;
            o = o > 0.f ? o : p.slope4 * o;
            nfs[r * LDX + col] = o;
            const int pos = pos_s[r];
            if (pos >= 0) const_cast<float*>(a.nf)[(int64_t)pos * C + col] = o;
        }
    }
    __syncthreads();
    // ---- token assembly + dropouts (assemble_qkv_kernel's, with nf / add taken from LDS)
    for (int r = wave; r < BM; r += NW) {
        const int64_t row = r0 + r;
        const bool on = row < rows;
        const int g = on ? (int)(row / T) : 0, t = on ? (int)(row - (int64_t)g * T) : 0;
        const bool tok = t == 0;
        const uint32_t r1 = tok ? (uint32_t)g : (uint32_t)(g * a.N + (t - 1));
        const uint32_t h1 = a.thr_pos ? dropout_row_hash(seed, r1 ^ (tok ? a.salt_tok : a.salt_nf)) : 0u;
        const uint32_t h2 = a.thr_in ? dropout_row_hash(seed, (uint32_t)row ^ a.salt_in) : 0u;
        const float rl = r < NW ? rl_pre[0] : rl_pre[1];
#pragma unroll
        for (int u = 0; u < C / 64; ++u) {
            const int c = lane + 64 * u;
            float o = 0.f;
            if (on) {
                float scale = 1.f;
                if (a.thr_pos) scale = dropout_bits16(seed, h1, (uint32_t)c) >= a.thr_pos ? a.keep_pos : 0.f;
                if (a.thr_in) scale *= dropout_bits16(seed, h2, (uint32_t)c) >= a.thr_in ? a.keep_in : 0.f;
                const float v = tok ? tokpe[u] : nfs[r * LDX + c] * rl + adds[r * LDX + c];
                o = v * scale;
                a.out[row * C + c] = o;
                a.out16[row * C + c] = bf16_bits(o);
            }
            ab[r * LDA + c] = bf16_bits(o);
        }
    }
    __syncthreads();
    const int j = lane & 15, q = lane >> 4;
    wg_gemm<BM, 3 * C, C, LDA>(ab, a.wq, [&](int gq, const f32x4 (&acc)[1]) {
        const int col = 16 * gq + j;
        const float bias = bf16_val(a.bq[col]);
#pragma unroll
        for (int v = 0; v < 4; ++v) qb[(4 * q + v) * LDQ + col] = bf16_bits(acc[0][v] + bias);
    }, pre);
    __syncthreads();
    store_rows<BM, 3 * C, LDQ>(qb, a.qkv, r0, (int)rows);
}

// ---- the same chain backwards: from d(out) down to the gradient of the attention output ------------------------------
//     dx2 = ffn_norm2'(dout);  df = dropout'(dx2)                      [dnxw, dnxb, db2]
//     du  = (df W2) * gelu'(u)
//     dz  = du W1
//     dx1 = dx2 + ffn_norm1'(dz);  dy = dropout'(dx1)                  [dn1w, dn1b, dbo]
//     da  = dy Wo
// (mobgt_dropout_add_ln_bwd, mobgt_layer_gemm GELU_BWD / plain, mobgt_dropout_add_ln_bwd, mobgt_layer_gemm as five
// launches: 28.7 us of the S-FSQ layer's backward.)  The weights are the TRANSPOSES packed in operand order (dX = dY W:
// W is the [K][N] operand).  df, du, dy (the weight-gradient GEMMs read them), da (the attention backward) and dx1 (the
// residual gradient the QKV data gradient is added to) go to global memory; the six column sums leave the workgroup as one
// atomic per column.
struct ChainBwdParams {
    const float *dout, *x2, *x1;             // [R,C] f32
    const uint16_t* u;                       // [R,F] bf16
    const float *mean1, *rstd1, *mean2, *rstd2, *n1w, *nxw;
    const uint16_t *w2t, *w1t, *wot;         // packed transposes: (N=F,K=C), (N=C,K=F), (N=C,K=C)
    uint16_t *df, *du, *dy, *da;             // bf16 [R,C] [R,F] [R,C] [R,C]
    float* dx1;                              // [R,C] f32
    float *dnxw, *dnxb, *db2, *dn1w, *dn1b, *dbo;     // [C] f32, accumulated
    float* db1;                              // [F] f32, accumulated: the 64-row form only (null: the caller sums du's columns)
    int R;
    uint32_t thr;
    float inv_keep;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt1, salt2;
    // What the layer ABOVE left undone (optional).  Its input gradient is dx1 + dqkv Wqkv: `dout` then holds only dx1 and the
    // product is finished here, per row block, in front of the first norm (t_dqkv [R,3C] bf16, t_wqt = Wqkv^T packed);
    // and its four weight-gradient problems ride in this launch as extra workgroups (blockIdx >= n_chain) on the ~200
    // compute units the 16-row chain workgroups leave idle -- the layer above then has NO launch after its attention backward.
    const uint16_t *t_dqkv, *t_wqt;
    // pre-LN layers (model.py:479-489): the second norm of the chain is the NEXT layer's self_attention_norm, applied to the
    // residual stream x2 to form that layer's attention input; the stream itself passes by.  So
    //     dx2 = dout + norm'(dqkv_next Wqkv_next)      (fq: dx2 = norm'(dout + dqkv_next Wqkv_next))
    // and without a successor in the chain (nxw null) dx2 = dout.
    int pre_ln;
    mobgt_wgrad::WgradParams wg[4];
    int wg_first[5], wg_tiles[4], wg_splits[4];
    int n_wg, n_chain;
    uint32_t* ws_gen;                        // cluster form: launch generations and the exchange area (see cluster_put / cluster_get)
    uint64_t* ws_ll;
    int nx;                                  // ... and the number of XCDs that take clusters
};

// gelu'(u) = Phi(u) + u phi(u); the exponential of the A&S erf IS phi's
__device__ __forceinline__ float gelu_grad_f(float u) {
    const float ax = fabsf(u) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float y = fmaf(1.061405429f, t, -1.453152027f);
    y = fmaf(y, t, 1.421413741f);
    y = fmaf(y, t, -0.284496736f);
    y = fmaf(y, t, 0.254829592f);
    const float e = __expf(-ax * ax);
    const float erf = copysignf(1.f - y * t * e, u);
    return fmaf(u * 0.3989422804014327f, e, 0.5f * (1.f + erf));
}

// LayerNorm backward of the block's rows (one wave per row): d (the gradient at the norm's output) comes from `d_lds`
// (f32 [BM][LDX]) or from global `d_g`; xhat from the saved pre-norm rows and statistics; `res` (LDS f32, or null) is added;
// the result goes to `dx_lds` (f32, LDS), optionally `dx_g` (global f32), and its dropout'ed bf16 form to `dy_lds` (the next
// GEMM's A operand) and `dy_g`.  Column sums of (d * xhat, d, dy) over the block's rows are accumulated into `red`
// ([3][NW][C] f32, LDS; the caller reduces over the waves).
// what a LayerNorm backward pass reads of its SAVED forward (pre-norm rows, statistics, weight) for this wave's rows,
// requested early: for the second norm of the backward chain at kernel start, two GEMMs ahead of its use
template <int BM, int C>
struct LnBwdPre {
    static constexpr int PER = (C + 63) / 64, NR = (BM + NW - 1) / NW;
    float w[PER], x[NR][PER], mu[NR], rs[NR];
    __device__ __forceinline__ void issue(const float* __restrict__ xpre, const float* __restrict__ g_mean,
                                          const float* __restrict__ g_rstd, const float* __restrict__ wp, int r0, int R) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (!wp) {          // no norm at this point of the chain (pre-LN, no successor): zeros make norm'(0) + res = res exactly
#pragma unroll
            for (int k = 0; k < PER; ++k) w[k] = 0.f;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                mu[i] = rs[i] = 0.f;
#pragma unroll
                for (int k = 0; k < PER; ++k) x[i][k] = 0.f;
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) w[k] = wp[min(lane + 64 * k, C - 1)];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int64_t row = min(r0 + min(wave + NW * i, BM - 1), R - 1);
            mu[i] = g_mean[row];
            rs[i] = g_rstd[row];
#pragma unroll
            for (int k = 0; k < PER; ++k) x[i][k] = xpre[row * C + min(lane + 64 * k, C - 1)];
        }
    }
};

template <int BM, int C, int LDX, int LDA>
__device__ __forceinline__ void ln_bwd_rows(const float* __restrict__ d_lds, const float* __restrict__ d_g,
                                            const LnBwdPre<BM, C>& pre,
                                            const float* res, float* dx_lds, float* __restrict__ dx_g,
                                            uint16_t* __restrict__ dy_lds, uint16_t* __restrict__ dy_g, float* __restrict__ red,
                                            int r0, int R, uint32_t thr, float inv_keep, uint64_t seed, uint32_t salt,
                                            int own_mod = 1, int own_rem = 0) {
    constexpr int PER = (C + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float (&wv)[PER] = pre.w;
    float ag[PER], ab[PER], ay[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) ag[k] = ab[k] = ay[k] = 0.f;
#pragma unroll
    for (int i = 0; i < LnBwdPre<BM, C>::NR; ++i) {
        const int r = wave + NW * i;
        if (r >= BM) break;
        const int64_t row = min(r0 + r, R - 1);
        const bool on = r0 + r < R;
        const float mu = pre.mu[i], rs = pre.rs[i];
        float d[PER], xh[PER], gg[PER], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = min(lane + 64 * k, C - 1);
            const bool in = lane + 64 * k < C && on;
            const float dv = d_lds ? d_lds[r * LDX + c] : d_g[row * C + c];
            d[k] = in ? dv : 0.f;
            xh[k] = in ? (pre.x[i][k] - mu) * rs : 0.f;
            gg[k] = d[k] * wv[k];
            ag[k] += d[k] * xh[k];
            ab[k] += d[k];
            s1 += gg[k];
            s2 += gg[k] * xh[k];
        }
        s1 = wave64_sum(s1) * (1.f / C);
        s2 = wave64_sum(s2) * (1.f / C);
        const uint32_t rowh = thr ? dropout_row_hash(seed, (uint32_t)(r0 + r) ^ salt) : 0u;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                float t = rs * (gg[k] - s1 - xh[k] * s2);
                if (res) t += res[r * LDX + c];
                if (!on) t = 0.f;
                dx_lds[r * LDX + c] = t;
                float yv = t;
                if (thr) yv = dropout_bits16(seed, rowh, (uint32_t)c) >= thr ? t * inv_keep : 0.f;
                ay[k] += yv;
                const uint16_t yb = bf16_bits(yv);
                dy_lds[r * LDA + c] = yb;
                if (on && (own_mod == 1 || r % own_mod == own_rem)) {
                    if (dx_g) dx_g[row * C + c] = t;
                    dy_g[row * C + c] = yb;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
            red[(0 * NW + wave) * C + c] = ag[k];
            red[(1 * NW + wave) * C + c] = ab[k];
            red[(2 * NW + wave) * C + c] = ay[k];
        }
    }
}

template <int C>
__device__ __forceinline__ void flush_colsums(const float* __restrict__ red, float* g0, float* g1, float* g2, int c_lo = 0, int c_n = C) {
    for (int e = threadIdx.x; e < 3 * c_n; e += NT) {
        const int which = e / c_n, c = c_lo + e % c_n;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[(which * NW + w) * C + c];
        float* dst = which == 0 ? g0 : (which == 1 ? g1 : g2);
        if (dst) atomicAdd(dst + c, s);
    }
}

// ln_bwd_rows with half a wave per row (the cluster kernels; see ln_rows_hw): ONE pass over the 16 rows; the column-sum
// partials are per ROW -- red is [3][BM][C] -- and flush_colsums_rows adds the rows up
template <int BM, int C>
struct LnBwdPreH {
    static constexpr int PER = C / 32;
    float w[PER], x[PER], mu, rs;
    __device__ __forceinline__ void issue(const float* __restrict__ xpre, const float* __restrict__ g_mean,
                                          const float* __restrict__ g_rstd, const float* __restrict__ wp, int r0, int R) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31;
        if (!wp) {          // (see LnBwdPre)
            mu = rs = 0.f;
#pragma unroll
            for (int k = 0; k < PER; ++k) w[k] = x[k] = 0.f;
            return;
        }
        const int64_t row = min(r0 + min(2 * wave + (lane >> 5), BM - 1), R - 1);
        mu = g_mean[row];
        rs = g_rstd[row];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            w[k] = wp[hl + 32 * k];
            x[k] = xpre[row * C + hl + 32 * k];
        }
    }
};
template <int BM, int C, int LDX, int LDA>
__device__ __forceinline__ void ln_bwd_rows_hw(const float* __restrict__ d_lds, const float* __restrict__ d_g,
                                               const LnBwdPreH<BM, C>& pre, const float* res, float* dx_lds, float* __restrict__ dx_g,
                                               uint16_t* __restrict__ dy_lds, uint16_t* __restrict__ dy_g, float* __restrict__ red,
                                               int r0, int R, uint32_t thr, float inv_keep, uint64_t seed, uint32_t salt,
                                               int own_mod, int own_rem) {
    static_assert(C % 64 == 0 && BM <= 2 * NW, "half a wave per row");
    constexpr int PER = C / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31;
    const int r = 2 * wave + (lane >> 5);
    if (r >= BM) return;
    const int64_t row = min(r0 + r, R - 1);
    const bool on = r0 + r < R;
    float d[PER], xh[PER], gg[PER], a1 = 0.f, b1 = 0.f, a2 = 0.f, b2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = hl + 32 * k;
        const float dv = d_lds ? d_lds[r * LDX + c] : d_g[row * C + c];
        d[k] = on ? dv : 0.f;
        xh[k] = on ? (pre.x[k] - pre.mu) * pre.rs : 0.f;
        gg[k] = d[k] * pre.w[k];
        if (k & 1) { b1 += gg[k]; b2 += gg[k] * xh[k]; } else { a1 += gg[k]; a2 += gg[k] * xh[k]; }
    }
    const float s1 = half_sum_as_wave(a1, b1) * (1.f / C);
    const float s2 = half_sum_as_wave(a2, b2) * (1.f / C);
    const uint32_t rowh = thr ? dropout_row_hash(seed, (uint32_t)(r0 + r) ^ salt) : 0u;
    const bool store = on && (own_mod == 1 || r % own_mod == own_rem);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = hl + 32 * k;
        float t = pre.rs * (gg[k] - s1 - xh[k] * s2);
        if (res) t += res[r * LDX + c];
        if (!on) t = 0.f;
        dx_lds[r * LDX + c] = t;
        float yv = t;
        if (thr) yv = dropout_bits16(seed, rowh, (uint32_t)c) >= thr ? t * inv_keep : 0.f;
        const uint16_t yb = bf16_bits(yv);
        dy_lds[r * LDA + c] = yb;
        if (store) {
            if (dx_g) dx_g[row * C + c] = t;
            dy_g[row * C + c] = yb;
        }
        red[(0 * BM + r) * C + c] = d[k] * xh[k];
        red[(1 * BM + r) * C + c] = d[k];
        red[(2 * BM + r) * C + c] = yv;
    }
}
template <int BM, int C>
__device__ __forceinline__ void flush_colsums_rows(const float* __restrict__ red, float* g0, float* g1, float* g2, int c_lo, int c_n) {
    for (int e = threadIdx.x; e < 3 * c_n; e += NT) {
        const int which = e / c_n, c = c_lo + e % c_n;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < BM; ++r) s += red[(which * BM + r) * C + c];
        float* dst = which == 0 ? g0 : (which == 1 ? g1 : g2);
        if (dst) atomicAdd(dst + c, s);
    }
}

template <int BM, int C, int F>
__global__ __launch_bounds__(NT) void layer_chain_bwd_kernel(const ChainBwdParams p) {
    constexpr int LDA = C + 8, LDH = F + 8, LDX = C + 4, MT = BM / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* dxb = reinterpret_cast<float*>(smem_raw);                             // [BM][LDX] f32: dx2, then dx1
    float* dzb = dxb + BM * LDX;                                                 // [BM][LDX] f32: dz (bf16-rounded values)
    float* red = dzb + BM * LDX;                                                 // [3][NW][C] f32 column-sum partials
    uint16_t* gb = reinterpret_cast<uint16_t*>(red + 3 * NW * C);                // [BM][LDA] bf16: df, then dy
    uint16_t* ub = gb + BM * LDA;                                                // [BM][LDH] bf16: u, later da
    uint16_t* dub = ub + BM * LDH;                                               // [BM][LDH] bf16: du
    if ((int)blockIdx.x >= p.n_chain) {          // passenger: one 32 x 32 tile of one of the upper layer's weight gradients
        const int bid = (int)blockIdx.x - p.n_chain;
        int qn = 0;
#pragma unroll
        for (int t = 1; t < 4; ++t)
            if (t < p.n_wg && bid >= p.wg_first[t]) qn = t;
        const int local = bid - p.wg_first[qn];
        mobgt_wgrad::wgrad_body<false, NW>(p.wg[qn], local % p.wg_tiles[qn], local / p.wg_tiles[qn], p.wg_splits[qn],
                                           reinterpret_cast<float*>(smem_raw));
        return;
    }
    const int r0 = blockIdx.x * BM;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;

    WPre<C, 3 * C> pre0;
    if (p.t_dqkv) pre0.issue(p.t_wqt);
    WPre<F, C> pre1;
    pre1.issue(p.w2t);
    LnBwdPre<BM, C> lp2, lp1;
    lp2.issue(p.x2, p.mean2, p.rstd2, p.nxw, r0, p.R);
    lp1.issue(p.x1, p.mean1, p.rstd1, p.n1w, r0, p.R);        // (used two GEMMs from here)
    for (int e = threadIdx.x; e < BM * (F / 8); e += NT) {                        // the block's rows of u -> LDS
        const int r = e / (F / 8), c = (e % (F / 8)) * 8;
        *reinterpret_cast<uint4*>(ub + r * LDH + c) = *reinterpret_cast<const uint4*>(p.u + (int64_t)min(r0 + r, p.R - 1) * F + c);
    }
    if (p.pre_ln) {
        // pre-LN: dout -> dxb (the residual path: added behind the norm's backward), dzb = 0 (+ the tail product below)
        for (int e = threadIdx.x; e < BM * (C / 4); e += NT) {
            const int r = e / (C / 4), c = (e % (C / 4)) * 4;
            *reinterpret_cast<float4*>(dxb + r * LDX + c) = *reinterpret_cast<const float4*>(p.dout + (int64_t)min(r0 + r, p.R - 1) * C + c);
            *reinterpret_cast<float4*>(dzb + r * LDX + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (p.t_dqkv) {
        // ---- the upper layer's input gradient, finished here: dout <- dx1 (what `dout` holds) + dqkv Wqkv
        if (!p.pre_ln)
        for (int e = threadIdx.x; e < BM * (C / 4); e += NT) {
            const int r = e / (C / 4), c = (e % (C / 4)) * 4;
            *reinterpret_cast<float4*>(dzb + r * LDX + c) = *reinterpret_cast<const float4*>(p.dout + (int64_t)min(r0 + r, p.R - 1) * C + c);
        }
        for (int e = threadIdx.x; e < BM * (3 * C / 8); e += NT) {
            const int r = e / (3 * C / 8), c = (e % (3 * C / 8)) * 8;
            *reinterpret_cast<uint4*>(dub + r * LDH + c) = *reinterpret_cast<const uint4*>(p.t_dqkv + (int64_t)min(r0 + r, p.R - 1) * (3 * C) + c);
        }
        __syncthreads();
        wg_gemm<BM, C, 3 * C, LDH>(dub, p.t_wqt, [&](int g, const f32x4 (&acc)[MT]) {
            const int col = 16 * g + j;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) dzb[(16 * t + 4 * q + v) * LDX + col] += acc[t][v];
        }, pre0);
        __syncthreads();
    }
    // ---- dx2 = ffn_norm2'(dout);  df = dropout'(dx2)      (pre-LN: dx2 = dout + norm'(tail product), see ChainBwdParams)
    if (p.pre_ln) {
        if (!p.t_dqkv) __syncthreads();
        ln_bwd_rows<BM, C, LDX, LDA>(dzb, nullptr, lp2, dxb, dxb, nullptr, gb, p.df, red, r0, p.R, p.thr, p.inv_keep, seed, p.salt2);
    } else
    ln_bwd_rows<BM, C, LDX, LDA>(p.t_dqkv ? dzb : nullptr, p.dout, lp2, nullptr, dxb, nullptr, gb, p.df, red, r0, p.R,
                                 p.thr, p.inv_keep, seed, p.salt2);
    __syncthreads();
    flush_colsums<C>(red, p.dnxw, p.dnxb, p.db2);
    // ---- du = (df W2) * gelu'(u)
    WPre<C, F> pre2;
    wg_gemm<BM, F, C, LDA>(gb, p.w2t, [&](int g, const f32x4 (&acc)[MT]) {
        const int col = 16 * g + j;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = 16 * t + 4 * q + v;
                dub[r * LDH + col] = bf16_bits(acc[t][v] * gelu_grad_f(bf16_val(ub[r * LDH + col])));
            }
    }, pre1);
    pre2.issue(p.w1t);
    __syncthreads();
    store_rows<BM, F, LDH>(dub, p.du, r0, p.R);
    // ---- dz = du W1  (rounded to bf16 where the separate launch wrote a bf16 tensor)
    WPre<C, C> pre3;
    wg_gemm<BM, C, F, LDH>(dub, p.w1t, [&](int g, const f32x4 (&acc)[MT]) {
        const int col = 16 * g + j;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) dzb[(16 * t + 4 * q + v) * LDX + col] = bf16_round(acc[t][v]);
    }, pre2);
    pre3.issue(p.wot);
    __syncthreads();
    // ---- dx1 = dx2 + ffn_norm1'(dz);  dy = dropout'(dx1)
    ln_bwd_rows<BM, C, LDX, LDA>(dzb, nullptr, lp1, dxb, dxb, p.dx1, gb, p.dy, red, r0, p.R, p.thr,
                                 p.inv_keep, seed, p.salt1);
    __syncthreads();
    flush_colsums<C>(red, p.dn1w, p.dn1b, p.dbo);
    // ---- da = dy Wo
    wg_gemm<BM, C, C, LDA>(gb, p.wot, [&](int g, const f32x4 (&acc)[MT]) {
        const int col = 16 * g + j;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) ub[(16 * t + 4 * q + v) * LDH + col] = bf16_bits(acc[t][v]);
    }, pre3);
    __syncthreads();
    store_rows<BM, C, LDH>(ub, p.da, r0, p.R);
}


// the backward chain, cluster form (see layer_chain_fwd_cl_kernel): member m owns the column range m of du (and of u) and of
// da, and the K range m of dz = du W1; the small products (the layer above's tail dqkv Wqkv, 221 KB) run in full on every member
template <int BM, int C, int F, int NCL>
__global__ __launch_bounds__(NT) void layer_chain_bwd_cl_kernel(const ChainBwdParams p) {
    constexpr int FS = F / NCL, CS = C / NCL;
    constexpr int LDA = C + 8, LDHS = FS + 8, LDX = C + 4, LDQ = 3 * C + 8, MT = BM / 16;
    static_assert(BM == 16 && CS % 16 == 0 && FS % 32 == 0 && CS <= FS, "cluster split");
    static_assert(3 * BM * C * 4 >= BM * LDQ * 2, "the dqkv tile lives where the column-sum partials go later");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* dxb = reinterpret_cast<float*>(smem_raw);                             // [BM][LDX] f32: dx2, then dx1
    float* dzb = dxb + BM * LDX;                                                 // [BM][LDX] f32: d(out), later dz
    float* pb = dzb + BM * LDX;                                                  // [BM][LDX] f32: this member's partial sums of dz
    float* red = pb + BM * LDX;                                                  // [3][BM][C] f32 column-sum partials (per row)
    uint16_t* tq = reinterpret_cast<uint16_t*>(red);                             //   (before them: [BM][LDQ] bf16, the upper layer's dqkv rows)
    uint16_t* gb = reinterpret_cast<uint16_t*>(red + 3 * BM * C);                // [BM][LDA] bf16: df, then dy
    uint16_t* ub = gb + BM * LDA;                                                // [BM][LDHS] bf16: this member's columns of u, later of da
    uint16_t* dub = ub + BM * LDHS;                                              // [BM][LDHS] bf16: ... of du
    if ((int)blockIdx.x >= p.n_chain) {          // passenger: one 32 x 32 tile of one of the upper layer's weight gradients
        const int bid = (int)blockIdx.x - p.n_chain;
        int qn = 0;
#pragma unroll
        for (int t = 1; t < 4; ++t)
            if (t < p.n_wg && bid >= p.wg_first[t]) qn = t;
        const int local = bid - p.wg_first[qn];
        mobgt_wgrad::wgrad_body<false, NW>(p.wg[qn], local % p.wg_tiles[qn], local / p.wg_tiles[qn], p.wg_splits[qn],
                                           reinterpret_cast<float*>(smem_raw));
        return;
    }
    int blk, m;
    cluster_ids<NCL>((int)blockIdx.x, p.nx, blk, m);
    const int r0 = blk * BM;
    if (blk < 0 || r0 >= p.R) return;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;

    WPre<C, 3 * C> pre0;
    if (p.t_dqkv) pre0.issue(p.t_wqt);
    WPre<FS, C> pre1;
    pre1.issue(p.w2t, m * (FS / 16));
    LnBwdPreH<BM, C> lp2, lp1;
    lp2.issue(p.x2, p.mean2, p.rstd2, p.nxw, r0, p.R);
    lp1.issue(p.x1, p.mean1, p.rstd1, p.n1w, r0, p.R);
    for (int e = threadIdx.x; e < BM * (FS / 8); e += NT) {                       // this member's columns of u -> LDS
        const int r = e / (FS / 8), c = (e % (FS / 8)) * 8;
        *reinterpret_cast<uint4*>(ub + r * LDHS + c) = *reinterpret_cast<const uint4*>(p.u + (int64_t)min(r0 + r, p.R - 1) * F + m * FS + c);
    }
    if (p.pre_ln) {
        // pre-LN: dout -> dxb (the residual path: added behind the norm's backward), dzb = 0 (+ the tail product below)
        for (int e = threadIdx.x; e < BM * (C / 4); e += NT) {
            const int r = e / (C / 4), c = (e % (C / 4)) * 4;
            *reinterpret_cast<float4*>(dxb + r * LDX + c) = *reinterpret_cast<const float4*>(p.dout + (int64_t)min(r0 + r, p.R - 1) * C + c);
            *reinterpret_cast<float4*>(dzb + r * LDX + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (p.t_dqkv) {
        // ---- the upper layer's input gradient, finished here by every member: dout <- dx1 (what `dout` holds) + dqkv Wqkv
        if (!p.pre_ln)
        for (int e = threadIdx.x; e < BM * (C / 4); e += NT) {
            const int r = e / (C / 4), c = (e % (C / 4)) * 4;
            *reinterpret_cast<float4*>(dzb + r * LDX + c) = *reinterpret_cast<const float4*>(p.dout + (int64_t)min(r0 + r, p.R - 1) * C + c);
        }
        for (int e = threadIdx.x; e < BM * (3 * C / 8); e += NT) {
            const int r = e / (3 * C / 8), c = (e % (3 * C / 8)) * 8;
            *reinterpret_cast<uint4*>(tq + r * LDQ + c) = *reinterpret_cast<const uint4*>(p.t_dqkv + (int64_t)min(r0 + r, p.R - 1) * (3 * C) + c);
        }
        __syncthreads();
        wg_gemm<BM, C, 3 * C, LDQ>(tq, p.t_wqt, [&](int g, const f32x4 (&acc)[MT]) {
            const int col = 16 * g + j;
#pragma unroll
            for (int v = 0; v < 4; ++v) dzb[(4 * q + v) * LDX + col] += acc[0][v];
        }, pre0);
        __syncthreads();
    }
    // ---- dx2 = ffn_norm2'(dout);  df = dropout'(dx2): all rows on every member, member m writes rows m (mod NCL) and
    //      the column sums of columns [m CS, (m + 1) CS)
    const uint32_t gen = __hip_atomic_load(p.ws_gen + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (used at the hand-over)
    if (p.pre_ln) {
        if (!p.t_dqkv) __syncthreads();
        ln_bwd_rows_hw<BM, C, LDX, LDA>(dzb, nullptr, lp2, dxb, dxb, nullptr, gb, p.df, red, r0, p.R, p.thr, p.inv_keep, seed, p.salt2,
                                        NCL, m);
    } else
    ln_bwd_rows_hw<BM, C, LDX, LDA>(p.t_dqkv ? dzb : nullptr, p.dout, lp2, nullptr, dxb, nullptr, gb, p.df, red, r0, p.R,
                                 p.thr, p.inv_keep, seed, p.salt2, NCL, m);
    __syncthreads();
    flush_colsums_rows<BM, C>(red, p.dnxw, p.dnxb, p.db2, m * CS, CS);
    // ---- columns [m FS, (m + 1) FS) of du = (df W2) * gelu'(u)
    constexpr bool DEEP = NCL == 4 && C <= 192;       // (registers)
    WPre<C, FS, F / 32> pre2;
    if constexpr (DEEP) pre2.issue(p.w1t, 0, m * (FS / 32));              // (a whole phase ahead of its product)
    wg_gemm<BM, FS, C, LDA>(gb, p.w2t, [&](int g, const f32x4 (&acc)[MT]) {
        const int cl = 16 * g + j;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = 4 * q + v;
            dub[r * LDHS + cl] = bf16_bits(acc[0][v] * gelu_grad_f(bf16_val(ub[r * LDHS + cl])));
        }
    }, pre1, m * (FS / 16));
    if constexpr (!DEEP) pre2.issue(p.w1t, 0, m * (FS / 32));
    __syncthreads();
    store_cols<BM, FS, LDHS>(dub, p.du, F, m * FS, r0, p.R);            // (not in front of the hand-over's polls: they would wait for it)
    // ---- dz = du W1 over this member's K range: partial sums
    WPre<CS, C> pre3;
    wg_gemm<BM, C, FS, LDHS>(dub, p.w1t, [&](int g, const f32x4 (&acc)[MT]) {
        const int col = 16 * g + j;
#pragma unroll
        for (int v = 0; v < 4; ++v) pb[(4 * q + v) * LDX + col] = acc[0][v];
    }, pre2, 0, m * (FS / 32));
    pre3.issue(p.wot, m * (CS / 16));
    __syncthreads();
    // ---- the partial sums change hands;  dz = bf16(sum in member order)
    const __amdgpu_buffer_rsrc_t area = cluster_area<BM, C, NCL>(p.ws_ll, blk);
    cluster_put<BM, C, LDX>(area, m, pb, gen + 1u);
    cluster_get<BM, C, LDX, NCL>(area, m, pb, gen + 1u, p.ws_gen, [&](int, int r, int c, float s0, float s1) {
        dzb[r * LDX + c] = bf16_round(s0);
        dzb[r * LDX + c + 1] = bf16_round(s1);
    });
    __syncthreads();
    // (every thread of this workgroup holds every member's words: all of them have read gen)
    if (m == 0 && threadIdx.x == 0) __hip_atomic_store(p.ws_gen + blk, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- dx1 = dx2 + ffn_norm1'(dz);  dy = dropout'(dx1)
    ln_bwd_rows_hw<BM, C, LDX, LDA>(dzb, nullptr, lp1, dxb, dxb, p.dx1, gb, p.dy, red, r0, p.R, p.thr,
                                 p.inv_keep, seed, p.salt1, NCL, m);
    __syncthreads();
    flush_colsums_rows<BM, C>(red, p.dn1w, p.dn1b, p.dbo, m * CS, CS);
    // ---- columns [m CS, (m + 1) CS) of da = dy Wo
    wg_gemm<BM, CS, C, LDA>(gb, p.wot, [&](int g, const f32x4 (&acc)[MT]) {
        const int cl = 16 * g + j;
#pragma unroll
        for (int v = 0; v < 4; ++v) ub[(4 * q + v) * LDHS + cl] = bf16_bits(acc[0][v]);
    }, pre3, m * (CS / 16));
    __syncthreads();
    store_cols<BM, CS, LDHS>(ub, p.da, C, m * CS, r0, p.R);
}

template <int BM, int C, int F, int NCL>
int launch_bwd_cl(ChainBwdParams& p, hipStream_t st) {
    constexpr size_t lds = 3 * BM * (C + 4) * 4 + 3 * BM * C * 4 + BM * (C + 8) * 2 + 2 * BM * (F / NCL + 8) * 2;
    static_assert(lds <= 152 * 1024 && lds >= mobgt_wgrad::wgrad_lds_floats<NW>() * sizeof(float), "LDS plan");
    int rc = (int)hipFuncSetAttribute((const void*)layer_chain_bwd_cl_kernel<BM, C, F, NCL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    const int nblk = (p.R + BM - 1) / BM;
    p.nx = cluster_xcds(nblk, NCL);
    p.n_chain = 8 * NCL * ((nblk + p.nx - 1) / p.nx);
    hipLaunchKernelGGL((layer_chain_bwd_cl_kernel<BM, C, F, NCL>), dim3(p.n_chain + (p.n_wg ? p.wg_first[p.n_wg] : 0)), dim3(NT), lds, st, p);
    return (int)hipGetLastError();
}

// ---- the LONG-batch form of the backward chain (round 5): 64 rows per workgroup ------------------------------------------------
// d(out) -> ffn_norm2' -> dropout' -> (W2, gelu') -> W1 -> ffn_norm1' + residual -> dropout' -> Wo for 64 rows (see
// layer_chain_fwd_big_kernel: the 16-row form re-streams the weights once per 16 rows).  The FFN is walked in the forward's chunks
// of 384 hidden columns:   du_c = (df W2[:, c]) gelu'(u_c)  -- u_c staged in the chunk tile, du_c written over it --  -> global
// (the weight-gradient launches read it) and   dz += du_c W1[c, :]   split-K into the f32 tile.  dx2 leaves for global (in the
// dx1 buffer) with the first norm's backward and is read back for the second.  LDS: dz f32 [64][C + 4], df / dy bf16
// [64][C + 8], the chunk tile [64][392] (its first 24 KB double as the column-sum partials between the phases): 147 KB at
// C = 256.  Eight waves (two per SIMD, 256 VGPRs).  The layer above's tail (dout + dqkv Wqkv) can be hosted as in the 16-row form;
// no weight-gradient passengers (past 4 096 rows they are the library's); b1's gradient is summed inside.  fq (post-LN) layers.
template <int C, int F>
__global__ __launch_bounds__(BIG_NW * 64) void layer_chain_bwd_big_kernel(const ChainBwdParams p) {
    constexpr int NWB = BIG_NW, NTB = NWB * 64;
    constexpr int BM = 64, MT = 4, LDA = C + 8, LDX = C + 4, FC = 384, LDC = FC + 8, NR = BM / NWB;
    static_assert(F > 2 * FC && F <= 3 * FC && (F - 2 * FC) % 32 == 0 && C % 64 == 0 && C <= LDC - 8, "three FFN chunks");
    static_assert(3 * NWB * C * 4 <= BM * LDC * 2, "the column-sum partials fit the chunk tile");
    constexpr int FL = F - 2 * FC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* dzb = reinterpret_cast<float*>(smem_raw);                             // [BM][LDX] f32: dz
    uint16_t* gb = reinterpret_cast<uint16_t*>(dzb + BM * LDX);                  // [BM][LDA] bf16: df, then dy
    uint16_t* tb = gb + BM * LDA;                                                // [BM][LDC] bf16: a chunk of u, then of du in place; da
    float* red = reinterpret_cast<float*>(tb);                                   // [3][NWB][C] f32 column-sum partials (between the phases)
    __shared__ uint32_t s_rowh[2][BM];
    const int r0 = blockIdx.x * BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    STAMP_DECL;
    STAMP(0);
    if (threadIdx.x < 2 * BM)
        s_rowh[threadIdx.x / BM][threadIdx.x % BM] =
            dropout_row_hash(seed, (uint32_t)(r0 + (int)(threadIdx.x % BM)) ^ (threadIdx.x < BM ? p.salt1 : p.salt2));
    const uint16_t* nobias = p.u;                        // (a product without a bias: any readable bf16 vector of >= 384 entries)
    const bool hosted = p.t_dqkv != nullptr;
    if (hosted) {
        // ---- the layer ABOVE's input gradient, finished here (as the 16-row form hosts it): `dout` holds that layer's dx1; the
        //      product dqkv Wqkv goes to the dz tile (f32) once every wave has left the dqkv tile under it, and the first norm
        //      reads  d = dout + tile
        constexpr int LDQ = 3 * C + 8;
        static_assert(BM * LDQ * 2 <= BM * LDX * 4 + BM * LDA * 2, "the dqkv tile overlays the dz and df tiles");
        uint16_t* tq = reinterpret_cast<uint16_t*>(smem_raw);
        WideGemm<BM, C, 3 * C, LDQ, 3 * C / 32, NWB> g_t;
        g_t.issue(p.t_wqt, nobias);
#pragma unroll
        for (int e = threadIdx.x; e < BM * (3 * C / 8); e += NTB) {
            const int r = e / (3 * C / 8), c = (e % (3 * C / 8)) * 8;
            *reinterpret_cast<uint4*>(tq + r * LDQ + c) = *reinterpret_cast<const uint4*>(p.t_dqkv + (int64_t)min(r0 + r, p.R - 1) * (3 * C) + c);
        }
        __syncthreads();
        constexpr int PERG = (C / 16 + NWB - 1) / NWB;           // column groups of C per wave (2 at C = 256)
        f32x4 keep[PERG][MT];
        g_t.run(tq, p.t_wqt, nobias, [&](int g, const f32x4 (&acc)[MT], float) {
#pragma unroll
            for (int i = 0; i < PERG; ++i)
                if (g == wave + NWB * i) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) keep[i][t] = acc[t];
                }
        });
        __syncthreads();                                 // (the dqkv tile has been multiplied by every wave)
#pragma unroll
        for (int i = 0; i < PERG; ++i) {
            const int g = wave + NWB * i;
            if (g < C / 16) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int v = 0; v < 4; ++v) dzb[(16 * t + 4 * q + v) * LDX + 16 * g + j] = keep[i][t][v];
            }
        }
    }
    WideGemm<BM, FC, C, LDA, C / 32, NWB> g_2a;
    g_2a.issue(p.w2t, nobias, 0, 0);
    __syncthreads();                                     // (also: the sums above are visible to the waves that read their rows below)

    // One norm backwards over the block's rows, a wave per row (rows wave, wave + 8, ...: everything a row needs is requested up
    // front) and FOUR consecutive columns per lane (16- / 8-byte pieces: see the forward kernel's norms), ln_bwd_rows' arithmetic:
    //     t = rstd (g - mean(g) - xh mean(g xh)) [+ res],  y = dropout'(t)
    //   d: the incoming gradient -- global f32 [R][C] (+ the dz tile: add_tile), or (from_tile) the dz tile, rounded to bf16 as the separate launch's tensor was
    //   res_g: added behind the norm (null: nothing);  dx_g <- t (f32);  gb, dy_g <- y (bf16);  column sums of d xh, d, y -> red
    constexpr int LPR = C / 4;
    const bool lact = lane < LPR;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ln_bwd = [&](auto from_tile, const float* __restrict__ d_g, const float* __restrict__ xpre, const float* __restrict__ g_mean,
                      const float* __restrict__ g_rstd, const float* __restrict__ wp, const float* res_g, float* dx_g,
                      uint16_t* __restrict__ dy_g, const int which, const bool add_tile) {
        const float4 w4 = lact ? *reinterpret_cast<const float4*>(wp + 4 * lane) : z4;
        float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f}, ay[4] = {0.f, 0.f, 0.f, 0.f};
        float4 xr[NR], dr[NR], rr[NR];
        float mu[NR], rs[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = wave + NWB * i;
            const int64_t row = min(r0 + r, p.R - 1);
            mu[i] = g_mean[row];
            rs[i] = g_rstd[row];
            xr[i] = lact ? *reinterpret_cast<const float4*>(xpre + row * C + 4 * lane) : z4;
            if constexpr (!decltype(from_tile)::value) {
                dr[i] = lact ? *reinterpret_cast<const float4*>(d_g + row * C + 4 * lane) : z4;
                if (add_tile) {                          // (wave-uniform: the hosted tail's product, f32, from the dz tile)
                    const float4 t = lact ? *reinterpret_cast<const float4*>(dzb + r * LDX + 4 * lane) : z4;
                    dr[i].x += t.x; dr[i].y += t.y; dr[i].z += t.z; dr[i].w += t.w;
                }
            } else {
                const float4 t = lact ? *reinterpret_cast<const float4*>(dzb + r * LDX + 4 * lane) : z4;
                dr[i] = make_float4(bf16_round(t.x), bf16_round(t.y), bf16_round(t.z), bf16_round(t.w));
            }
            rr[i] = (res_g && lact) ? *reinterpret_cast<const float4*>(res_g + row * C + 4 * lane) : z4;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = wave + NWB * i;
            const int64_t row = min(r0 + r, p.R - 1);
            const bool on = r0 + r < p.R && lact;
            const float dv[4] = {dr[i].x, dr[i].y, dr[i].z, dr[i].w}, xv[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w}, rv[4] = {rr[i].x, rr[i].y, rr[i].z, rr[i].w};
            float d[4], xh[4], gg[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = on ? dv[k] : 0.f;
                xh[k] = on ? (xv[k] - mu[i]) * rs[i] : 0.f;
                gg[k] = d[k] * wv[k];
                ag[k] += d[k] * xh[k];
                ab[k] += d[k];
                s1 += gg[k];
                s2 += gg[k] * xh[k];
            }
            s1 = wave64_sum(s1) * (1.f / C);
            s2 = wave64_sum(s2) * (1.f / C);
            float t[4], yv[4];
            uint32_t hw[2] = {0u, 0u};
            if (p.thr) {                                   // (columns 2 m and 2 m + 1 share a hash word: dropout_bits16)
                hw[0] = dropout_bits16(seed, s_rowh[which][r], (uint32_t)(4 * lane)) | (dropout_bits16(seed, s_rowh[which][r], (uint32_t)(4 * lane + 1)) << 16);
                hw[1] = dropout_bits16(seed, s_rowh[which][r], (uint32_t)(4 * lane + 2)) | (dropout_bits16(seed, s_rowh[which][r], (uint32_t)(4 * lane + 3)) << 16);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k] = on ? rs[i] * (gg[k] - s1 - xh[k] * s2) + rv[k] : 0.f;
                yv[k] = t[k];
                if (p.thr) yv[k] = ((hw[k >> 1] >> (16 * (k & 1))) & 0xffffu) >= p.thr ? t[k] * p.inv_keep : 0.f;
                ay[k] += yv[k];
            }
            const uint2 yb = make_uint2(bf16_bits(yv[0]) | ((uint32_t)bf16_bits(yv[1]) << 16), bf16_bits(yv[2]) | ((uint32_t)bf16_bits(yv[3]) << 16));
            if (lact) {
                *reinterpret_cast<uint2*>(gb + r * LDA + 4 * lane) = yb;
                if (on) {
                    *reinterpret_cast<float4*>(dx_g + row * C + 4 * lane) = make_float4(t[0], t[1], t[2], t[3]);
                    *reinterpret_cast<uint2*>(dy_g + row * C + 4 * lane) = yb;
                }
            }
        }
        if (lact) {
            *reinterpret_cast<float4*>(red + (0 * NWB + wave) * C + 4 * lane) = make_float4(ag[0], ag[1], ag[2], ag[3]);
            *reinterpret_cast<float4*>(red + (1 * NWB + wave) * C + 4 * lane) = make_float4(ab[0], ab[1], ab[2], ab[3]);
            *reinterpret_cast<float4*>(red + (2 * NWB + wave) * C + 4 * lane) = make_float4(ay[0], ay[1], ay[2], ay[3]);
        }
    };
    auto flush = [&](float* g0, float* g1, float* g2) {
        for (int e = threadIdx.x; e < 3 * C; e += NTB) {
            const int which = e / C, c = e % C;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWB; ++w) s += red[(which * NWB + w) * C + c];
            float* dst = which == 0 ? g0 : (which == 1 ? g1 : g2);
            if (dst) atomicAdd(dst + c, s);
        }
    };
    // ---- dx2 = ffn_norm2'(dout) -> global (in the dx1 buffer);  df = dropout'(dx2)
    ln_bwd(std::false_type{}, p.dout, p.x2, p.mean2, p.rstd2, p.nxw, nullptr, p.dx1, p.df, 1, hosted);
    __syncthreads();
    STAMP(1);
    flush(p.dnxw, p.dnxb, p.db2);
    __syncthreads();
    STAMP(2);
    // ---- the FFN backwards, in chunks
    auto stage_u = [&](auto nc_c, const int c0) {
        constexpr int NC = decltype(nc_c)::value;
#pragma unroll
        for (int e = threadIdx.x; e < BM * (NC / 8); e += NTB) {
            const int r = e / (NC / 8), c = (e % (NC / 8)) * 8;
            *reinterpret_cast<uint4*>(tb + r * LDC + c) = *reinterpret_cast<const uint4*>(p.u + (int64_t)min(r0 + r, p.R - 1) * F + c0 + c);
        }
    };
    auto chunk = [&](auto nc_c, const int c0, auto& g2) {
        constexpr int NC = decltype(nc_c)::value;
        // du_c = (df W2[:, c]) gelu'(u_c), in place over u_c
        g2.run(gb, p.w2t, nobias, [&](int g, const f32x4 (&acc)[MT], float) {
            const int col = 16 * g + j;
            float uv[MT][4];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) uv[t][v] = bf16_val(tb[(16 * t + 4 * q + v) * LDC + col]);
            float cs = 0.f;                              // b1's gradient: the column sum of the ROUNDED du (rows >= R: df is zero there)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const uint16_t db = bf16_bits(acc[t][v] * gelu_grad_f(uv[t][v]));
                    tb[(16 * t + 4 * q + v) * LDC + col] = db;
                    cs += bf16_val(db);
                }
            if (p.db1) {
                cs += __shfl_xor(cs, 16, 64);
                cs += __shfl_xor(cs, 32, 64);
                if (q == 0) atomicAdd(p.db1 + c0 + col, cs);
            }
        }, c0 / 16, 0);
        WideGemm<BM, C, NC, LDC, F / 32, NWB> g1;
        g1.issue(p.w1t, nobias, 0, c0 / 32);
        __syncthreads();
        if (c0 == 0) STAMP(4);
        for (int e = threadIdx.x; e < BM * (NC / 8); e += NTB) {
            const int r = e / (NC / 8), c = (e % (NC / 8)) * 8;
            if (r0 + r < p.R) *reinterpret_cast<uint4*>(p.du + (int64_t)(r0 + r) * F + c0 + c) = *reinterpret_cast<const uint4*>(tb + r * LDC + c);
        }
        // dz (+)= du_c W1[c, :]
        g1.run(tb, p.w1t, nobias, [&](int g, const f32x4 (&acc)[MT], float) {
            const int col = 16 * g + j;
            float fv[MT][4];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) fv[t][v] = c0 ? dzb[(16 * t + 4 * q + v) * LDX + col] : 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) dzb[(16 * t + 4 * q + v) * LDX + col] = fv[t][v] + acc[t][v];
        }, 0, c0 / 32);
    };
    stage_u(std::integral_constant<int, FC>{}, 0);
    __syncthreads();
    STAMP(3);
    chunk(std::integral_constant<int, FC>{}, 0, g_2a);
    WideGemm<BM, FC, C, LDA, C / 32, NWB> g_2b;
    g_2b.issue(p.w2t, nobias, FC / 16, 0);
    __syncthreads();                                     // (the chunk's du has been multiplied and stored: its tile may be overwritten)
    STAMP(5);
    stage_u(std::integral_constant<int, FC>{}, FC);
    __syncthreads();
    chunk(std::integral_constant<int, FC>{}, FC, g_2b);
    WideGemm<BM, FL, C, LDA, C / 32, NWB> g_2c;
    g_2c.issue(p.w2t, nobias, 2 * FC / 16, 0);
    __syncthreads();
    STAMP(6);
    stage_u(std::integral_constant<int, FL>{}, 2 * FC);
    __syncthreads();
    chunk(std::integral_constant<int, FL>{}, 2 * FC, g_2c);
    WideGemm<BM, C, C, LDA, C / 32, NWB> g_o;
    g_o.issue(p.wot, nobias);
    __syncthreads();
    STAMP(7);
    // ---- dx1 = dx2 + ffn_norm1'(dz);  dy = dropout'(dx1)
    ln_bwd(std::true_type{}, nullptr, p.x1, p.mean1, p.rstd1, p.n1w, p.dx1, p.dx1, p.dy, 0, false);
    __syncthreads();
    STAMP(8);
    flush(p.dn1w, p.dn1b, p.dbo);
    __syncthreads();
    STAMP(9);
    // ---- da = dy Wo
    g_o.run(gb, p.wot, nobias, [&](int g, const f32x4 (&acc)[MT], float) {
        const int col = 16 * g + j;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) tb[(16 * t + 4 * q + v) * LDC + col] = bf16_bits(acc[t][v]);
    });
    __syncthreads();
    STAMP(10);
    store_rows<BM, C, LDC, NTB>(tb, p.da, r0, p.R);
    STAMP(11);
    STAMP_DUMP();
}

template <int C, int F>
int launch_bwd_big(const ChainBwdParams& p, hipStream_t st) {
    constexpr int BM = 64, LDA = C + 8, LDX = C + 4, LDC = 384 + 8;
    constexpr size_t lds = BM * LDX * 4 + BM * LDA * 2 + BM * LDC * 2;
    static_assert(lds <= 152 * 1024, "LDS plan");
    int rc = (int)hipFuncSetAttribute((const void*)layer_chain_bwd_big_kernel<C, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    hipLaunchKernelGGL((layer_chain_bwd_big_kernel<C, F>), dim3((p.R + BM - 1) / BM), dim3(BIG_NW * 64), lds, st, p);
    return (int)hipGetLastError();
}

template <int BM, int C, int F>
int launch_bwd(const ChainBwdParams& p, hipStream_t st) {
    constexpr size_t lds = 2 * BM * (C + 4) * 4 + 3 * NW * C * 4 + BM * (C + 8) * 2 + 2 * BM * (F + 8) * 2;
    static_assert(lds <= 152 * 1024 && lds >= mobgt_wgrad::wgrad_lds_floats<NW>() * sizeof(float), "LDS plan");
    int rc = (int)hipFuncSetAttribute((const void*)layer_chain_bwd_kernel<BM, C, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    hipLaunchKernelGGL((layer_chain_bwd_kernel<BM, C, F>), dim3(p.n_chain + (p.n_wg ? p.wg_first[p.n_wg] : 0)), dim3(NT), lds, st, p);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mobgt_pack_mfma_b(int n, const void* const* src, void* const* dst, const int* N, const int* K,
                                 const int* transposed, void* stream) {
    if (n <= 0) return 0;
    PackJobs jobs = {};
    int blocks = 0;
    const int rc = mobgt_pack::fill_jobs(jobs, n, src, dst, N, K, transposed, &blocks);
    if (rc) return rc;
    hipLaunchKernelGGL(pack_mfma_b_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, jobs, n, blocks);
    return (int)hipGetLastError();
}

#ifdef CH_DEBUG
extern "C" int mobgt_chain_debug_buffer(int* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_dbg), &buf, sizeof(buf)); }
#endif

extern "C" int mobgt_layer_chain_fwd(const void* a, const float* x, const void* wo, const void* bo, const float* n1w,
                                     const float* n1b, const void* w1, const void* b1, const void* w2, const void* b2,
                                     const float* nxw, const float* nxb, const void* wq_next, const void* bq_next, float* x1,
                                     void* z, void* u, void* h, float* x2, float* out, void* out_a, void* qkv_next,
                                     float* mean1, float* rstd1, float* mean2, float* rstd2, int64_t R, int C, int F,
                                     float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt1, uint32_t salt2,
                                     void* ws, void* stream) {
    if (R <= 0) return 0;
    if (R > 0x7fffffff) return MOBGT_EBADDIM;
    if ((uintptr_t)ws & 15) return MOBGT_EALIGN;
    if (((uintptr_t)a | (uintptr_t)x | (uintptr_t)wo | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)wq_next | (uintptr_t)h | (uintptr_t)u | (uintptr_t)qkv_next) & 15)
        return MOBGT_EALIGN;
    if ((wq_next == nullptr) != (qkv_next == nullptr)) return MOBGT_EBADDIM;
    if (!nxw && wq_next) return MOBGT_EBADDIM;          // (the next projection multiplies the second norm's output)
    ChainParams p = {};
    typedef const uint16_t* cu;
    p.a = (cu)a; p.x = x; p.wo = (cu)wo; p.bo = (cu)bo; p.w1 = (cu)w1; p.b1 = (cu)b1; p.w2 = (cu)w2; p.b2 = (cu)b2;
    p.wq = (cu)wq_next; p.bq = (cu)bq_next; p.n1w = n1w; p.n1b = n1b; p.nxw = nxw; p.nxb = nxb;
    p.x1 = x1; p.x2 = x2; p.out = out; p.z = (uint16_t*)z; p.u = (uint16_t*)u; p.h = (uint16_t*)h; p.out_a = (uint16_t*)out_a;
    p.qkv = (uint16_t*)qkv_next; p.mean1 = mean1; p.rstd1 = rstd1; p.mean2 = mean2; p.rstd2 = rstd2; p.R = (int)R;
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt1 = salt1; p.salt2 = salt2;
    hipStream_t st = (hipStream_t)stream;
    if (R > CHAIN_BIG_ROWS) {                      // long batches: 64 rows per workgroup (layer_chain_fwd_big_kernel)
        if (C == 128 && F == 1024) return launch_big<128, 1024>(p, st);
        if (C == 192 && F == 1024) return launch_big<192, 1024>(p, st);
        if (C == 256 && F == 1024) return launch_big<256, 1024>(p, st);
        return MOBGT_EBADDIM;
    }
    const int ncl = pick_ncl(R, ws);
    if (ncl > 1) {
        p.ws_gen = reinterpret_cast<uint32_t*>(ws);
        p.ws_ll = reinterpret_cast<uint64_t*>(reinterpret_cast<uint32_t*>(ws) + WS_GEN_INTS);
        if (C == 128 && F == 1024) return ncl == 4 ? launch_cl<16, 128, 1024, 4>(p, st) : launch_cl<16, 128, 1024, 2>(p, st);
        if (C == 192 && F == 1024) return ncl == 4 ? launch_cl<16, 192, 1024, 4>(p, st) : launch_cl<16, 192, 1024, 2>(p, st);
        if (C == 256 && F == 1024) return ncl == 4 ? launch_cl<16, 256, 1024, 4>(p, st) : launch_cl<16, 256, 1024, 2>(p, st);
        return MOBGT_EBADDIM;
    }
    if (C == 128 && F == 1024) return launch<16, 128, 1024>(p, st);
    if (C == 192 && F == 1024) return launch<16, 192, 1024>(p, st);
    if (C == 256 && F == 1024) return launch<16, 256, 1024>(p, st);
    return MOBGT_EBADDIM;                    // the instantiated widths: MobGT's hidden 128 (model.py) / 192 / 256 (+ 64 of embeddings), ffn 1024
}

static int chain_bwd_impl(int pre_ln, const float* dout, const float* x2, const float* x1, const void* u, const float* mean1,
                                     const float* rstd1, const float* mean2, const float* rstd2, const float* n1w,
                                     const float* nxw, const void* w2t, const void* w1t, const void* wot, void* df, void* du,
                                     void* dy, void* da, float* dx1, float* dnxw, float* dnxb, float* db2, float* dn1w,
                                     float* dn1b, float* dbo, int64_t R, int C, int F, float dropout_p, uint64_t seed,
                                     const uint64_t* seed_dev, uint32_t salt1, uint32_t salt2, const void* tail_dqkv,
                                     const void* tail_wqkv_t, int n_wg, const void* const* wg_g, const int64_t* wg_ldg,
                                     const void* const* wg_x, const int64_t* wg_ldx, float* const* wg_dw, const int64_t* wg_ldw,
                                     float* const* wg_db, const int* wg_M, const int* wg_N, void* ws, void* stream) {
    if (R <= 0) return 0;
    if (!pre_ln && !nxw) return MOBGT_EBADDIM;              // (post-LN layers always have their second norm)
    if (pre_ln && !nxw && tail_dqkv) return MOBGT_EBADDIM;  // (a hosted tail goes back through the successor's norm)
    if ((uintptr_t)ws & 15) return MOBGT_EALIGN;
    if (R > 0x7fffffff || n_wg < 0 || n_wg > 4) return MOBGT_EBADDIM;
    if (((uintptr_t)u | (uintptr_t)w2t | (uintptr_t)w1t | (uintptr_t)wot | (uintptr_t)du | (uintptr_t)da | (uintptr_t)tail_dqkv |
         (uintptr_t)tail_wqkv_t | (uintptr_t)dout) & 15) return MOBGT_EALIGN;
    if ((tail_dqkv == nullptr) != (tail_wqkv_t == nullptr)) return MOBGT_EBADDIM;
    ChainBwdParams p = {};
    typedef const uint16_t* cu;
    p.dout = dout; p.x2 = x2; p.x1 = x1; p.u = (cu)u; p.mean1 = mean1; p.rstd1 = rstd1; p.mean2 = mean2; p.rstd2 = rstd2;
    p.n1w = n1w; p.nxw = nxw; p.w2t = (cu)w2t; p.w1t = (cu)w1t; p.wot = (cu)wot;
    p.df = (uint16_t*)df; p.du = (uint16_t*)du; p.dy = (uint16_t*)dy; p.da = (uint16_t*)da; p.dx1 = dx1;
    p.dnxw = dnxw; p.dnxb = dnxb; p.db2 = db2; p.dn1w = dn1w; p.dn1b = dn1b; p.dbo = dbo; p.R = (int)R;
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt1 = salt1; p.salt2 = salt2;
    p.t_dqkv = (cu)tail_dqkv; p.t_wqt = (cu)tail_wqkv_t;
    p.pre_ln = pre_ln ? 1 : 0;
    if (R > CHAIN_BIG_ROWS) {                // long batches: 64 rows per workgroup; no guests (see layer_chain_bwd_big_kernel)
        if (pre_ln || tail_dqkv || n_wg || !nxw) return MOBGT_EBADDIM;
        if (C == 128 && F == 1024) return launch_bwd_big<128, 1024>(p, (hipStream_t)stream);
        if (C == 192 && F == 1024) return launch_bwd_big<192, 1024>(p, (hipStream_t)stream);
        if (C == 256 && F == 1024) return launch_bwd_big<256, 1024>(p, (hipStream_t)stream);
        return MOBGT_EBADDIM;
    }
    p.n_chain = (int)((R + 15) / 16);
    p.n_wg = n_wg;
    int total = 0;
    for (int i = 0; i < n_wg; ++i) {             // the passengers: bf16 operands, R rows, 12 waves per workgroup like the chain's
        const int rc = mobgt_wgrad::fill_problem(p.wg[i], wg_g[i], wg_ldg[i], wg_x[i], wg_ldx[i], wg_dw[i], wg_ldw[i],
                                                 wg_db ? wg_db[i] : nullptr, R, wg_M[i], wg_N[i], 64, &p.wg_tiles[i],
                                                 &p.wg_splits[i], 0, NW);
        if (rc) return rc;
        p.wg_first[i] = total;
        total += p.wg_tiles[i] * p.wg_splits[i];
    }
    for (int i = n_wg; i <= 4; ++i) p.wg_first[i] = total;
    hipStream_t st = (hipStream_t)stream;
    const int ncl = pick_ncl(R, ws);
    if (ncl > 1) {
        p.ws_gen = reinterpret_cast<uint32_t*>(ws);
        p.ws_ll = reinterpret_cast<uint64_t*>(reinterpret_cast<uint32_t*>(ws) + WS_GEN_INTS);
        if (C == 128 && F == 1024) return ncl == 4 ? launch_bwd_cl<16, 128, 1024, 4>(p, st) : launch_bwd_cl<16, 128, 1024, 2>(p, st);
        if (C == 192 && F == 1024) return ncl == 4 ? launch_bwd_cl<16, 192, 1024, 4>(p, st) : launch_bwd_cl<16, 192, 1024, 2>(p, st);
        if (C == 256 && F == 1024) return ncl == 4 ? launch_bwd_cl<16, 256, 1024, 4>(p, st) : launch_bwd_cl<16, 256, 1024, 2>(p, st);
        return MOBGT_EBADDIM;
    }
    if (C == 128 && F == 1024) return launch_bwd<16, 128, 1024>(p, st);
    if (C == 192 && F == 1024) return launch_bwd<16, 192, 1024>(p, st);
    if (C == 256 && F == 1024) return launch_bwd<16, 256, 1024>(p, st);
    return MOBGT_EBADDIM;
}

#define CHAIN_BWD_ARGS dout, x2, x1, u, mean1, rstd1, mean2, rstd2, n1w, nxw, w2t, w1t, wot, df, du, dy, da, dx1, dnxw, dnxb, db2, dn1w, \
                       dn1b, dbo, R, C, F, dropout_p, seed, seed_dev, salt1, salt2, tail_dqkv, tail_wqkv_t, n_wg, wg_g, wg_ldg, wg_x,   \
                       wg_ldx, wg_dw, wg_ldw, wg_db, wg_M, wg_N, ws, stream
extern "C" int mobgt_layer_chain_bwd(const float* dout, const float* x2, const float* x1, const void* u, const float* mean1,
                                     const float* rstd1, const float* mean2, const float* rstd2, const float* n1w,
                                     const float* nxw, const void* w2t, const void* w1t, const void* wot, void* df, void* du,
                                     void* dy, void* da, float* dx1, float* dnxw, float* dnxb, float* db2, float* dn1w,
                                     float* dn1b, float* dbo, int64_t R, int C, int F, float dropout_p, uint64_t seed,
                                     const uint64_t* seed_dev, uint32_t salt1, uint32_t salt2, const void* tail_dqkv,
                                     const void* tail_wqkv_t, int n_wg, const void* const* wg_g, const int64_t* wg_ldg,
                                     const void* const* wg_x, const int64_t* wg_ldx, float* const* wg_dw, const int64_t* wg_ldw,
                                     float* const* wg_db, const int* wg_M, const int* wg_N, void* ws, void* stream) {
    return chain_bwd_impl(0, CHAIN_BWD_ARGS);
}
// The same launch for PRE-LN layers (graphormer/model.py:479-489; see ChainBwdParams::pre_ln): nxw / mean2 / rstd2 / x2 describe the
// SUCCESSOR's self_attention_norm (null: no successor in the chain), dnxw / dnxb receive that norm's gradients.
extern "C" int mobgt_layer_chain_bwd_preln(const float* dout, const float* x2, const float* x1, const void* u, const float* mean1,
                                     const float* rstd1, const float* mean2, const float* rstd2, const float* n1w,
                                     const float* nxw, const void* w2t, const void* w1t, const void* wot, void* df, void* du,
                                     void* dy, void* da, float* dx1, float* dnxw, float* dnxb, float* db2, float* dn1w,
                                     float* dn1b, float* dbo, int64_t R, int C, int F, float dropout_p, uint64_t seed,
                                     const uint64_t* seed_dev, uint32_t salt1, uint32_t salt2, const void* tail_dqkv,
                                     const void* tail_wqkv_t, int n_wg, const void* const* wg_g, const int64_t* wg_ldg,
                                     const void* const* wg_x, const int64_t* wg_ldx, float* const* wg_dw, const int64_t* wg_ldw,
                                     float* const* wg_db, const int* wg_M, const int* wg_N, void* ws, void* stream) {
    return chain_bwd_impl(1, CHAIN_BWD_ARGS);
}

/* The 64-row backward chain on its own entry point (R of any size): mobgt_layer_chain_bwd's arguments without the weight-gradient
 * passengers and the workspace, plus db1 [F] f32 (accumulated; may be null). */
extern "C" int mobgt_layer_chain_bwd_big(const float* dout, const float* x2, const float* x1, const void* u, const float* mean1,
                                         const float* rstd1, const float* mean2, const float* rstd2, const float* n1w,
                                         const float* nxw, const void* w2t, const void* w1t, const void* wot, void* df, void* du,
                                         void* dy, void* da, float* dx1, float* dnxw, float* dnxb, float* db2, float* dn1w,
                                         float* dn1b, float* dbo, float* db1, int64_t R, int C, int F, float dropout_p, uint64_t seed,
                                         const uint64_t* seed_dev, uint32_t salt1, uint32_t salt2, const void* tail_dqkv,
                                         const void* tail_wqkv_t, void* stream) {
    if (R <= 0) return 0;
    if (!nxw || R > 0x7fffffff || (tail_dqkv == nullptr) != (tail_wqkv_t == nullptr)) return MOBGT_EBADDIM;
    if (((uintptr_t)tail_dqkv | (uintptr_t)tail_wqkv_t) & 15) return MOBGT_EALIGN;
    if (((uintptr_t)u | (uintptr_t)w2t | (uintptr_t)w1t | (uintptr_t)wot | (uintptr_t)du | (uintptr_t)da | (uintptr_t)dout | (uintptr_t)x1 |
         (uintptr_t)x2 | (uintptr_t)dx1 | (uintptr_t)df | (uintptr_t)dy) & 15) return MOBGT_EALIGN;
    ChainBwdParams p = {};
    typedef const uint16_t* cu;
    p.dout = dout; p.x2 = x2; p.x1 = x1; p.u = (cu)u; p.mean1 = mean1; p.rstd1 = rstd1; p.mean2 = mean2; p.rstd2 = rstd2;
    p.n1w = n1w; p.nxw = nxw; p.w2t = (cu)w2t; p.w1t = (cu)w1t; p.wot = (cu)wot;
    p.df = (uint16_t*)df; p.du = (uint16_t*)du; p.dy = (uint16_t*)dy; p.da = (uint16_t*)da; p.dx1 = dx1;
    p.dnxw = dnxw; p.dnxb = dnxb; p.db2 = db2; p.dn1w = dn1w; p.dn1b = dn1b; p.dbo = dbo; p.db1 = db1; p.R = (int)R;
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt1 = salt1; p.salt2 = salt2;
    p.t_dqkv = (cu)tail_dqkv; p.t_wqt = (cu)tail_wqkv_t;
    if (C == 128 && F == 1024) return launch_bwd_big<128, 1024>(p, (hipStream_t)stream);
    if (C == 192 && F == 1024) return launch_bwd_big<192, 1024>(p, (hipStream_t)stream);
    if (C == 256 && F == 1024) return launch_bwd_big<256, 1024>(p, (hipStream_t)stream);
    return MOBGT_EBADDIM;
}

extern "C" int64_t mobgt_chain_ws_bytes(void) {
    return (int64_t)(WS_GEN_INTS * sizeof(uint32_t) + WS_LL_WORDS * sizeof(uint64_t));
}
extern "C" int64_t mobgt_chain_ws_fault_offset(void) { return (int64_t)WS_FAULT * (int64_t)sizeof(uint32_t); }
extern "C" int64_t mobgt_chain_ws_limit_offset(void) { return (int64_t)WS_LIMIT * (int64_t)sizeof(uint32_t); }

extern "C" int mobgt_assemble_tokens_qkv(const float* nf, const float* real, const float* add, const float* token,
                                         const float* pe0, float* out, void* out_bf16, const void* wqkv_packed, const void* bqkv,
                                         void* qkv, int G, int N, int C, float p_pos, float p_in, uint64_t seed,
                                         const uint64_t* seed_dev, uint32_t salt_nf, uint32_t salt_tok, uint32_t salt_in,
                                         void* stream) {
    if (G <= 0 || N < 0 || !out_bf16 || !wqkv_packed || !bqkv || !qkv) return MOBGT_EBADDIM;
    if (((uintptr_t)wqkv_packed | (uintptr_t)qkv) & 15) return MOBGT_EALIGN;
    AsmQkvParams p = {};
    p.nf = nf; p.real = real; p.add = add; p.token = token; p.pe0 = pe0; p.out = out; p.out16 = (uint16_t*)out_bf16;
    p.wq = (const uint16_t*)wqkv_packed; p.bq = (const uint16_t*)bqkv; p.qkv = (uint16_t*)qkv; p.G = G; p.N = N;
    p.thr_pos = p_pos > 0.f ? dropout_threshold(p_pos) : 0u;
    p.thr_in = p_in > 0.f ? dropout_threshold(p_in) : 0u;
    p.keep_pos = p.thr_pos ? 1.f / (1.f - (float)p.thr_pos / 65536.f) : 1.f;
    p.keep_in = p.thr_in ? 1.f / (1.f - (float)p.thr_in / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt_nf = salt_nf; p.salt_tok = salt_tok; p.salt_in = salt_in;
    const int64_t rows = (int64_t)G * (N + 1);
    const dim3 grid((unsigned)((rows + 15) / 16)), block(NT);
    hipStream_t st = (hipStream_t)stream;
    if (C == 192) hipLaunchKernelGGL(assemble_qkv_kernel<192>, grid, block, 0, st, p);
    else if (C == 256) hipLaunchKernelGGL(assemble_qkv_kernel<256>, grid, block, 0, st, p);
    else return MOBGT_EBADDIM;
    return (int)hipGetLastError();
}

/* Round 4: mobgt_embed_gather_multi (forward) + FuseEmbeddings-2 + FuseEmbeddings-4 + mobgt_assemble_tokens_qkv as one launch
 * (see token_fwd_chain_kernel).  C = 192, W2 = 160 (the fq model at hidden_dim 128). */
extern "C" int mobgt_token_fwd_chain(int n, const float* const* tables, const void* const* idx, const int* width, const int* coff,
                                     const int* accum, const int* slot, int idx_dtype, float* pt, float* x4, float* add, float* nf,
                                     int W2, const float* w2, const float* b2, float slope2, const float* w4, const float* b4,
                                     float slope4, const float* real, const float* token, const float* pe0, float* out,
                                     void* out_bf16, const void* wqkv_packed, const void* bqkv, void* qkv, int G, int N, int C,
                                     float p_pos, float p_in, uint64_t seed, const uint64_t* seed_dev, uint32_t salt_nf,
                                     uint32_t salt_tok, uint32_t salt_in, void* stream) {
    if (G <= 0 || N < 0 || !out_bf16 || !wqkv_packed || !bqkv || !qkv || !pt || !x4 || !add || !nf || !w2 || !b2 || !w4 || !b4) return MOBGT_EBADDIM;
    if (C != 192 || W2 != 160 || n < 3 || n > 7 + TokGather::MAXFOLD) return MOBGT_EBADDIM;
    if (idx_dtype != MOBGT_I64 && idx_dtype != MOBGT_I32) return MOBGT_EDTYPE;
    if (((uintptr_t)wqkv_packed | (uintptr_t)qkv | (uintptr_t)pt | (uintptr_t)x4 | (uintptr_t)add | (uintptr_t)nf | (uintptr_t)w2 | (uintptr_t)w4) & 15)
        return MOBGT_EALIGN;
    TokFwdParams p = {};
    // the job list, sorted by destination (what the kernel's gather stage is laid out for; anything else: MOBGT_EBADDIM)
    int last0 = -1;
    for (int t = 0; t < n; ++t) {
        if (!tables[t] || ((uintptr_t)tables[t] & 15)) return MOBGT_EALIGN;
        if (width[t] <= 0 || width[t] > 256 || (width[t] & 3) || (coff[t] & 3)) return MOBGT_EBADDIM;
        if (accum[t] == 2) {                          // folded into the FIRST pt table
            if (last0 != 0 || p.g.n0 != 1 || p.g.n_fold >= TokGather::MAXFOLD || width[t] != p.g.s0_w[0]) return MOBGT_EBADDIM;
            p.g.fold[p.g.n_fold++] = tables[t];
            continue;
        }
        if (slot[t] == 0) {
            if (p.g.n0 >= 2 || accum[t] || coff[t] + width[t] > W2) return MOBGT_EBADDIM;
            p.g.s0_tab[p.g.n0] = tables[t]; p.g.s0_idx[p.g.n0] = idx[t]; p.g.s0_w[p.g.n0] = width[t]; p.g.s0_coff[p.g.n0] = coff[t];
            last0 = p.g.n0++;
        } else if (slot[t] == 1) {
            if (p.g.s1_tab || accum[t] || coff[t] < W2 || coff[t] + width[t] > C) return MOBGT_EBADDIM;
            p.g.s1_tab = tables[t]; p.g.s1_idx = idx[t]; p.g.s1_w = width[t]; p.g.s1_coff = coff[t];
            last0 = -1;
        } else if (slot[t] == 2) {
            if (p.g.n2 >= 4 || width[t] != C || coff[t] != 0 || (accum[t] != 0) != (p.g.n2 > 0)) return MOBGT_EBADDIM;
            p.g.s2_tab[p.g.n2] = tables[t]; p.g.s2_idx[p.g.n2] = idx[t]; ++p.g.n2;
            last0 = -1;
        } else return MOBGT_EBADDIM;
    }
    if (p.g.n0 < 1 || !p.g.s1_tab || p.g.n2 < 1) return MOBGT_EBADDIM;
    for (int q = p.g.n0; q < 2; ++q) { p.g.s0_tab[q] = p.g.s0_tab[0]; p.g.s0_idx[q] = p.g.s0_idx[0]; p.g.s0_w[q] = 4; p.g.s0_coff[q] = 0; }
    for (int q = p.g.n2; q < 4; ++q) { p.g.s2_tab[q] = p.g.s2_tab[0]; p.g.s2_idx[q] = p.g.s2_idx[0]; }
    p.g.idx64 = idx_dtype == MOBGT_I64;
    p.pt = pt; p.x4 = x4; p.w2 = w2; p.b2 = b2; p.w4 = w4; p.b4 = b4; p.slope2 = slope2; p.slope4 = slope4;
    AsmQkvParams& a = p.a;
    a.nf = nf; a.real = real; a.add = add; a.token = token; a.pe0 = pe0; a.out = out; a.out16 = (uint16_t*)out_bf16;
    a.wq = (const uint16_t*)wqkv_packed; a.bq = (const uint16_t*)bqkv; a.qkv = (uint16_t*)qkv; a.G = G; a.N = N;
    a.thr_pos = p_pos > 0.f ? dropout_threshold(p_pos) : 0u;
    a.thr_in = p_in > 0.f ? dropout_threshold(p_in) : 0u;
    a.keep_pos = a.thr_pos ? 1.f / (1.f - (float)a.thr_pos / 65536.f) : 1.f;
    a.keep_in = a.thr_in ? 1.f / (1.f - (float)a.thr_in / 65536.f) : 1.f;
    a.seed = seed; a.seed_dev = seed_dev; a.salt_nf = salt_nf; a.salt_tok = salt_tok; a.salt_in = salt_in;
    const int64_t rows = (int64_t)G * (N + 1);
    hipLaunchKernelGGL((token_fwd_chain_kernel<192, 160>), dim3((unsigned)((rows + 15) / 16)), dim3(NT), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
