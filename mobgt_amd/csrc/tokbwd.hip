// Backward of the encoder input from d(tokens) down to the gathered rows, one launch (model_fqandtoyo.py:1264-1298, 1338-1347;
// FuseEmbeddings :444-456):
//
//     d        = input_dropout'( pos_dropout'( d(out)[g, 1 + n, :] ) )          mobgt_assemble_tokens_bwd
//     d_add    = d;    d_nf = d * real[g, n]
//     g4       = d_nf * LeakyReLU'(nf)                                           FuseEmbeddings-4's activation (from its OUTPUT nf)
//     dx4      = g4 W4                    [.., :W2] = d(f2), [.., W2:] = d(category rows)
//     g2       = dx4[:, :W2] * LeakyReLU'(f2)
//     d_pt     = g2 W2                                                           FuseEmbeddings-2
//     d_token += column sums of the graph-token rows' d
//
// As launches these were assemble_tokens<true> (5 us), two small f32 GEMMs with a masked operand (10 + 9 us): three ramps and
// two round trips through global memory for 20 MFLOP on a few hundred rows.  Here a workgroup of 12 waves owns 16 node rows and
// walks the chain with its rows in LDS.  Products: full-f32 v_mfma_f32_16x16x4_f32, the weight ([out, in] row-major = the
// [K, N] operand of dX = g W) read as 16-byte runs ALONG a row -- 4 adjacent columns serve 4 MFMAs whose output column for lane j
// is n0 + 4 j + n -- so a wave instruction touches 4 rows x 256 contiguous bytes; 64-column blocks x K quarters over the 12
// waves, the quarters meet in an LDS tile (ds_add_f32).  d_nf and dx4 are written out because the two weight gradients
// (g4^T x4, g2^T pt -- leaves of the step's grouped weight-gradient launch) load them with the same activation masks.
#include "common.h"
#include "mobgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TNW = 12, TNT = TNW * 64, TBM = 16;

struct TokBwdParams {
    const float* dout;                       // [G, N+1, C]
    const float* real;                       // [G*N]
    const float* y4;                         // nf = FuseEmbeddings-4's output [G*N, C]
    const float* y2; int64_t ld2;            // f2 = FuseEmbeddings-2's output [G*N, W2], row stride ld2 (the leading columns of x4)
    const float *w4, *w2;                    // [C, C], [W2, W2] row-major (out, in)
    float *d_nf, *d_add;                     // [G*N, C]
    float* dx4; int64_t ldx4;                // [G*N, C]
    float* d_pt;                             // [G*N, W2]
    float* d_token;                          // [C], accumulated
    int G, N;
    float slope4, slope2;
    uint32_t thr_pos, thr_in;
    float keep_pos, keep_in;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt_nf, salt_tok, salt_in;
    int n_node_blocks;
};

__device__ __forceinline__ float leaky_grad(float y, float slope) { return y > 0.f ? 1.f : slope; }

#ifdef TB_STAMP
__device__ int* g_tb_dbg = nullptr;
#define TSTAMP_DECL int st_[8] = {}
#define TSTAMP(i) st_[i] = (int)wall_clock64()
#define TSTAMP_DUMP() do { if (g_tb_dbg && threadIdx.x == 0) for (int q_ = 0; q_ < 8; ++q_) g_tb_dbg[blockIdx.x * 8 + q_] = st_[q_]; } while (0)
#else
#define TSTAMP_DECL
#define TSTAMP(i)
#define TSTAMP_DUMP()
#endif

// out[ks] [TBM][N] (LDS partial tiles, one per K split ks < gemm_ks<N>(): the caller adds them up) = A [TBM][K] (LDS, row stride
// LDA) x rows [K range ks] of W [K][N] (global, row-major, row stride N).  Plain stores: ds_add_f32 into ONE tile measured 13 us
// per product (64 lanes on 8 banks, a read-modify-write each) against 3 us for the whole product without it.
template <int N> constexpr int gemm_ks() { return TNW / ((N + 63) / 64); }
template <int K, int N, int LDA, int LDO>
__device__ __forceinline__ void gemm_kn(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out) {
    constexpr int NB = (N + 63) / 64, KS = TNW / NB, NT16 = K / 16, TMAX = (NT16 + KS - 1) / KS;
    static_assert(K % 16 == 0 && N % 32 == 0 && KS >= 1, "shape");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    if (wave >= NB * KS) return;
    const int blk = wave / KS, ks = wave % KS;
    const int t_lo = ks * NT16 / KS, t_hi = (ks + 1) * NT16 / KS;
    const int n0 = 64 * blk;
    const bool col_ok = n0 + 4 * j < N;                       // (the last block of a width that is not a multiple of 64)
    const int cl = col_ok ? n0 + 4 * j : 0;
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bw[TMAX][4];
#pragma unroll
    for (int tt = 0; tt < TMAX; ++tt) {
        const int t = min(t_lo + tt, t_hi - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) bw[tt][i] = *reinterpret_cast<const f32x4*>(W + (int64_t)(16 * t + 4 * q + i) * N + cl);
    }
#pragma unroll
    for (int tt = 0; tt < TMAX; ++tt) {
        if (t_lo + tt < t_hi) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(A + j * LDA + 16 * (t_lo + tt) + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bw[tt][i][n], acc[n], 0, 0, 0);
        }
    }
    if (col_ok) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v) out[(ks * TBM + 4 * q + v) * LDO + n0 + 4 * j + n] = acc[n][v];
    }
}

template <int C, int W2>
__global__ __launch_bounds__(TNT) void token_bwd_chain_kernel(const TokBwdParams p) {
    constexpr int LDA = C + 4, LD2 = W2 + 4;
    __shared__ __attribute__((aligned(16))) float a4[TBM * LDA];            // g4 rows, later g2 rows (the A operands)
    constexpr int KS4 = gemm_ks<C>(), KS2 = gemm_ks<W2>();
    __shared__ __attribute__((aligned(16))) float t4[(KS4 > KS2 ? KS4 : KS2) * TBM * LDA];     // partial tiles of dx4, later of d_pt
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = p.N + 1;
    const uint64_t seed = (p.thr_pos || p.thr_in) ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    if ((int)blockIdx.x >= p.n_node_blocks) {
        // ---- the graph-token rows: d(token) (= d(pe[0])) += d, one wave per graph
        const int g = ((int)blockIdx.x - p.n_node_blocks) * TNW + wave;
        if (g >= p.G) return;
        const int64_t row = (int64_t)g * T;
        const uint32_t h1 = p.thr_pos ? dropout_row_hash(seed, (uint32_t)g ^ p.salt_tok) : 0u;
        const uint32_t h2 = p.thr_in ? dropout_row_hash(seed, (uint32_t)row ^ p.salt_in) : 0u;
        for (int c = lane; c < C; c += 64) {
            float scale = 1.f;
            if (p.thr_pos) scale = dropout_bits16(seed, h1, (uint32_t)c) >= p.thr_pos ? p.keep_pos : 0.f;
            if (p.thr_in) scale *= dropout_bits16(seed, h2, (uint32_t)c) >= p.thr_in ? p.keep_in : 0.f;
            const float d = p.dout[row * C + c] * scale;
            if (d != 0.f) atomicAdd(&p.d_token[c], d);
        }
        return;
    }
    const int64_t R = (int64_t)p.G * p.N;
    const int64_t j0 = (int64_t)blockIdx.x * TBM;
    TSTAMP_DECL;
    TSTAMP(0);
    // ---- token assembly backwards + FuseEmbeddings-4's activation derivative: TBM x C elements dealt out flat, every load of a
    //      thread requested before the first is used (a wave per row took two dependent passes: 3.6 us)
    constexpr int EPT = TBM * C / TNT;
    static_assert(TBM * C % TNT == 0, "flat split");
    float dv[EPT], yv[EPT], rl[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = threadIdx.x + k * TNT, r = e / C, c = e % C;
        const int64_t jc = min(j0 + r, R - 1);
        const int g = (int)(jc / p.N), n = (int)(jc - (int64_t)g * p.N);
        dv[k] = p.dout[((int64_t)g * T + n + 1) * C + c];
        yv[k] = p.y4[jc * C + c];
        rl[k] = p.real[jc];
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = threadIdx.x + k * TNT, r = e / C, c = e % C;
        const int64_t jn = j0 + r;
        const bool on = jn < R;
        const int64_t jc = on ? jn : R - 1;
        const int g = (int)(jc / p.N), n = (int)(jc - (int64_t)g * p.N);
        const int64_t row = (int64_t)g * T + n + 1;
        float scale = 1.f;
        if (p.thr_pos) scale = dropout_bits16(seed, dropout_row_hash(seed, (uint32_t)jc ^ p.salt_nf), (uint32_t)c) >= p.thr_pos ? p.keep_pos : 0.f;
        if (p.thr_in) scale *= dropout_bits16(seed, dropout_row_hash(seed, (uint32_t)row ^ p.salt_in), (uint32_t)c) >= p.thr_in ? p.keep_in : 0.f;
        const float d = on ? dv[k] * scale : 0.f;
        const float dnf = d * rl[k];
        a4[r * LDA + c] = dnf * leaky_grad(yv[k], p.slope4);
        if (on) {
            p.d_add[jn * C + c] = d;
            p.d_nf[jn * C + c] = dnf;
        }
    }
    __syncthreads();
    TSTAMP(1);
    // ---- dx4 = g4 W4
    gemm_kn<C, C, LDA, LDA>(a4, p.w4, t4);
    __syncthreads();
    TSTAMP(2);
    // dx4 out; g2 = dx4[:, :W2] * LeakyReLU'(f2) -> the next A operand
    for (int e = threadIdx.x; e < TBM * C; e += TNT) {
        const int r = e / C, c = e % C;
        const int64_t jn = j0 + r;
        float v = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS4; ++ks) v += t4[(ks * TBM + r) * LDA + c];
        if (jn < R) p.dx4[jn * p.ldx4 + c] = v;
        if (c < W2) a4[r * LD2 + c] = jn < R ? v * leaky_grad(p.y2[jn * p.ld2 + c], p.slope2) : 0.f;
    }
    __syncthreads();
    TSTAMP(3);
    // ---- d_pt = g2 W2
    gemm_kn<W2, W2, LD2, LD2>(a4, p.w2, t4);
    __syncthreads();
    TSTAMP(4);
    for (int e = threadIdx.x; e < TBM * W2; e += TNT) {
        const int r = e / W2, c = e % W2;
        const int64_t jn = j0 + r;
        float v = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) v += t4[(ks * TBM + r) * LD2 + c];
        if (jn < R) p.d_pt[jn * W2 + c] = v;
    }
    TSTAMP(5);
    TSTAMP_DUMP();
}

}  // namespace

#ifdef TB_STAMP
extern "C" int mobgt_tokbwd_debug_buffer(int* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tb_dbg), &buf, sizeof(buf)); }
#endif

extern "C" int mobgt_token_bwd_chain(const float* dout, const float* real, const float* y4, const float* y2, int64_t ld_y2,
                                     const float* w4, const float* w2, float* d_nf, float* d_add, float* dx4, int64_t ld_dx4,
                                     float* d_pt, float* d_token, int G, int N, int C, int W2, float slope4, float slope2,
                                     float p_pos, float p_in, uint64_t seed, const uint64_t* seed_dev, uint32_t salt_nf,
                                     uint32_t salt_tok, uint32_t salt_in, void* stream) {
    if (G <= 0 || N <= 0) return MOBGT_EBADDIM;
    if (!(C == 192 && W2 == 160)) return MOBGT_EBADDIM;          // the instantiated widths (MobGT's hidden 128: C = 192, [poi ; time] = 160)
    if (((uintptr_t)w4 | (uintptr_t)w2) & 15) return MOBGT_EALIGN;
    TokBwdParams p = {};
    p.dout = dout; p.real = real; p.y4 = y4; p.y2 = y2; p.ld2 = ld_y2; p.w4 = w4; p.w2 = w2; p.d_nf = d_nf; p.d_add = d_add;
    p.dx4 = dx4; p.ldx4 = ld_dx4; p.d_pt = d_pt; p.d_token = d_token; p.G = G; p.N = N; p.slope4 = slope4; p.slope2 = slope2;
    p.thr_pos = p_pos > 0.f ? dropout_threshold(p_pos) : 0u;
    p.thr_in = p_in > 0.f ? dropout_threshold(p_in) : 0u;
    p.keep_pos = p.thr_pos ? 1.f / (1.f - (float)p.thr_pos / 65536.f) : 1.f;
    p.keep_in = p.thr_in ? 1.f / (1.f - (float)p.thr_in / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt_nf = salt_nf; p.salt_tok = salt_tok; p.salt_in = salt_in;
    p.n_node_blocks = (int)(((int64_t)G * N + TBM - 1) / TBM);
    const dim3 grid(p.n_node_blocks + (G + TNW - 1) / TNW), block(TNT);
    hipLaunchKernelGGL((token_bwd_chain_kernel<192, 160>), grid, block, 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
