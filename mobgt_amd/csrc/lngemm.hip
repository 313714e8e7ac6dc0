// LayerNorm / dropout / residual steps of the encoder layer as the PROLOGUE of the GEMM that consumes their result
// (graphormer/model.py:479-489, model_fqandtoyo.py:1731-1743 and their autograd).
//
// Every launch of the S-FSQ train step is latency (4.5-7 us for a few hundred rows), so what counts is the number of
// dependent launches.  Three of the layer's `dropout_add_ln` launches produce the A operand of a GEMM whose
// contraction axis is the model width C, i.e. a workgroup of that GEMM walks WHOLE rows of A anyway:
//     forward   z  = LayerNorm(x + dropout(y))           -> u  = z W1^T + b1, h = gelu(u)          (FFN layer 1)
//     backward  df = dropout'(LayerNorm2'(dout))         -> du = (df W2) * gelu'(u)
//     backward  dy = dropout'(dx2 + LayerNorm1'(dz))     -> da = dy Wo
// Here the workgroup owning a 32-row output tile first evaluates the elementwise / normalisation step for its 32
// rows (same arithmetic, same dropout masks as layer.hip's stand-alone kernels) into an LDS image of A in bf16, then
// runs the split-K MFMA loop of gemm_body.h with A fragments read from LDS.  The N/64 column tiles of a row block
// repeat that prologue (32 x C elements: nothing next to a launch); only the FIRST column tile writes the step's
// own outputs -- x1 / statistics / z, or dx1 / dy and the dgamma / dbeta / dbias column sums -- that the backward
// pass and the weight gradients read later.  One launch instead of two, three times per layer: 18 of 84 launches.
#include "common.h"
#include "mobgt_hip.h"
#include "gemm_body.h"

namespace {

using namespace mobgt_gemm;

constexpr int LN_MAXC = 256;                 // lane l owns columns l + 64 k, k < 4
constexpr int APITCH = LN_MAXC + 8;          // bf16 elements per LDS row of A (528 B = 33 x 16 B)

struct LnGemmParams {
    // ---- prologue (names as in layer.hip's LnParams)
    const float* x;            // fwd: residual in [R,C] f32
    const uint16_t* y;         // fwd: branch output [R,C] bf16
    float* x1;                 // fwd out: x + dropout(y);   bwd in: the LayerNorm's input
    const float *w, *b;        // LayerNorm affine
    uint16_t* z;               // fwd out: LayerNorm output bf16 (the GEMM's A, also kept for the weight gradient)
    float *mean, *rstd;        // [R] fwd out / bwd in
    const uint16_t* dz;        // bwd: grad of the LN output, bf16, or null
    const float* dz32;         // bwd: grad of the LN output, f32, or null (added)
    const float* dres;         // bwd: grad arriving at x1 from downstream, f32, or null
    float* dx1;                // bwd out: total grad at x1
    uint16_t* dy;              // bwd out: grad of the branch output (the GEMM's A), bf16
    float *dgamma, *dbeta, *dbias;
    float inv_keep;
    uint32_t thr;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt;
    // ---- GEMM: C[M,N] = A[M,K=C] . B (+ bias), epilogues of gemm_body.h
    GemmParams g;
};

// Rows m0 .. m0+31 of A into `As`.  ROW-PARALLEL: eight adjacent lanes share a row (thread t: row t >> 3, columns
// [part C/8, (part+1) C/8) in groups of four), so all 32 rows are in flight at once and the prologue costs ONE global-load
// round trip plus two 3-step lane reductions.  (A first version walked the rows one after another, eight per wave as
// layer.hip's streaming kernels do: eight exposed load latencies, 20 us for a step that takes 5 us on its own.)
__device__ __forceinline__ float sum8(float v) {          // over the 8 lanes of a row
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}
__device__ __forceinline__ void ld4f(const float* p, float (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void ld4h(const uint16_t* p, float (&v)[4]) {
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    v[0] = bf16_lo(a.x); v[1] = bf16_hi(a.x); v[2] = bf16_lo(a.y); v[3] = bf16_hi(a.y);
}
__device__ __forceinline__ uint2 pack4(const float (&v)[4]) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (bf16_t)v[i];
    return __builtin_bit_cast(uint2, o);
}

constexpr int LN_Q = LN_MAXC / 32;           // groups of 4 columns per thread, at most

// dgamma / dbeta / dbias of a SIDE workgroup: the three per-column sums over its 32 rows.  Lanes l, l^8, l^16, l^32 hold
// the same columns of different rows: three xor steps sum a wave's 8 rows, the 4 waves meet in LDS, one global atomic
// per column.  (LDS atomics straight from every thread -- same-address 8-way within an instruction -- cost ~20 us.)
__device__ __forceinline__ float sum_rows8(float v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ void ln_colsums(const LnGemmParams& p, const float (&dd)[LN_Q][4], const float (&xh)[LN_Q][4],
                                           const float (&yv)[LN_Q][4], float (*colred)[4][LN_MAXC]) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, part = tid & 7;
    const int C = p.g.K, per = C >> 3, c0 = part * per, n4 = per >> 2;
#pragma unroll
    for (int q = 0; q < LN_Q; ++q)
        if (q < n4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = sum_rows8(dd[q][i] * xh[q][i]), b = sum_rows8(dd[q][i]), y = sum_rows8(yv[q][i]);
                if (lane < 8) {
                    colred[0][wave][c0 + 4 * q + i] = a;
                    colred[1][wave][c0 + 4 * q + i] = b;
                    colred[2][wave][c0 + 4 * q + i] = y;
                }
            }
        }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        if (p.dgamma) {
            atomicAdd(&p.dgamma[c], colred[0][0][c] + colred[0][1][c] + colred[0][2][c] + colred[0][3][c]);
            atomicAdd(&p.dbeta[c], colred[1][0][c] + colred[1][1][c] + colred[1][2][c] + colred[1][3][c]);
        }
        if (p.dbias) atomicAdd(&p.dbias[c], colred[2][0][c] + colred[2][1][c] + colred[2][2][c] + colred[2][3][c]);
    }
}


template <bool BWD>
__device__ __forceinline__ void ln_prologue(const LnGemmParams& p, const int m0, const bool side, uint16_t (*As)[APITCH],
                                            float (*colred)[4][LN_MAXC]) {
    const int tid = threadIdx.x, rr = tid >> 3, part = tid & 7;
    const int C = p.g.K, per = C >> 3, c0 = part * per, n4 = per >> 2;
    const int64_t r = m0 + rr;
    const bool on = r < p.g.M;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    const uint32_t rowh = (p.thr && on) ? dropout_row_hash(seed, (uint32_t)r ^ p.salt) : 0u;
    const float invC = 1.f / (float)C;
    float out[LN_Q][4];
    if (!BWD) {
        // x1 = x + dropout(y);  z = LayerNorm(x1)     (layer.hip: dropout_add_ln_fwd_kernel)
        float v[LN_Q][4], s = 0.f;
#pragma unroll
        for (int q = 0; q < LN_Q; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[q][i] = 0.f;
            if (q < n4 && on) {
                const int c = c0 + 4 * q;
                float yv[4];
                ld4f(p.x + r * C + c, v[q]);
                if (p.y) {                                           // (NULL: a plain LayerNorm of x -- the pre-LN layer's first norm)
                    ld4h(p.y + r * C + c, yv);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (p.thr) yv[i] = dropout_bits16(seed, rowh, (uint32_t)(c + i)) >= p.thr ? yv[i] * p.inv_keep : 0.f;
                        v[q][i] += yv[i];
                    }
                }
                if (side && p.x1) *reinterpret_cast<float4*>(p.x1 + r * C + c) = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s += v[q][i];
        }
        const float mu = sum8(s) * invC;
        float qq = 0.f;
#pragma unroll
        for (int q = 0; q < LN_Q; ++q)
            if (q < n4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = v[q][i] - mu; qq += d * d; }
            }
        const float rs = rsqrtf(sum8(qq) * invC + 1e-5f);
        if (side && on && part == 0) { p.mean[r] = mu; p.rstd[r] = rs; }
#pragma unroll
        for (int q = 0; q < LN_Q; ++q)
            if (q < n4) {
                const int c = c0 + 4 * q;
                float w4[4], b4[4];
                ld4f(p.w + c, w4);
                ld4f(p.b + c, b4);
#pragma unroll
                for (int i = 0; i < 4; ++i) out[q][i] = on ? (v[q][i] - mu) * rs * w4[i] + b4[i] : 0.f;
            }
    } else {
        // dx1 = dres + LayerNorm'(dz + dz32);  dy = dropout'(dx1)     (layer.hip: dropout_add_ln_bwd_kernel)
        float xh[LN_Q][4], g[LN_Q][4], dres[LN_Q][4], dd[LN_Q][4], s1 = 0.f, s2 = 0.f;
        const float mu = on ? p.mean[r] : 0.f, rs = on ? p.rstd[r] : 0.f;
#pragma unroll
        for (int q = 0; q < LN_Q; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { xh[q][i] = 0.f; g[q][i] = 0.f; dres[q][i] = 0.f; dd[q][i] = 0.f; }
            if (q < n4 && on) {
                const int c = c0 + 4 * q;
                float d[4] = {0.f, 0.f, 0.f, 0.f}, t[4], x4[4], w4[4];
                if (p.dz) { ld4h(p.dz + r * C + c, t); for (int i = 0; i < 4; ++i) d[i] += t[i]; }
                if (p.dz32) { ld4f(p.dz32 + r * C + c, t); for (int i = 0; i < 4; ++i) d[i] += t[i]; }
                if (p.dres) ld4f(p.dres + r * C + c, dres[q]);
                ld4f(p.x1 + r * C + c, x4);
                ld4f(p.w + c, w4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xh[q][i] = (x4[i] - mu) * rs;
                    g[q][i] = d[i] * w4[i];
                    s1 += g[q][i];
                    s2 += g[q][i] * xh[q][i];
                    dd[q][i] = d[i];
                }
            }
        }
        s1 = sum8(s1) * invC;
        s2 = sum8(s2) * invC;
#pragma unroll
        for (int q = 0; q < LN_Q; ++q)
            if (q < n4) {
                const int c = c0 + 4 * q;
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    t[i] = on ? rs * (g[q][i] - s1 - xh[q][i] * s2) + dres[q][i] : 0.f;
                    float yv = t[i];
                    if (p.thr) yv = dropout_bits16(seed, rowh, (uint32_t)(c + i)) >= p.thr ? t[i] * p.inv_keep : 0.f;
                    out[q][i] = yv;
                }
                if (side && on) *reinterpret_cast<float4*>(p.dx1 + r * C + c) = make_float4(t[0], t[1], t[2], t[3]);
            }
        if (side) {           // (rows past M and columns past n4 hold zeros)
#pragma unroll
            for (int q = 0; q < LN_Q; ++q)
                if (q >= n4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[q][i] = 0.f;
                }
            ln_colsums(p, dd, xh, out, colred);
        }
    }
    uint16_t* gout = BWD ? p.dy : p.z;
#pragma unroll
    for (int q = 0; q < LN_Q; ++q)
        if (q < n4) {
            const int c = c0 + 4 * q;
            const uint2 h = pack4(out[q]);
            *reinterpret_cast<uint2*>(&As[rr][c]) = h;
            if (side && on && gout) *reinterpret_cast<uint2*>(gout + r * C + c) = h;
        }
}

template <bool BWD, bool BKN, int EPI, int NB, int NW>
__global__ __launch_bounds__(NW * 64) void ln_gemm_kernel(const LnGemmParams p) {
    constexpr int BN = 16 * NB;
    constexpr int LDP = BN + 4;
    __shared__ __attribute__((aligned(16))) uint16_t As[BM][APITCH];
    __shared__ __attribute__((aligned(16))) float part[NW][BM * LDP];
    __shared__ float colred[BWD ? 3 : 1][4][LN_MAXC];
    static_assert(NW == 4, "the prologue maps 256 threads onto 32 rows x 8 parts");
    const GemmParams& g = p.g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int bid = blockIdx.x;
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

    // B fragments of this wave's k-steps (k = 32 (wave + 4 j), at most MAXS of them for K <= 256) are requested FIRST: they
    // depend on nothing the prologue computes, so their load latency hides behind the prologue's own round trip.
    constexpr int MAXS = (LN_MAXC / KSTEP + NW - 1) / NW;
    const uint16_t* bp[NB];
    if (BKN) {
#pragma unroll
        for (int bb = 0; bb < NB; ++bb)
            bp[bb] = bb < NB / 2 ? g.B + (int64_t)(8 * kq) * g.ldb + min(n0 + 32 * bb + 2 * i, g.N - 2) : nullptr;
    } else {
#pragma unroll
        for (int t = 0; t < NB; ++t) bp[t] = g.B + (int64_t)min(n0 + 16 * t + i, g.N - 1) * g.ldb + 8 * kq;
    }
    uint32_t braw[MAXS][BKN ? NB / 2 : 1][BKN ? 8 : 1];
    uint4 bvec[MAXS][BKN ? 1 : NB];
#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
        const int k = (wave + NW * j) * KSTEP;
        if (k < g.K) {
            if constexpr (BKN) {
#pragma unroll
                for (int bb = 0; bb < NB / 2; ++bb)
#pragma unroll
                    for (int r = 0; r < 8; ++r) braw[j][bb][r] = *reinterpret_cast<const uint32_t*>(bp[bb] + (int64_t)(k + r) * g.ldb);
            } else {
#pragma unroll
                for (int t = 0; t < NB; ++t) bvec[j][t] = *reinterpret_cast<const uint4*>(bp[t] + k);
            }
        }
    }

    ln_prologue<BWD>(p, m0, bid % tiles_n == 0, As, colred);
    __syncthreads();

    f32x4 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int j = 0; j < MAXS; ++j) {
        const int k = (wave + NW * j) * KSTEP;
        if (k < g.K) {
            bf16x8 bf[NB];
            if constexpr (BKN) {
#pragma unroll
                for (int bb = 0; bb < NB / 2; ++bb) split_pairs(braw[j][bb], bf[2 * bb], bf[2 * bb + 1]);
            } else {
#pragma unroll
                for (int t = 0; t < NB; ++t) bf[t] = __builtin_bit_cast(bf16x8, bvec[j][t]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&As[16 * a + i][k + 8 * kq]));
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[b], acc[a][b], 0, 0, 0);
            }
        }
    }

    // from here on: gemm_body.h's reduction of the waves' partial tiles and its epilogues
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
    constexpr int TPR = BN / 8;
    if (threadIdx.x >= BM * TPR) return;
    const int r = threadIdx.x / TPR, c = (threadIdx.x % TPR) * 8;
    const int row = m0 + r, col = n0 + c;
    if (row >= g.M || col >= g.N) return;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const int nw_used = min(NW, (g.K + KSTEP - 1) / KSTEP);
    for (int w = 0; w < nw_used; ++w) {
        const float4 x0 = *reinterpret_cast<const float4*>(&part[w][r * LDP + c]);
        const float4 x1 = *reinterpret_cast<const float4*>(&part[w][r * LDP + c + 4]);
        s[0] += x0.x; s[1] += x0.y; s[2] += x0.z; s[3] += x0.w; s[4] += x1.x; s[5] += x1.y; s[6] += x1.z; s[7] += x1.w;
    }
    if (g.bias) {
        float bv[8];
        load8(reinterpret_cast<const bf16_t*>(g.bias) + col, bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += bv[e];
    }
    const int64_t o = (int64_t)row * g.ldc + col;
    if constexpr (EPI == EPI_BIAS) {
        store8(reinterpret_cast<bf16_t*>(g.C) + o, s);
    } else if constexpr (EPI == EPI_GELU) {
        const bf16x8 u8 = pack8(s);
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(g.C) + o) = u8;
        float h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = gelu_f((float)u8[e]);
        store8(reinterpret_cast<bf16_t*>(g.aux_out) + o, h);
    } else {        // EPI_GELU_BWD
        float u[8];
        load8(reinterpret_cast<const bf16_t*>(g.aux_in) + o, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] *= gelu_grad(u[e]);
        store8(reinterpret_cast<bf16_t*>(g.C) + o, s);
    }
}

void set_drop(LnGemmParams& p, float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt) {
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
}

template <bool BWD, bool BKN, int EPI>
int launch(const LnGemmParams& p, hipStream_t st) {
    const GemmParams& g = p.g;
    const int tiles64 = ((g.M + BM - 1) / BM) * ((g.N + 63) / 64);
    if (tiles64 < 128) {
        const dim3 grid(((g.M + BM - 1) / BM) * ((g.N + 31) / 32));
        hipLaunchKernelGGL((ln_gemm_kernel<BWD, BKN, EPI, 2, 4>), grid, dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL((ln_gemm_kernel<BWD, BKN, EPI, 4, 4>), dim3(tiles64), dim3(256), 0, st, p);
    }
    return (int)hipGetLastError();
}

int check_gemm(int64_t R, int C, int N, int64_t ldb, int64_t ldc, bool bkn, const void* b, const void* c) {
    if (R <= 0 || R > 0x7fffffff || C <= 0 || C > LN_MAXC || (C % KSTEP) || N <= 0 || (N & 7) || (ldc & 7)) return MOBGT_EBADDIM;
    if (bkn ? (ldb & 1) : (ldb & 7)) return MOBGT_EBADDIM;
    if ((uintptr_t)c & 15) return MOBGT_EALIGN;
    if ((uintptr_t)b & (bkn ? 3 : 15)) return MOBGT_EALIGN;
    return 0;
}

}  // namespace

extern "C" int mobgt_ln_gemm_fwd(const float* x, const void* y, float* x1, const float* ln_w, const float* ln_b, void* z,
                                 float* mean, float* rstd, int64_t R, int C, float dropout_p, uint64_t seed,
                                 const uint64_t* seed_dev, uint32_t salt, const void* weight, int64_t ldw, const void* bias,
                                 void* out, int64_t ld_out, int epilogue, void* aux_out, int N, void* stream) {
    int rc = check_gemm(R, C, N, ldw, ld_out, false, weight, out);
    if (rc) return rc;
    if (!x || (!y) != (!x1) || !ln_w || !ln_b || !z || !mean || !rstd) return MOBGT_EBADDIM;      // y, x1: both or neither
    if (epilogue == EPI_GELU ? !aux_out : epilogue != EPI_BIAS) return MOBGT_EBADDIM;
    LnGemmParams p = {};
    p.x = x; p.y = reinterpret_cast<const uint16_t*>(y); p.x1 = x1; p.w = ln_w; p.b = ln_b;
    p.z = reinterpret_cast<uint16_t*>(z); p.mean = mean; p.rstd = rstd;
    set_drop(p, dropout_p, seed, seed_dev, salt);
    p.g = GemmParams{nullptr, 0, reinterpret_cast<const uint16_t*>(weight), ldw, reinterpret_cast<const uint16_t*>(bias), out,
                     ld_out, nullptr, reinterpret_cast<uint16_t*>(aux_out), (int)R, N, C};
    hipStream_t st = (hipStream_t)stream;
    return epilogue == EPI_GELU ? launch<false, false, EPI_GELU>(p, st) : launch<false, false, EPI_BIAS>(p, st);
}

extern "C" int mobgt_ln_gemm_bwd(const void* dz, const float* dz32, const float* dres, const float* x1, const float* mean,
                                 const float* rstd, const float* ln_w, float* dx1, void* dy, float* dgamma, float* dbeta,
                                 float* dbias, int64_t R, int C, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                 uint32_t salt, const void* weight_kn, int64_t ldw, void* out, int64_t ld_out, int epilogue,
                                 const void* aux_in, int N, void* stream) {
    int rc = check_gemm(R, C, N, ldw, ld_out, true, weight_kn, out);
    if (rc) return rc;
    if ((!dz && !dz32) || !x1 || !mean || !rstd || !ln_w || !dx1 || !dy) return MOBGT_EBADDIM;
    if (epilogue == EPI_GELU_BWD ? !aux_in : epilogue != EPI_BIAS) return MOBGT_EBADDIM;
    LnGemmParams p = {};
    p.dz = reinterpret_cast<const uint16_t*>(dz); p.dz32 = dz32; p.dres = dres; p.x1 = const_cast<float*>(x1);
    p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd); p.w = ln_w; p.dx1 = dx1;
    p.dy = reinterpret_cast<uint16_t*>(dy); p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias;
    set_drop(p, dropout_p, seed, seed_dev, salt);
    p.g = GemmParams{nullptr, 0, reinterpret_cast<const uint16_t*>(weight_kn), ldw, nullptr, out, ld_out, aux_in, nullptr,
                     (int)R, N, C};
    hipStream_t st = (hipStream_t)stream;
    return epilogue == EPI_GELU_BWD ? launch<true, true, EPI_GELU_BWD>(p, st) : launch<true, true, EPI_BIAS>(p, st);
}
