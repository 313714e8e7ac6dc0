// The classifier head in front of out_proj, one launch each way (model_fqandtoyo.py:1239-1240, 1353-1364; FuseEmbeddings:
// 452-455):
//
//     x3  = [ encoder output of the graph token | user_embedding[user + offset] ]          [G, W],  W = C + U
//     u3  = x3 W3^T + b3                                                                    FuseEmbeddings' Linear (f32)
//     tok = dropout( ELU( LayerNorm( LeakyReLU_0.2(u3) ) ) )
//
// As separate launches (gather, small GEMM, activation chain) these were 16 us forward and 17 us backward of the S-FSQ step for
// 16 rows: three / three round trips through global memory and three launch ramps for 3 MFLOP.  Full-f32 products on
// v_mfma_f32_16x16x4_f32; forward the k index of a step is permuted (lane quarter q, element i of a 16-byte load at
// k = 16 t + 4 q + i) so that BOTH operands are 16-byte loads -- A from LDS, B straight from the row-major weight.
//
// Backward: d(tok) -> activation chain backwards (dgamma, dbeta) -> du3 (kept: the weight gradient du3^T x3 is one of the
// step's grouped leaf products) -> dx3 = du3 W3, whose first C columns are the graph-token rows of d(encoder output) (all
// other rows of it are zero-filled by the launch's other workgroups) and whose last U columns are added into the user table's
// gradient.
#include "common.h"
#include "mobgt_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int HBM = 16;                        // rows per block

struct HeadChainParams {
    const float* enc;                        // [G,T,C]
    const void* user; int user_dtype; int64_t user_offset;
    const float* table; int64_t n_rows;      // [n_rows, U]
    const float *w3, *b3, *ln_w, *ln_b;      // [W,W] (out, in) row-major, [W], [W], [W]
    float *x3, *u3, *out, *mean, *rstd;      // forward: written;  backward: x3 unused, u3 / mean / rstd read
    const float* dout;                       // backward: d(tok) [G,W]
    float *du3, *denc, *dtable, *dgamma, *dbeta;      // du3 [G,W] written; denc [G,T,C] written in full; the rest accumulated
    int G, T, C, U;
    float eps, slope, inv_keep;
    uint32_t thr;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt;
};

__device__ __forceinline__ float hwave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int64_t hc_user(const HeadChainParams& p, int g) {
    const int64_t u = p.user_dtype == MOBGT_I32 ? (int64_t)reinterpret_cast<const int32_t*>(p.user)[g]
                                                 : reinterpret_cast<const int64_t*>(p.user)[g];
    return u + p.user_offset;
}

// the activation chain of one row (one wave per row), forward or backward; `a_src` = this row of u3 in LDS or global
template <int W, bool BWD>
__device__ __forceinline__ void head_act_row(const HeadChainParams& p, int row, bool store, bool on, const float* __restrict__ u_row,
                                             const float* __restrict__ d_row, float* __restrict__ o_row, float* __restrict__ colred,
                                             int c0, float mu_in, float rs_in, uint64_t seed) {
    constexpr int PER = W / 64;
    const int lane = threadIdx.x & 63;
    const uint32_t rowh = p.thr ? dropout_row_hash(seed, (uint32_t)row ^ p.salt) : 0u;
    float a[PER], uu[PER], s = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        uu[k] = u_row[lane + 64 * k];
        a[k] = uu[k] > 0.f ? uu[k] : p.slope * uu[k];
        s += a[k];
    }
    float mu = mu_in, rs = rs_in;
    if (!BWD) {
        mu = hwave_sum(s) * (1.f / W);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float d = a[k] - mu;
            q += d * d;
        }
        rs = rsqrtf(hwave_sum(q) * (1.f / W) + p.eps);
        if (on && lane == 0) { p.mean[row] = mu; p.rstd[row] = rs; }
    }
    float g[PER], xh[PER], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = lane + 64 * k;
        xh[k] = (a[k] - mu) * rs;
        const float z = xh[k] * p.ln_w[c] + p.ln_b[c];
        const float keep = p.thr ? (dropout_bits16(seed, rowh, (uint32_t)c) >= p.thr ? p.inv_keep : 0.f) : 1.f;
        if (!BWD) {
            if (on) o_row[c] = (z > 0.f ? z : expm1f(z)) * keep;
        } else {
            const float dz = on ? d_row[c] * keep * (z > 0.f ? 1.f : expf(z)) : 0.f;      // ELU'(z) = e^z for z <= 0
            if (c >= c0 && c < c0 + 16) {            // this member's columns of the dgamma | dbeta terms (colred [2][HBM][16], row given)
                colred[c - c0] = dz * xh[k];
                colred[HBM * 16 + c - c0] = dz;
            }
            g[k] = dz * p.ln_w[c];
            s1 += g[k];
            s2 += g[k] * xh[k];
        }
    }
    if (!BWD) return;
    s1 = hwave_sum(s1) * (1.f / W);
    s2 = hwave_sum(s2) * (1.f / W);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = lane + 64 * k;
        const float da = rs * (g[k] - s1 - xh[k] * s2);
        const float du = on ? da * (uu[k] > 0.f ? 1.f : p.slope) : 0.f;
        o_row[c] = du;                               // LDS: the A operand of dx3 = du3 W3
        if (store) p.du3[(int64_t)row * W + c] = du;
    }
}

// ---- hand-over between the workgroups of a cluster (see csrc/chain.hip: cluster_put / cluster_get) ------------------------------
// One f32 MFMA pass over the 16 x W x W product is 5.3 us on ONE compute unit (v_mfma_f32_16x16x4_f32: 256 flop per clock and
// unit), and its 410 KB weight would stream through one L1: the first form of this file -- one workgroup per 16 rows -- took
// 24 + 43 us.  So a 16-row block belongs to a CLUSTER of W / 16 workgroups; member m owns 16 columns: of u3 forward (its 16 rows
// of the weight: 20 KB), of dx3 backward (its 16 columns of the weight).  Forward the members exchange their [16 x 16] tiles of
// u3 once (the LayerNorm needs whole rows) as flag-carrying 16-byte packets {v0, tag, v1, tag}; backward needs no exchange
// (every member recomputes the 16 rows of du3 from u3 / d(tok), 20 KB each).  tag = gen[block] + 1, raised by member 0 once its
// whole workgroup holds everybody's packets.  Members sit on one XCD (block ids that agree modulo 8).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int HLL_SC0 = 1, HLL_SC1 = 16, HLL_VOLATILE = (int)0x80000000;
constexpr int HWS_GEN_INTS = 64;                               // gen[<= 16 row blocks] (+ spare)
constexpr int HMAX_BLOCKS = 10;                                 // row blocks per launch: 10 x 24 members <= 256 compute units

template <int NCL>
__device__ __forceinline__ void head_cluster_ids(int bid, int& blk, int& m) {
    const int xcd = bid & 7, slot = bid >> 3;
    m = slot % NCL;
    blk = (slot / NCL) * 8 + xcd;
}

template <int W>
__global__ __launch_bounds__(256) void head_chain_fwd_kernel(const HeadChainParams p, uint32_t* ws_gen, uint64_t* ws_ll) {
    constexpr int LD = W + 4, NCL = W / 16, NTS = W / 16, TPW = NTS / 4;      // members; k-steps of 16, per wave
    static_assert(NTS % 4 == 0, "the K range splits over the four waves");
    constexpr int NP = HBM * 16 / 2;                                         // packets per member tile
    __shared__ __attribute__((aligned(16))) float xs[HBM * LD];             // x3 rows, later u3 rows
    __shared__ __attribute__((aligned(16))) float part[4 * HBM * 16];
    int blk, m;
    head_cluster_ids<NCL>((int)blockIdx.x, blk, m);
    const int r0 = blk * HBM;
    if (r0 >= p.G) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    // x3 rows (all of K: every member multiplies whole rows) and this wave's K quarter of the member's 16 weight rows
    constexpr int XE = (HBM * (W / 4) + 255) / 256;
    f32x4 x_in[XE];
#pragma unroll
    for (int k = 0; k < XE; ++k) {
        const int e = min((int)threadIdx.x + k * 256, HBM * (W / 4) - 1), r = e / (W / 4), c = (e % (W / 4)) * 4;
        const int g = min(r0 + r, p.G - 1);
        if (c < p.C) {
            x_in[k] = *reinterpret_cast<const f32x4*>(p.enc + (int64_t)g * p.T * p.C + c);
        } else {
            const int64_t u = hc_user(p, g);
            x_in[k] = (u >= 0 && u < p.n_rows) ? *reinterpret_cast<const f32x4*>(p.table + u * p.U + (c - p.C)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 b[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
        b[t] = *reinterpret_cast<const f32x4*>(p.w3 + (int64_t)(16 * m + j) * W + 16 * (wave * TPW + t) + 4 * q);
    const float bias = p.b3[16 * m + (threadIdx.x & 15)];
    const uint32_t gen = __hip_atomic_load(ws_gen + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < XE; ++k) {
        const int e = (int)threadIdx.x + k * 256, r = e / (W / 4), c = (e % (W / 4)) * 4;
        if (e < HBM * (W / 4)) {
            *reinterpret_cast<f32x4*>(xs + r * LD + c) = x_in[k];
            if (m == 0 && r0 + r < p.G) *reinterpret_cast<f32x4*>(p.x3 + (int64_t)(r0 + r) * W + c) = x_in[k];      // (the weight gradient reads it)
        }
    }
    __syncthreads();
    // ---- this member's 16 columns of u3 = x3 W3^T + b3: a K quarter per wave, the partial tiles meet in LDS
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xs + j * LD + 16 * (wave * TPW + t) + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[t][i], acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) part[(wave * HBM + 4 * q + v) * 16 + j] = acc[v];
    }
    __syncthreads();
    const uint32_t tag = gen + 1u;
    const __amdgpu_buffer_rsrc_t area = __builtin_amdgcn_make_buffer_rsrc(ws_ll + (int64_t)blk * NCL * NP * 2, 0, NCL * NP * 16, 0x00020000);
    {
        const int r = threadIdx.x >> 4, c = threadIdx.x & 15;               // 256 threads = the 16 x 16 tile
        const float u = part[(0 * HBM + r) * 16 + c] + part[(1 * HBM + r) * 16 + c] + part[(2 * HBM + r) * 16 + c] +
                        part[(3 * HBM + r) * 16 + c] + bias;
        if (r0 + r < p.G) p.u3[(int64_t)(r0 + r) * W + 16 * m + c] = u;
        const float un = __shfl_down(u, 1, 64);                              // column c + 1 of the same row
        if (!(c & 1)) {
            const u32x4 v = {__float_as_uint(u), tag, __float_as_uint(un), tag};
            __builtin_amdgcn_raw_buffer_store_b128(v, area, (m * NP + r * 8 + (c >> 1)) * 16, 0, HLL_SC1);
        }
    }
    // ---- every member's tile -> u3 rows in LDS (xs is free: the barrier above is behind every wave's last read of it)
    {
        constexpr int EPT = (NCL * NP + 255) / 256;
        u32x4 w[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) w[k] = u32x4{0u, 0u, 0u, 0u};
        int rounds = 0;
        bool missing;
        const uint32_t lim_raw = __builtin_nontemporal_load(ws_gen + (HWS_GEN_INTS - 2));
        const int limit = (int)(lim_raw ? lim_raw : (1u << 21));
        do {
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int e = min((int)threadIdx.x + k * 256, NCL * NP - 1);
                if (w[k][1] != tag || w[k][3] != tag) {
                    if (rounds < 24) w[k] = __builtin_amdgcn_raw_buffer_load_b128(area, e * 16, 0, HLL_SC0 | HLL_VOLATILE);
                    else w[k] = __builtin_amdgcn_raw_buffer_load_b128(area, e * 16, 0, HLL_SC1 | HLL_VOLATILE);
                }
            }
            asm volatile("" ::: "memory");
            missing = false;
#pragma unroll
            for (int k = 0; k < EPT; ++k) missing |= w[k][1] != tag || w[k][3] != tag;
            if (++rounds > limit) {          // give up instead of trapping (see csrc/chain.hip: WS_FAULT); the host finds the count
                if (missing) atomicAdd(ws_gen + (HWS_GEN_INTS - 1), 1u);
                break;
            }
        } while (missing);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = (int)threadIdx.x + k * 256;
            if (e < NCL * NP) {
                const int mm = e / NP, o = e % NP, r = o / 8, c = 16 * mm + 2 * (o % 8);
                xs[r * LD + c] = __uint_as_float(w[k][0]);
                xs[r * LD + c + 1] = __uint_as_float(w[k][2]);
            }
        }
    }
    __syncthreads();
    if (m == 0 && threadIdx.x == 0) __hip_atomic_store(ws_gen + blk, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- the activation chain: member m < 16 takes row m (one wave)
    if (m < HBM && wave == 0) {
        const int row = r0 + m;
        head_act_row<W, false>(p, row, false, row < p.G, xs + m * LD, nullptr, p.out + (int64_t)min(row, p.G - 1) * W, nullptr, 0, 0.f, 0.f, seed);
    }
}

// backward: the first `n_cluster` block ids are clusters (member m: columns [16 m, 16 m + 16) of dx3 and of dgamma / dbeta / du3);
// the blocks behind them zero-fill the rows of d(enc) that are not graph tokens
template <int W>
__global__ __launch_bounds__(512) void head_chain_bwd_kernel(const HeadChainParams p, int n_cluster) {
    constexpr int LD = W + 4, NCL = W / 16, NWV = 8, KS = W / 4;           // k-steps of 4 over the whole K
    static_assert(KS % NWV == 0, "the K range splits over the eight waves");
    constexpr int SPW = KS / NWV;                                           // k-steps per wave
    __shared__ __attribute__((aligned(16))) float ds[HBM * LD];             // du3 rows (the A operand)
    __shared__ float colred[2 * HBM * 16];                                  // this member's columns of dz * xhat | dz, per row
    __shared__ float part[NWV * HBM * 16];
    if ((int)blockIdx.x >= n_cluster) {
        const int64_t per_g = (int64_t)(p.T - 1) * p.C / 4, total = (int64_t)p.G * per_g;
        const int64_t stride = (int64_t)(gridDim.x - n_cluster) * 512;
        for (int64_t e = (int64_t)(blockIdx.x - n_cluster) * 512 + threadIdx.x; e < total; e += stride) {
            const int64_t g = e / per_g, o = e - g * per_g;
            *reinterpret_cast<f32x4*>(p.denc + (g * p.T + 1) * p.C + o * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    int blk, m;
    head_cluster_ids<NCL>((int)blockIdx.x, blk, m);
    const int r0 = blk * HBM;
    if (r0 >= p.G) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    // this wave's K range of the member's 16 weight columns: B[k][n] = W3[k][16 m + n], one float per lane and k-step
    float bw[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) bw[s] = p.w3[(int64_t)(4 * (wave * SPW + s) + q) * W + 16 * m + j];
    // ---- activation chain backwards: every member recomputes all 16 rows of du3 (two per wave)
#pragma unroll
    for (int i = 0; i < HBM / NWV; ++i) {
        const int r = wave + NWV * i, row = min(r0 + r, p.G - 1);
        const bool on = r0 + r < p.G;
        head_act_row<W, true>(p, r0 + r, on && m == 0, on, p.u3 + (int64_t)row * W, p.dout + (int64_t)row * W, ds + r * LD,
                              colred + r * 16, 16 * m, p.mean[row], p.rstd[row], seed);
    }
    __syncthreads();
    if (threadIdx.x < 32) {                                                  // dgamma | dbeta of the member's columns
        const int which = threadIdx.x >> 4, c = threadIdx.x & 15;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < HBM; ++r) s += colred[(which * HBM + r) * 16 + c];
        atomicAdd((which ? p.dbeta : p.dgamma) + 16 * m + c, s);
    }
    // ---- the member's 16 columns of dx3 = du3 W3
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < SPW; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[j * LD + 4 * (wave * SPW + s) + q], bw[s], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) part[(wave * HBM + 4 * q + v) * 16 + j] = acc[v];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int r = threadIdx.x >> 4, c = threadIdx.x & 15, col = 16 * m + c, g = r0 + r;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) s += part[(w * HBM + r) * 16 + c];
        if (g < p.G) {
            if (col < p.C) {
                p.denc[(int64_t)g * p.T * p.C + col] = s;
            } else {
                const int64_t u = hc_user(p, g);
                if (u >= 0 && u < p.n_rows) atomicAdd(p.dtable + u * p.U + (col - p.C), s);
            }
        }
    }
}

int fill(HeadChainParams& p, float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt) {
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
    return 0;
}

}  // namespace

extern "C" int64_t mobgt_head_chain_ws_bytes(void) {
    return (int64_t)(HWS_GEN_INTS * sizeof(uint32_t) + (size_t)HMAX_BLOCKS * 24 * (HBM * 16 / 2) * 16);
}
// (the last two words of the generation array: workgroups that gave up waiting for their cluster partners / poll limit, 0 = default)
extern "C" int64_t mobgt_head_chain_ws_fault_offset(void) { return (int64_t)(HWS_GEN_INTS - 1) * (int64_t)sizeof(uint32_t); }
extern "C" int64_t mobgt_head_chain_ws_limit_offset(void) { return (int64_t)(HWS_GEN_INTS - 2) * (int64_t)sizeof(uint32_t); }

extern "C" int mobgt_head_chain_fwd(const float* enc, const void* user, int user_dtype, int64_t user_offset, const float* table,
                                    int64_t n_rows, const float* w3, const float* b3, const float* ln_w, const float* ln_b,
                                    float* x3, float* u3, float* out, float* mean, float* rstd, int G, int T, int C, int U,
                                    float eps, float slope, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                    uint32_t salt, void* ws, void* stream) {
    if (G <= 0) return 0;
    if (T <= 0 || (C & 15) || (U & 15) || !ws) return MOBGT_EBADDIM;
    if (user_dtype != MOBGT_I32 && user_dtype != MOBGT_I64) return MOBGT_EBADDIM;
    if (((uintptr_t)enc | (uintptr_t)table | (uintptr_t)w3 | (uintptr_t)x3 | (uintptr_t)ws) & 15) return MOBGT_EALIGN;
    const int nblk = (G + HBM - 1) / HBM;
    if (nblk > HMAX_BLOCKS) return MOBGT_EBADDIM;          // (every member of every cluster must be resident at once)
    HeadChainParams p = {};
    p.enc = enc; p.user = user; p.user_dtype = user_dtype; p.user_offset = user_offset; p.table = table; p.n_rows = n_rows;
    p.w3 = w3; p.b3 = b3; p.ln_w = ln_w; p.ln_b = ln_b; p.x3 = x3; p.u3 = u3; p.out = out; p.mean = mean; p.rstd = rstd;
    p.G = G; p.T = T; p.C = C; p.U = U; p.eps = eps; p.slope = slope;
    fill(p, dropout_p, seed, seed_dev, salt);
    uint32_t* gen = reinterpret_cast<uint32_t*>(ws);
    uint64_t* ll = reinterpret_cast<uint64_t*>(gen + HWS_GEN_INTS);
    const int W = C + U;
    const dim3 grid(8 * (W / 16) * ((nblk + 7) / 8)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (W == 320) hipLaunchKernelGGL(head_chain_fwd_kernel<320>, grid, block, 0, st, p, gen, ll);
    else if (W == 384) hipLaunchKernelGGL(head_chain_fwd_kernel<384>, grid, block, 0, st, p, gen, ll);
    else return MOBGT_EBADDIM;               // the instantiated widths: MobGT's C = 192 / 256 next to the 128-wide user embedding
    return (int)hipGetLastError();
}

extern "C" int mobgt_head_chain_bwd(const float* dout, const float* u3, const float* mean, const float* rstd, const void* user,
                                    int user_dtype, int64_t user_offset, int64_t n_rows, const float* w3, const float* ln_w,
                                    const float* ln_b, float* du3, float* denc, float* dtable, float* dgamma, float* dbeta, int G,
                                    int T, int C, int U, float eps, float slope, float dropout_p, uint64_t seed,
                                    const uint64_t* seed_dev, uint32_t salt, void* stream) {
    if (G <= 0) return 0;
    if (T <= 0 || (C & 15) || (U & 15)) return MOBGT_EBADDIM;
    if (user_dtype != MOBGT_I32 && user_dtype != MOBGT_I64) return MOBGT_EBADDIM;
    if (((uintptr_t)w3 | (uintptr_t)denc) & 15) return MOBGT_EALIGN;
    HeadChainParams p = {};
    p.dout = dout; p.u3 = const_cast<float*>(u3); p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd);
    p.user = user; p.user_dtype = user_dtype; p.user_offset = user_offset; p.n_rows = n_rows; p.w3 = w3; p.ln_w = ln_w; p.ln_b = ln_b;
    p.du3 = du3; p.denc = denc; p.dtable = dtable; p.dgamma = dgamma; p.dbeta = dbeta;
    p.G = G; p.T = T; p.C = C; p.U = U; p.eps = eps; p.slope = slope;
    fill(p, dropout_p, seed, seed_dev, salt);
    const int nblk = (G + HBM - 1) / HBM, W = C + U;
    const int n_cluster = 8 * (W / 16) * ((nblk + 7) / 8);
    const int64_t zero_items = (int64_t)G * (T - 1) * C / 4;
    int nz = (int)((zero_items + 4 * 512 - 1) / (4 * 512));     // ~4 stores per thread
    if (nz > 512) nz = 512;
    const dim3 grid(n_cluster + nz), block(512);
    hipStream_t st = (hipStream_t)stream;
    if (W == 320) hipLaunchKernelGGL(head_chain_bwd_kernel<320>, grid, block, 0, st, p, n_cluster);
    else if (W == 384) hipLaunchKernelGGL(head_chain_bwd_kernel<384>, grid, block, 0, st, p, n_cluster);
    else return MOBGT_EBADDIM;
    return (int)hipGetLastError();
}
