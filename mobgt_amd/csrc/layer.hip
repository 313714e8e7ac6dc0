// Fused elementwise / normalisation kernels of the encoder layer (graphormer/model.py:479-489,
// model_fqandtoyo.py:1731-1743): everything between the library GEMMs and the attention core.
//
//   dropout_add_ln  : x1 = x + dropout(y);  z = LayerNorm(x1)          (model.py:482-485)
//   gelu / gelu_bwd : exact-erf GELU of the FFN (model.py:398) and its backward
//   colsum          : bias gradients of the Linear layers (column sums of the output gradient)
//
// One wave owns whole rows (C <= 512 columns, lane l owns columns l, l+64, ...), so LayerNorm statistics are
// two wave reductions and -- because a lane owns the SAME columns in every row it visits -- the per-column
// gradient sums (dgamma, dbeta, dbias) accumulate in registers across rows and leave the workgroup as one
// f32 atomic per column.  Residual stream and statistics are f32; GEMM-facing tensors are f32 or bf16.
// Dropout uses the same counter hash as the attention kernels (common.h), regenerated in the backward.
#include <cstdlib>
#include "common.h"
#include "mobgt_hip.h"
#include "front_body.h"
#include "pack_body.h"
#include "hop_body.h"

namespace {

constexpr int MAXC_PER_LANE = 8;     // C <= 512

template <typename T> __device__ __forceinline__ float ldf(const T* p, int64_t i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, int64_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void stf(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, int64_t i, float v) { p[i] = (bf16_t)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct LnParams {
    const float* x;        // residual in  [R,C] f32
    const void* y;         // branch output [R,C] (A) or null
    float* x1;             // residual out [R,C] f32 (may alias x when y == null)
    const float *w, *b;    // LayerNorm affine or null
    void* z;               // LN output (A) or null
    float* z32;            // LN output f32 or null
    float *mean, *rstd;    // [R]
    // backward
    const void* dz;        // grad of LN output (A) or null
    const float* dz32;     // grad of LN output f32 or null (added to dz)
    const float* dres;     // grad arriving at x1 from downstream, f32 or null
    float* dx1;            // out: total grad at x1 (f32) = grad of the residual input
    void* dy;              // out: grad of the branch output y (A) or null
    float *dgamma, *dbeta, *dbias;   // [C] accumulated (atomics) or null
    int64_t R;
    int C;
    float inv_keep;
    uint32_t thr;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt;
    int rows_per_wg;
};

template <typename TA>
__global__ __launch_bounds__(256) void dropout_add_ln_fwd_kernel(const LnParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    const float invC = 1.f / (float)p.C;
    for (int rr = wave; rr < p.rows_per_wg; rr += 4) {
        const int64_t r = (int64_t)blockIdx.x * p.rows_per_wg + rr;
        if (r >= p.R) break;
        float v[MAXC_PER_LANE];
        const uint32_t rowh = p.thr ? dropout_row_hash(seed, (uint32_t)r ^ p.salt) : 0u;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC_PER_LANE; ++k) {
            const int c = lane + 64 * k;
            float t = 0.f;
            if (c < p.C) {
                t = p.x[r * p.C + c];
                if (p.y) {
                    float yv = ldf<TA>(reinterpret_cast<const TA*>(p.y), r * p.C + c);
                    if (p.thr) yv = dropout_bits16(seed, rowh, (uint32_t)c) >= p.thr ? yv * p.inv_keep : 0.f;
                    t += yv;
                    p.x1[r * p.C + c] = t;
                }
            }
            v[k] = t;
            s += t;
        }
        if (!p.w) continue;
        const float mu = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC_PER_LANE; ++k) {
            const int c = lane + 64 * k;
            const float d = c < p.C ? v[k] - mu : 0.f;
            q += d * d;
        }
        const float rs = rsqrtf(wave_sum(q) * invC + 1e-5f);
        if (lane == 0) { p.mean[r] = mu; p.rstd[r] = rs; }
#pragma unroll
        for (int k = 0; k < MAXC_PER_LANE; ++k) {
            const int c = lane + 64 * k;
            if (c < p.C) {
                const float o = (v[k] - mu) * rs * p.w[c] + p.b[c];
                if (p.z) stf<TA>(reinterpret_cast<TA*>(p.z), r * p.C + c, o);
                if (p.z32) p.z32[r * p.C + c] = o;
            }
        }
    }
}

template <typename TA>
__global__ __launch_bounds__(256) void dropout_add_ln_bwd_kernel(const LnParams p) {
    __shared__ float red[3][4][64 * MAXC_PER_LANE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    const float invC = 1.f / (float)p.C;
    float ag[MAXC_PER_LANE], ab[MAXC_PER_LANE], ay[MAXC_PER_LANE];
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k) { ag[k] = 0.f; ab[k] = 0.f; ay[k] = 0.f; }
    for (int rr = wave; rr < p.rows_per_wg; rr += 4) {
        const int64_t r = (int64_t)blockIdx.x * p.rows_per_wg + rr;
        if (r >= p.R) break;
        float dxv[MAXC_PER_LANE];
#pragma unroll
        for (int k = 0; k < MAXC_PER_LANE; ++k) dxv[k] = 0.f;
        if (p.w) {
            const float mu = p.mean[r], rs = p.rstd[r];
            float xh[MAXC_PER_LANE], g[MAXC_PER_LANE];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < MAXC_PER_LANE; ++k) {
                const int c = lane + 64 * k;
                xh[k] = 0.f; g[k] = 0.f;
                if (c < p.C) {
                    float d = 0.f;
                    if (p.dz) d += ldf<TA>(reinterpret_cast<const TA*>(p.dz), r * p.C + c);
                    if (p.dz32) d += p.dz32[r * p.C + c];
                    xh[k] = (p.x1[r * p.C + c] - mu) * rs;
                    g[k] = d * p.w[c];
                    ag[k] += d * xh[k];
                    ab[k] += d;
                    s1 += g[k];
                    s2 += g[k] * xh[k];
                }
            }
            s1 = wave_sum(s1) * invC;
            s2 = wave_sum(s2) * invC;
#pragma unroll
            for (int k = 0; k < MAXC_PER_LANE; ++k) dxv[k] = rs * (g[k] - s1 - xh[k] * s2);
        }
        const uint32_t rowh = p.thr ? dropout_row_hash(seed, (uint32_t)r ^ p.salt) : 0u;
#pragma unroll
        for (int k = 0; k < MAXC_PER_LANE; ++k) {
            const int c = lane + 64 * k;
            if (c < p.C) {
                float t = dxv[k];
                if (p.dres) t += p.dres[r * p.C + c];
                if (p.dx1) p.dx1[r * p.C + c] = t;
                if (p.dy) {
                    float yv = t;
                    if (p.thr) yv = dropout_bits16(seed, rowh, (uint32_t)c) >= p.thr ? t * p.inv_keep : 0.f;
                    stf<TA>(reinterpret_cast<TA*>(p.dy), r * p.C + c, yv);
                    ay[k] += yv;
                }
            }
        }
    }
    // column sums: registers -> LDS across the 4 waves -> one atomic per column per workgroup
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k) {
        red[0][wave][lane + 64 * k] = ag[k];
        red[1][wave][lane + 64 * k] = ab[k];
        red[2][wave][lane + 64 * k] = ay[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += 256) {
        if (p.dgamma) {
            atomicAdd(&p.dgamma[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
            atomicAdd(&p.dbeta[c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
        }
        if (p.dbias) atomicAdd(&p.dbias[c], red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c]);
    }
}

// ---- 16-byte-lane variants (C % 4 == 0) -------------------------------------------------------------------------
// Same maths and the same dropout bits as the kernels above, but lane l owns the FOUR consecutive columns
// 256*k + 4*l .. +3 (k = 0, 1), moved as one 16-byte (f32) / 8-byte (bf16) access, and a wave works on TWO rows
// at a time so that their loads and their two reduction chains overlap.  One-row-at-a-time with 4-byte lanes ran
// at 14-16 % of HBM speed at R = 12 560 (30 us forward / 45 us backward for 39 / 51 MB).
constexpr int V4_CH = 2;                 // column chunks of 256: C <= 512

template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    v[0] = bf16_lo(a.x); v[1] = bf16_hi(a.x); v[2] = bf16_lo(a.y); v[3] = bf16_hi(a.y);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(p) = o;
}

// U rows per wave in flight per sweep: 2 for short batches, 8 for long ones (at R = 12 560 a workgroup owns ~25 rows:
// with 2 per wave that was four dependent sweeps of loads)
template <typename TA, int U>
__global__ __launch_bounds__(256) void dropout_add_ln_fwd_v4_kernel(const LnParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    const float invC = 1.f / (float)p.C;
    const TA* Y = reinterpret_cast<const TA*>(p.y);
    TA* Z = reinterpret_cast<TA*>(p.z);
    for (int rr = wave; rr < p.rows_per_wg; rr += 4 * U) {
        int64_t r[U];
        bool on[U];
        float v[U][V4_CH][4];
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = (int64_t)blockIdx.x * p.rows_per_wg + rr + 4 * u;
            on[u] = rr + 4 * u < p.rows_per_wg && r[u] < p.R;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t rowh = (p.thr && on[u]) ? dropout_row_hash(seed, (uint32_t)r[u] ^ p.salt) : 0u;
#pragma unroll
            for (int k = 0; k < V4_CH; ++k) {
                const int c = 256 * k + 4 * lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[u][k][i] = 0.f;
                if (on[u] && c < p.C) {
                    ld4<float>(p.x + r[u] * p.C + c, v[u][k]);
                    if (Y) {
                        float yv[4];
                        ld4<TA>(Y + r[u] * p.C + c, yv);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (p.thr) yv[i] = dropout_bits16(seed, rowh, (uint32_t)(c + i)) >= p.thr ? yv[i] * p.inv_keep : 0.f;
                            v[u][k][i] += yv[i];
                        }
                        st4<float>(p.x1 + r[u] * p.C + c, v[u][k]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) s[u] += v[u][k][i];
            }
        }
        if (!p.w) continue;
        float mu[U], q[U], rs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) q[u] = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) mu[u] = wave_sum(s[u]) * invC;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < V4_CH; ++k)
                if (256 * k + 4 * lane < p.C) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float d = v[u][k][i] - mu[u]; q[u] += d * d; }
                }
#pragma unroll
        for (int u = 0; u < U; ++u) rs[u] = rsqrtf(wave_sum(q[u]) * invC + 1e-5f);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!on[u]) continue;
            if (lane == 0) { p.mean[r[u]] = mu[u]; p.rstd[r[u]] = rs[u]; }
#pragma unroll
            for (int k = 0; k < V4_CH; ++k) {
                const int c = 256 * k + 4 * lane;
                if (c < p.C) {
                    float w4[4], b4[4], o[4];
                    ld4<float>(p.w + c, w4);
                    ld4<float>(p.b + c, b4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (v[u][k][i] - mu[u]) * rs[u] * w4[i] + b4[i];
                    if (Z) st4<TA>(Z + r[u] * p.C + c, o);
                    if (p.z32) st4<float>(p.z32 + r[u] * p.C + c, o);
                }
            }
        }
    }
}

// NWV waves per workgroup: 4, or 16 for long batches -- every workgroup ends with one f32 atomic per column onto the SAME
// addresses, ~75 ns each in a row, so at R = 12 560 the 256-512 four-wave workgroups the rows need to be in flight spent
// ~20 of their 27 us queueing there: 64 workgroups of 16 waves keep as many rows in flight with a quarter of the atomics.
template <typename TA, int U, int NWV>
__global__ __launch_bounds__(NWV * 64) void dropout_add_ln_bwd_v4_kernel(const LnParams p) {
    __shared__ float red[3][NWV][256 * V4_CH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    const float invC = 1.f / (float)p.C;
    const TA* DZ = reinterpret_cast<const TA*>(p.dz);
    TA* DY = reinterpret_cast<TA*>(p.dy);
    float ag[V4_CH][4], ab[V4_CH][4], ay[V4_CH][4];
#pragma unroll
    for (int k = 0; k < V4_CH; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) { ag[k][i] = 0.f; ab[k][i] = 0.f; ay[k][i] = 0.f; }
    for (int rr = wave; rr < p.rows_per_wg; rr += NWV * U) {
        int64_t r[U];
        bool on[U];
        float xh[U][V4_CH][4], gg[U][V4_CH][4], dxv[U][V4_CH][4];
        float s1[U], s2[U], rs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { s1[u] = 0.f; s2[u] = 0.f; rs[u] = 0.f; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = (int64_t)blockIdx.x * p.rows_per_wg + rr + NWV * u;
            on[u] = rr + NWV * u < p.rows_per_wg && r[u] < p.R;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < V4_CH; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) { xh[u][k][i] = 0.f; gg[u][k][i] = 0.f; dxv[u][k][i] = 0.f; }
        if (p.w) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!on[u]) continue;
                const float mu = p.mean[r[u]];
                rs[u] = p.rstd[r[u]];
#pragma unroll
                for (int k = 0; k < V4_CH; ++k) {
                    const int c = 256 * k + 4 * lane;
                    if (c < p.C) {
                        float d[4] = {0.f, 0.f, 0.f, 0.f}, t[4], x4[4], w4[4];
                        if (DZ) { ld4<TA>(DZ + r[u] * p.C + c, t); for (int i = 0; i < 4; ++i) d[i] += t[i]; }
                        if (p.dz32) { ld4<float>(p.dz32 + r[u] * p.C + c, t); for (int i = 0; i < 4; ++i) d[i] += t[i]; }
                        ld4<float>(p.x1 + r[u] * p.C + c, x4);
                        ld4<float>(p.w + c, w4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            xh[u][k][i] = (x4[i] - mu) * rs[u];
                            gg[u][k][i] = d[i] * w4[i];
                            ag[k][i] += d[i] * xh[u][k][i];
                            ab[k][i] += d[i];
                            s1[u] += gg[u][k][i];
                            s2[u] += gg[u][k][i] * xh[u][k][i];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { s1[u] = wave_sum(s1[u]) * invC; s2[u] = wave_sum(s2[u]) * invC; }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < V4_CH; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) dxv[u][k][i] = rs[u] * (gg[u][k][i] - s1[u] - xh[u][k][i] * s2[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!on[u]) continue;
            const uint32_t rowh = p.thr ? dropout_row_hash(seed, (uint32_t)r[u] ^ p.salt) : 0u;
#pragma unroll
            for (int k = 0; k < V4_CH; ++k) {
                const int c = 256 * k + 4 * lane;
                if (c < p.C) {
                    float t[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[i] = dxv[u][k][i];
                    if (p.dres) {
                        float d4[4];
                        ld4<float>(p.dres + r[u] * p.C + c, d4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) t[i] += d4[i];
                    }
                    if (p.dx1) st4<float>(p.dx1 + r[u] * p.C + c, t);
                    if (DY) {
                        float yv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            yv[i] = t[i];
                            if (p.thr) yv[i] = dropout_bits16(seed, rowh, (uint32_t)(c + i)) >= p.thr ? t[i] * p.inv_keep : 0.f;
                            ay[k][i] += yv[i];
                        }
                        st4<TA>(DY + r[u] * p.C + c, yv);
                    }
                }
            }
        }
    }
    // column sums: registers -> LDS across the waves -> one atomic per column per workgroup
#pragma unroll
    for (int k = 0; k < V4_CH; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[0][wave][256 * k + 4 * lane + i] = ag[k][i];
            red[1][wave][256 * k + 4 * lane + i] = ab[k][i];
            red[2][wave][256 * k + 4 * lane + i] = ay[k][i];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += NWV * 64) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { t0 += red[0][w][c]; t1 += red[1][w][c]; t2 += red[2][w][c]; }
        if (p.dgamma) {
            atomicAdd(&p.dgamma[c], t0);
            atomicAdd(&p.dbeta[c], t1);
        }
        if (p.dbias) atomicAdd(&p.dbias[c], t2);
    }
}

// ------------------------------------------------------------------------------------------- GELU
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float u) {
    const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * u * u);
    return cdf + u * pdf;
}

template <typename TA>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const TA* __restrict__ u, TA* __restrict__ h, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i + k < n) stf<TA>(h, i + k, gelu_f(ldf<TA>(u, i + k)));
}

// du = dh * gelu'(u); dbias[c] += colsum(du).  Column-major ownership: thread t of the workgroup owns the
// columns t, t+256, ... and walks ROWS rows.
template <typename TA, bool GELU>
__global__ __launch_bounds__(256) void colsum_kernel(const TA* __restrict__ dh, const TA* __restrict__ u, TA* __restrict__ du,
                                                     float* __restrict__ dbias, int64_t R, int C, int rows_per_wg) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    for (int c = blockIdx.y * 256 + threadIdx.x; c < C; c += 256 * gridDim.y) {    // grid.y = column chunks of 256
        float acc = 0.f;
        for (int rr = 0; rr < rows_per_wg; ++rr) {
            const int64_t r = r0 + rr;
            if (r >= R) break;
            float g = ldf<TA>(dh, r * C + c);
            if (GELU) {
                g *= gelu_grad(ldf<TA>(u, r * C + c));
                stf<TA>(du, r * C + c, g);
            }
            acc += g;
        }
        if (dbias) atomicAdd(&dbias[c], acc);
    }
}

// C % 4 == 0: a workgroup covers 256 columns x rows_per_wg rows; lane l owns the four consecutive columns 4l..4l+3
// (8- / 16-byte accesses), the four waves take different rows (two in flight each), LDS combines them and the
// workgroup leaves one atomic per column.
template <typename TA, bool GELU, int U, int NWV>
__global__ __launch_bounds__(NWV * 64) void colsum_v4_kernel(const TA* __restrict__ dh, const TA* __restrict__ u, TA* __restrict__ du,
                                                        float* __restrict__ dbias, int64_t R, int C, int rows_per_wg) {
    __shared__ float red[NWV][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    const int c = 256 * blockIdx.y + 4 * lane;
    const int nrow = (int)min((int64_t)rows_per_wg, R - r0);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        // U rows per wave in flight per sweep (2 for short batches, 8 for long ones)
        for (int rr = wave; rr < nrow; rr += NWV * U) {
            float g[U][4], uu[U][4];
#pragma unroll
            for (int k = 0; k < U; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) g[k][i] = 0.f;
                if (rr + NWV * k < nrow) {
                    ld4<TA>(dh + (r0 + rr + NWV * k) * C + c, g[k]);
                    if (GELU) ld4<TA>(u + (r0 + rr + NWV * k) * C + c, uu[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (GELU && rr + NWV * k < nrow) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) g[k][i] *= gelu_grad(uu[k][i]);
                    st4<TA>(du + (r0 + rr + NWV * k) * C + c, g[k]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += g[k][i];
            }
        }
    }
    if (!dbias) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][4 * lane + i] = acc[i];
    __syncthreads();
    const int cc = 256 * blockIdx.y + threadIdx.x;
    if (threadIdx.x < 256 && cc < C) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) t += red[w][threadIdx.x];
        atomicAdd(&dbias[cc], t);
    }
}

template <typename TA>
__global__ __launch_bounds__(256) void gelu_fwd_v4_kernel(const TA* __restrict__ u, TA* __restrict__ h, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float v[4];
    ld4<TA>(u + i, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = gelu_f(v[k]);
    st4<TA>(h + i, v);
}

int pick_rows(int64_t R) {
    // 4 rows (one per wave) per workgroup while that gives <= 512 workgroups, more rows beyond: every
    // workgroup ends with one f32 atomic per column onto the SAME C addresses, and that contention (not the
    // streaming) set the kernel time at R = 12.5k rows (measured 54 us with 3140 workgroups)
    // Few rows: aim at ~128 workgroups (8 rows each, two per wave: one sweep) rather than one row per wave -- measured
    // on the whole step at R = 608: 12 rows (51 workgroups) 13.89 k check-ins/s, 8 rows (76) 14.05 k, 4 rows (152
    // workgroups x 576 atomics on the same addresses) 13.85 k.
    static const int cap = 256;
    int64_t wgs = R / 16;
    wgs = wgs < 128 ? 128 : (wgs > cap ? cap : wgs);     // (256 vs 512 at R = 12 560: 11.33 vs 11.45 ms per S-BIG step)
    int rows = (int)((R + wgs - 1) / wgs);
    rows = (rows + 3) / 4 * 4;
    return rows < 4 ? 4 : rows;
}

// kernels without a per-workgroup atomic tail: one row per wave while that gives <= 512 workgroups
int pick_rows_stream(int64_t R) {
    static const int cap = 4096;
    int rows = (int)((R + cap - 1) / cap);
    rows = (rows + 3) / 4 * 4;
    return rows < 4 ? 4 : rows;
}

void set_drop(LnParams& p, float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt) {
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
}

}  // namespace

extern "C" int mobgt_dropout_add_ln_fwd(const float* x, const void* y, float* x1, const float* ln_w, const float* ln_b,
                                        void* z, float* z32, float* mean, float* rstd, int64_t R, int C,
                                        float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt,
                                        int act_dtype, void* stream) {
    if (R <= 0) return 0;
    if (C <= 0 || C > 64 * MAXC_PER_LANE) return MOBGT_EBADDIM;
    LnParams p = {};
    p.x = x; p.y = y; p.x1 = x1; p.w = ln_w; p.b = ln_b; p.z = z; p.z32 = z32; p.mean = mean; p.rstd = rstd;
    p.R = R; p.C = C;
    set_drop(p, y ? dropout_p : 0.f, seed, seed_dev, salt);
    p.rows_per_wg = pick_rows_stream(R);
    const dim3 grid((unsigned)((R + p.rows_per_wg - 1) / p.rows_per_wg)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = C % 4 == 0 && C <= 256 * V4_CH;
    if (act_dtype == MOBGT_F32) {
        if (v4 && p.rows_per_wg > 8) hipLaunchKernelGGL((dropout_add_ln_fwd_v4_kernel<float, 8>), grid, block, 0, st, p);
        else if (v4) hipLaunchKernelGGL((dropout_add_ln_fwd_v4_kernel<float, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL(dropout_add_ln_fwd_kernel<float>, grid, block, 0, st, p);
    } else if (act_dtype == MOBGT_BF16) {
        if (v4 && p.rows_per_wg > 8) hipLaunchKernelGGL((dropout_add_ln_fwd_v4_kernel<bf16_t, 8>), grid, block, 0, st, p);
        else if (v4) hipLaunchKernelGGL((dropout_add_ln_fwd_v4_kernel<bf16_t, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL(dropout_add_ln_fwd_kernel<bf16_t>, grid, block, 0, st, p);
    } else return MOBGT_EDTYPE;
    return (int)hipGetLastError();
}

extern "C" int mobgt_dropout_add_ln_bwd(const void* dz, const float* dz32, const float* dres, const float* x1,
                                        const float* mean, const float* rstd, const float* ln_w, float* dx1, void* dy,
                                        float* dgamma, float* dbeta, float* dbias, int64_t R, int C, float dropout_p,
                                        uint64_t seed, const uint64_t* seed_dev, uint32_t salt, int act_dtype,
                                        void* stream) {
    if (R <= 0) return 0;
    if (C <= 0 || C > 64 * MAXC_PER_LANE) return MOBGT_EBADDIM;
    LnParams p = {};
    p.dz = dz; p.dz32 = dz32; p.dres = dres; p.x1 = const_cast<float*>(x1); p.mean = const_cast<float*>(mean);
    p.rstd = const_cast<float*>(rstd); p.w = ln_w; p.dx1 = dx1; p.dy = dy; p.dgamma = dgamma; p.dbeta = dbeta;
    p.dbias = dbias; p.R = R; p.C = C;
    set_drop(p, dy ? dropout_p : 0.f, seed, seed_dev, salt);
    const bool v4 = C % 4 == 0 && C <= 256 * V4_CH;
    static const int64_t wide_from = 4096;
    const bool wide = v4 && R >= wide_from;
    static const int wide_wgs = 256;    // S-BIG step: 64 / 128 / 256 / 384 / 512 -> 11.22 / 10.95 / 10.58-10.70 / 10.94 / 10.79 ms (narrow form: 11.06)
    p.rows_per_wg = wide ? (int)((R + wide_wgs - 1) / wide_wgs) : pick_rows(R);
    if (wide) p.rows_per_wg = (p.rows_per_wg + 15) / 16 * 16;
    const dim3 grid((unsigned)((R + p.rows_per_wg - 1) / p.rows_per_wg)), block(256), wgrid = grid;
    hipStream_t st = (hipStream_t)stream;
    if (act_dtype == MOBGT_F32) {
        if (v4 && wide) hipLaunchKernelGGL((dropout_add_ln_bwd_v4_kernel<float, 2, 16>), wgrid, dim3(1024), 0, st, p);
        else if (v4) hipLaunchKernelGGL((dropout_add_ln_bwd_v4_kernel<float, 2, 4>), grid, block, 0, st, p);
        else hipLaunchKernelGGL(dropout_add_ln_bwd_kernel<float>, grid, block, 0, st, p);
    } else if (act_dtype == MOBGT_BF16) {
        if (v4 && wide) hipLaunchKernelGGL((dropout_add_ln_bwd_v4_kernel<bf16_t, 2, 16>), wgrid, dim3(1024), 0, st, p);
        else if (v4) hipLaunchKernelGGL((dropout_add_ln_bwd_v4_kernel<bf16_t, 2, 4>), grid, block, 0, st, p);
        else hipLaunchKernelGGL(dropout_add_ln_bwd_kernel<bf16_t>, grid, block, 0, st, p);
    } else return MOBGT_EDTYPE;
    return (int)hipGetLastError();
}

extern "C" int mobgt_gelu_fwd(const void* u, void* h, int64_t n, int act_dtype, void* stream) {
    if (n <= 0) return 0;
    const dim3 grid((unsigned)((n + 1023) / 1024)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = n % 4 == 0 && (((uintptr_t)u | (uintptr_t)h) & 15) == 0;
    if (act_dtype == MOBGT_F32) {
        if (v4) hipLaunchKernelGGL(gelu_fwd_v4_kernel<float>, grid, block, 0, st, (const float*)u, (float*)h, n);
        else hipLaunchKernelGGL(gelu_fwd_kernel<float>, grid, block, 0, st, (const float*)u, (float*)h, n);
    } else if (act_dtype == MOBGT_BF16) {
        if (v4) hipLaunchKernelGGL(gelu_fwd_v4_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)u, (bf16_t*)h, n);
        else hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)u, (bf16_t*)h, n);
    } else return MOBGT_EDTYPE;
    return (int)hipGetLastError();
}

extern "C" int mobgt_gelu_bwd_colsum(const void* dh, const void* u, void* du, float* dbias, int64_t R, int C,
                                     int act_dtype, void* stream) {
    if (R <= 0 || C <= 0) return 0;
    static const int64_t wide_from = 4096;
    static const int wide_wgs = 128;
    const bool wide = C % 4 == 0 && R >= wide_from;
    const int rows = wide ? (int)((R + wide_wgs - 1) / wide_wgs) : pick_rows(R);
    const bool v4 = C % 4 == 0 && (((uintptr_t)dh | (uintptr_t)u | (uintptr_t)du) & 15) == 0;
    const dim3 grid((unsigned)((R + rows - 1) / rows), (unsigned)((C + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (act_dtype == MOBGT_F32) {
        if (v4 && wide) hipLaunchKernelGGL((colsum_v4_kernel<float, true, 2, 16>), grid, dim3(1024), 0, st, (const float*)dh, (const float*)u, (float*)du, dbias, R, C, rows);
        else if (v4) hipLaunchKernelGGL((colsum_v4_kernel<float, true, 2, 4>), grid, block, 0, st, (const float*)dh, (const float*)u, (float*)du, dbias, R, C, rows);
        else hipLaunchKernelGGL((colsum_kernel<float, true>), grid, block, 0, st, (const float*)dh, (const float*)u, (float*)du, dbias, R, C, rows);
    } else if (act_dtype == MOBGT_BF16) {
        if (v4 && wide) hipLaunchKernelGGL((colsum_v4_kernel<bf16_t, true, 2, 16>), grid, dim3(1024), 0, st, (const bf16_t*)dh, (const bf16_t*)u, (bf16_t*)du, dbias, R, C, rows);
        else if (v4) hipLaunchKernelGGL((colsum_v4_kernel<bf16_t, true, 2, 4>), grid, block, 0, st, (const bf16_t*)dh, (const bf16_t*)u, (bf16_t*)du, dbias, R, C, rows);
        else hipLaunchKernelGGL((colsum_kernel<bf16_t, true>), grid, block, 0, st, (const bf16_t*)dh, (const bf16_t*)u, (bf16_t*)du, dbias, R, C, rows);
    } else return MOBGT_EDTYPE;
    return (int)hipGetLastError();
}

extern "C" int mobgt_colsum(const void* g, float* out, int64_t R, int C, int act_dtype, void* stream) {
    if (R <= 0 || C <= 0) return 0;
    static const int64_t wide_from = 4096;
    static const int wide_wgs = 128;
    const bool wide = C % 4 == 0 && R >= wide_from;
    const int rows = wide ? (int)((R + wide_wgs - 1) / wide_wgs) : pick_rows(R);
    const bool v4 = C % 4 == 0 && ((uintptr_t)g & 15) == 0;
    const dim3 grid((unsigned)((R + rows - 1) / rows), (unsigned)((C + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (act_dtype == MOBGT_F32) {
        if (v4 && wide) hipLaunchKernelGGL((colsum_v4_kernel<float, false, 2, 16>), grid, dim3(1024), 0, st, (const float*)g, (const float*)nullptr, (float*)nullptr, out, R, C, rows);
        else if (v4) hipLaunchKernelGGL((colsum_v4_kernel<float, false, 2, 4>), grid, block, 0, st, (const float*)g, (const float*)nullptr, (float*)nullptr, out, R, C, rows);
        else hipLaunchKernelGGL((colsum_kernel<float, false>), grid, block, 0, st, (const float*)g, (const float*)nullptr, (float*)nullptr, out, R, C, rows);
    } else if (act_dtype == MOBGT_BF16) {
        if (v4 && wide) hipLaunchKernelGGL((colsum_v4_kernel<bf16_t, false, 2, 16>), grid, dim3(1024), 0, st, (const bf16_t*)g, (const bf16_t*)nullptr, (bf16_t*)nullptr, out, R, C, rows);
        else if (v4) hipLaunchKernelGGL((colsum_v4_kernel<bf16_t, false, 2, 4>), grid, block, 0, st, (const bf16_t*)g, (const bf16_t*)nullptr, (bf16_t*)nullptr, out, R, C, rows);
        else hipLaunchKernelGGL((colsum_kernel<bf16_t, false>), grid, block, 0, st, (const bf16_t*)g, (const bf16_t*)nullptr, (bf16_t*)nullptr, out, R, C, rows);
    } else return MOBGT_EDTYPE;
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// GradientTailLoss (graphormer/model_fqandtoyo.py:545-550), forward value and d(loss)/d(logits) in one pass:
//   p = sigmoid(z);  loss = mean( -alpha*(1-p)*onehot*log(p) - (1-onehot)*p*log(1-p) )      (beta = k = 1)
// The reference evaluates it as ~12 elementwise kernels forward and ~15 backward over [G, P+1].
namespace {
constexpr int GTL_MAX_BLOCKS = 1024;
// per-workgroup partial sums and the arrival ticket of the last-workgroup reduction below (stream-ordered use)
__device__ float gtl_partial[GTL_MAX_BLOCKS];
__device__ unsigned int gtl_ticket = 0;

// *loss is OVERWRITTEN by the last workgroup to arrive (sum of the per-workgroup partials in index order): no
// zero-fill of *loss before the launch.  A hipMemsetAsync(loss, 0, 4) node in front of an atomicAdd version was
// observed not to take effect inside replayed hipGraphs (the scalar kept a stale value and the loss read
// "stale + loss", found in the 2-rank run), so the kernel no longer depends on any prior state of *loss.
__global__ __launch_bounds__(256) void gtl_kernel(const float* __restrict__ z, const int64_t* __restrict__ target,
                                                  int64_t target_offset, float* __restrict__ dz, float* __restrict__ loss,
                                                  int64_t G, int64_t V, float alpha) {
    const int64_t n = G * V;
    const float inv_n = 1.f / (float)n;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i / V, c = i - g * V;
        const float x = z[i];
        const float p = 1.f / (1.f + __expf(-x));
        const float q = 1.f - p;
        float l, d;
        if (target[g] + target_offset == c) {
            const float lp = logf(p);
            l = -alpha * q * lp;
            d = alpha * p * q * lp - alpha * q * q;
        } else {
            const float lq = logf(q);
            l = -p * lq;
            d = -p * q * lq + p * p;
        }
        acc += l;
        dz[i] = d * inv_n;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ float part[4];
    __shared__ bool last;
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&gtl_partial[blockIdx.x], part[0] + part[1] + part[2] + part[3], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned int t = __hip_atomic_fetch_add(&gtl_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    float tot = 0.f;
    for (unsigned int b = threadIdx.x; b < gridDim.x; b += 256)
        tot += __hip_atomic_load(&gtl_partial[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        *loss = (part[0] + part[1] + part[2] + part[3]) * inv_n;
        __hip_atomic_store(&gtl_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
}  // namespace

extern "C" int mobgt_gradient_tail_loss(const float* logits, const int64_t* targets, int64_t target_offset, float* dlogits,
                                        float* loss, int64_t G, int64_t V, float alpha, void* stream) {
    if (G <= 0 || V <= 0) return MOBGT_EBADDIM;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = G * V;
    const unsigned blocks = (unsigned)((n + 1023) / 1024 < GTL_MAX_BLOCKS ? (n + 1023) / 1024 : GTL_MAX_BLOCKS);
    hipLaunchKernelGGL(gtl_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, logits, targets, target_offset, dlogits, loss,
                       G, V, alpha);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The stock variant's encoder input (model.py:193-205): row (g, 0) = graph_token, row (g, 1 + n) = atom_encoder[x[g,n]] +
// in_degree_encoder[deg[g,n]] + out_degree_encoder[deg'[g,n]], then input_dropout over the whole [G, T = N+1, C] tensor -- one
// launch each way instead of gather + cat + dropout (+ index casts) / dropout + reduce + copy + scatter.  The dropout mask is
// mobgt_dropout's for the same salt: element (row g T + t, column c).  A negative index contributes nothing; row `skip` of
// a table (padding_idx) is read but receives no gradient.  Backward: f32 atomics into ZEROED (or sink) buffers.
namespace {
struct StockTokParams {
    const void *x, *din, *dout;            // [G, N] indices (x: idx_dtype; the two degree tensors: deg_dtype)
    int idx_dtype, deg_dtype;
    const float *atom, *indeg, *outdeg, *gtok;   // [*, C] tables, [C] graph token
    float *y;                              // [G, T, C]
    const float* dy;                       // backward
    float *d_atom, *d_indeg, *d_outdeg, *d_gtok;
    int G, N, C;
    int64_t n_atom, n_in, n_out, skip;
    uint32_t thr; float inv_keep; uint64_t seed; const uint64_t* seed_dev; uint32_t salt;
};
__device__ __forceinline__ int64_t st_idx(const void* p, int dt, int64_t i) {
    return dt == MOBGT_I64 ? reinterpret_cast<const int64_t*>(p)[i]
         : dt == MOBGT_I32 ? (int64_t)reinterpret_cast<const int32_t*>(p)[i] : (int64_t)reinterpret_cast<const int16_t*>(p)[i];
}
// virtual block `bid` of `nb` (256 threads each) of the grid-stride loop over the [G, T, C / 4] pieces
template <bool BWD>
__device__ __forceinline__ void stock_tokens_body(const StockTokParams& p, const int bid, const int nb) {
    const int T = p.N + 1, c4 = p.C / 4;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    const int64_t total = (int64_t)p.G * T * c4;
    for (int64_t e = (int64_t)bid * 256 + threadIdx.x; e < total; e += (int64_t)nb * 256) {
        const int64_t r = e / c4;
        const int c = (int)(e - r * c4) * 4;
        const int g = (int)(r / T), t = (int)(r - (int64_t)g * T);
        float keep[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.thr) {
            const uint32_t rowh = dropout_row_hash(seed, (uint32_t)r ^ p.salt);
#pragma unroll
            for (int i = 0; i < 4; ++i) keep[i] = dropout_bits16(seed, rowh, (uint32_t)(c + i)) >= p.thr ? p.inv_keep : 0.f;
        }
        int64_t ia = -1, ii = -1, io = -1;
        if (t > 0) {
            const int64_t at = (int64_t)g * p.N + (t - 1);
            ia = st_idx(p.x, p.idx_dtype, at); ii = st_idx(p.din, p.deg_dtype, at); io = st_idx(p.dout, p.deg_dtype, at);
            if (ia >= p.n_atom) ia = -1;
            if (ii >= p.n_in) ii = -1;
            if (io >= p.n_out) io = -1;
        }
        if (!BWD) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            auto add = [&](const float* tab, int64_t i) {
                if (i < 0) return;
                const float4 a = *reinterpret_cast<const float4*>(tab + i * p.C + c);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            };
            if (t == 0) v = *reinterpret_cast<const float4*>(p.gtok + c);
            else { add(p.atom, ia); add(p.indeg, ii); add(p.outdeg, io); }
            *reinterpret_cast<float4*>(p.y + r * p.C + c) = make_float4(v.x * keep[0], v.y * keep[1], v.z * keep[2], v.w * keep[3]);
        } else {
            const float4 d = *reinterpret_cast<const float4*>(p.dy + r * p.C + c);
            const float dv[4] = {d.x * keep[0], d.y * keep[1], d.z * keep[2], d.w * keep[3]};
            auto scat = [&](float* tab, int64_t i) {
                if (!tab || i < 0 || i == p.skip) return;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (dv[q] != 0.f) atomicAdd(tab + i * p.C + c + q, dv[q]);
            };
            if (t == 0) {
                if (p.d_gtok) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (dv[q] != 0.f) atomicAdd(p.d_gtok + c + q, dv[q]);
                }
            } else { scat(p.d_atom, ia); scat(p.d_indeg, ii); scat(p.d_outdeg, io); }
        }
    }
}
template <bool BWD>
__global__ __launch_bounds__(256) void stock_tokens_kernel(const StockTokParams p) {
    stock_tokens_body<BWD>(p, (int)blockIdx.x, (int)gridDim.x);
}
// The stock step's three front launches as ONE grid (round 4): the encoder input, the hop table's forward and the MFMA-order
// weight pack are independent of one another (5 + 5 + 8 us alone, each mostly launch ramp); the bias build that reads the hop
// table and the first layer that reads the other two follow on the stream.  Blocks [0, tok_blocks) run the token rows,
// the next hop_blocks the hop table, the rest one pack block each.
__global__ __launch_bounds__(256) void stock_front_kernel(const StockTokParams p, int tok_blocks, const mobgt_front::HopFwd hf,
                                                          int hop_blocks, const mobgt_pack::PackJobs jobs, int njobs, int nvb) {
    const int b = (int)blockIdx.x;
    if (b < tok_blocks) stock_tokens_body<false>(p, b, tok_blocks);
    else if (b < tok_blocks + hop_blocks) mobgt_front::hop_table_fwd_body(hf, b - tok_blocks);
    else mobgt_pack::pack_blocks<1>(jobs, njobs, b - tok_blocks - hop_blocks, 0, nvb);
}
// ... and two of the stock step's tail launches (round 4): the backward of the encoder input (scatter of d(tokens) into the three
// tables) and the hop table's backward (H = 8; csrc/hop_body.h) read different gradients and write different tables -- 12.9 + 9.6
// us alone at E = 1 537 edge ids, which is too wide for the grouped weight-gradient launch's hop slot (E <= 256).  Blocks
// [0, hop_blocks) run the hop body (the long ones first), the rest the token rows; the dynamic LDS is the hop body's.
__global__ __launch_bounds__(256) void stock_tail_kernel(const StockTokParams p, int tok_blocks, const mobgt_hop::HopBwd hp, int hop_blocks) {
    extern __shared__ __attribute__((aligned(16))) float tail_sm[];
    const int b = (int)blockIdx.x;
    if (b < hop_blocks) mobgt_hop::hop_table_bwd8_body(hp, b, tail_sm);
    else stock_tokens_body<true>(p, b - hop_blocks, tok_blocks);
}
int stock_tok_fill(StockTokParams& p, const void* x, const void* din, const void* dout, int idx_dtype, int deg_dtype, int G, int N, int C,
                   int64_t n_atom, int64_t n_in, int64_t n_out, int64_t skip, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                   uint32_t salt) {
    if (G <= 0 || N < 0 || C <= 0 || (C & 3)) return MOBGT_EBADDIM;
    if (idx_dtype != MOBGT_I64 && idx_dtype != MOBGT_I32 && idx_dtype != MOBGT_I16) return MOBGT_EDTYPE;
    if (deg_dtype != MOBGT_I64 && deg_dtype != MOBGT_I32 && deg_dtype != MOBGT_I16) return MOBGT_EDTYPE;
    p = StockTokParams{};
    p.x = x; p.din = din; p.dout = dout; p.idx_dtype = idx_dtype; p.deg_dtype = deg_dtype; p.G = G; p.N = N; p.C = C;
    p.n_atom = n_atom; p.n_in = n_in; p.n_out = n_out; p.skip = skip;
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
    return 0;
}
}  // namespace

extern "C" int mobgt_stock_tokens_fwd(const void* x, const void* in_degree, const void* out_degree, int idx_dtype, int deg_dtype, const float* atom,
                                      const float* indeg, const float* outdeg, const float* graph_token, float* y, int G, int N,
                                      int C, int64_t n_atom, int64_t n_in, int64_t n_out, float dropout_p, uint64_t seed,
                                      const uint64_t* seed_dev, uint32_t salt, void* stream) {
    StockTokParams p;
    const int rc = stock_tok_fill(p, x, in_degree, out_degree, idx_dtype, deg_dtype, G, N, C, n_atom, n_in, n_out, -1, dropout_p, seed, seed_dev, salt);
    if (rc) return rc;
    if (((uintptr_t)atom | (uintptr_t)indeg | (uintptr_t)outdeg | (uintptr_t)graph_token | (uintptr_t)y) & 15) return MOBGT_EALIGN;
    p.atom = atom; p.indeg = indeg; p.outdeg = outdeg; p.gtok = graph_token; p.y = y;
    const int64_t total = (int64_t)G * (N + 1) * (C / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(stock_tokens_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int mobgt_stock_front_fwd(const void* x, const void* in_degree, const void* out_degree, int idx_dtype, int deg_dtype, const float* atom,
                                     const float* indeg, const float* outdeg, const float* graph_token, float* y, int G, int N,
                                     int C, int64_t n_atom, int64_t n_in, int64_t n_out, float dropout_p, uint64_t seed,
                                     const uint64_t* seed_dev, uint32_t salt, int n_pack, const void* const* pack_src,
                                     void* const* pack_dst, const int* pack_N, const int* pack_K, const int* pack_transposed,
                                     int has_hop, const float* edge_encoder, const float* edge_dis_encoder, float* hop_table, int D,
                                     int n_edge, int H, int fp16_roundtrip, void* stream) {
    StockTokParams p;
    int rc = stock_tok_fill(p, x, in_degree, out_degree, idx_dtype, deg_dtype, G, N, C, n_atom, n_in, n_out, -1, dropout_p, seed, seed_dev, salt);
    if (rc) return rc;
    if (((uintptr_t)atom | (uintptr_t)indeg | (uintptr_t)outdeg | (uintptr_t)graph_token | (uintptr_t)y) & 15) return MOBGT_EALIGN;
    p.atom = atom; p.indeg = indeg; p.outdeg = outdeg; p.gtok = graph_token; p.y = y;
    const int64_t total = (int64_t)G * (N + 1) * (C / 4);
    const int tok_blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    static mobgt_pack::PackJobs jobs;            // (by value into the launch; 3 KB -- not on the stack of every call)
    int nvb = 0;
    if (n_pack > 0) {
        jobs = mobgt_pack::PackJobs{};
        rc = mobgt_pack::fill_jobs(jobs, n_pack, pack_src, pack_dst, pack_N, pack_K, pack_transposed, &nvb);
        if (rc) return rc;
    }
    mobgt_front::HopFwd hf = {};
    int hop_blocks = 0;
    if (has_hop) {
        if (D <= 0 || n_edge <= 0 || H <= 0) return MOBGT_EBADDIM;
        hf = mobgt_front::HopFwd{edge_encoder, edge_dis_encoder, hop_table, D, n_edge, H, fp16_roundtrip};
        hop_blocks = (D * n_edge * H + 255) / 256;
    }
    static_assert(sizeof(StockTokParams) + sizeof(mobgt_pack::PackJobs) + sizeof(mobgt_front::HopFwd) + 32 <= 4096, "kernel arguments");
    hipLaunchKernelGGL(stock_front_kernel, dim3(tok_blocks + hop_blocks + nvb), dim3(256), 0, (hipStream_t)stream, p, tok_blocks, hf,
                       hop_blocks, jobs, n_pack > 0 ? n_pack : 0, nvb);
    return (int)hipGetLastError();
}

extern "C" int mobgt_stock_tokens_bwd(const float* dy, const void* x, const void* in_degree, const void* out_degree, int idx_dtype,
                                      int deg_dtype, float* d_atom, float* d_indeg, float* d_outdeg, float* d_graph_token, int G, int N, int C,
                                      int64_t n_atom, int64_t n_in, int64_t n_out, int64_t padding_idx, float dropout_p, uint64_t seed,
                                      const uint64_t* seed_dev, uint32_t salt, void* stream) {
    StockTokParams p;
    const int rc = stock_tok_fill(p, x, in_degree, out_degree, idx_dtype, deg_dtype, G, N, C, n_atom, n_in, n_out, padding_idx, dropout_p, seed,
                                  seed_dev, salt);
    if (rc) return rc;
    if ((uintptr_t)dy & 15) return MOBGT_EALIGN;
    p.dy = dy; p.d_atom = d_atom; p.d_indeg = d_indeg; p.d_outdeg = d_outdeg; p.d_gtok = d_graph_token;
    const int64_t total = (int64_t)G * (N + 1) * (C / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(stock_tokens_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// final_ln on the graph-token rows only (model.py:211-217 normalises every token and then reads row 0 of every graph:
// LayerNorm is per row, so only those rows are normalised -- same value, same gradient).  y[g,:] = LN(enc[g,0,:]); the
// backward also produces the whole d(enc) [G,T,C], zero outside the token rows.  torch: a strided copy + layer_norm forward,
// five launches backward (copy, grad-input, gamma/beta, fill, copy).  One workgroup per graph, C <= 1024.
namespace {
__device__ __forceinline__ float tl_block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void token_ln_fwd_kernel(const float* __restrict__ enc, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int T, int C, float eps) {
    __shared__ float sh[4];
    const int g = blockIdx.x;
    const float* x = enc + (int64_t)g * T * C;
    float v[4], s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + 256 * k;
        v[k] = c < C ? x[c] : 0.f;
        s += v[k];
    }
    const float mu = tl_block_sum(s, sh) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + 256 * k;
        const float d = c < C ? v[k] - mu : 0.f;
        q += d * d;
    }
    const float rs = rsqrtf(tl_block_sum(q, sh) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < C) y[(int64_t)g * C + c] = (v[k] - mu) * rs * w[c] + b[c];
    }
    if (threadIdx.x == 0) { mean[g] = mu; rstd[g] = rs; }
}

__global__ __launch_bounds__(256) void token_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ enc,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ w, float* __restrict__ denc,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int T, int C) {
    __shared__ float sh[4];
    const int g = blockIdx.x;
    float* dx = denc + (int64_t)g * T * C;
    // rows 1 .. T-1 of this graph receive no gradient from the head
    const int64_t n4 = ((int64_t)(T - 1) * C) / 4;
    if ((C & 3) == 0 && ((uintptr_t)dx & 15) == 0) {
        float4* z = reinterpret_cast<float4*>(dx + C);
        for (int64_t i = threadIdx.x; i < n4; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int64_t i = threadIdx.x; i < (int64_t)(T - 1) * C; i += 256) dx[C + i] = 0.f;
    }
    const float* x = enc + (int64_t)g * T * C;
    const float mu = mean[g], rs = rstd[g];
    float xh[4], gg[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + 256 * k;
        xh[k] = gg[k] = 0.f;
        if (c < C) {
            const float d = dy[(int64_t)g * C + c];
            xh[k] = (x[c] - mu) * rs;
            gg[k] = d * w[c];
            atomicAdd(&dgamma[c], d * xh[k]);
            atomicAdd(&dbeta[c], d);
            s1 += gg[k];
            s2 += gg[k] * xh[k];
        }
    }
    s1 = tl_block_sum(s1, sh) / (float)C;
    s2 = tl_block_sum(s2, sh) / (float)C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < C) dx[c] = rs * (gg[k] - s1 - xh[k] * s2);
    }
}
}  // namespace

extern "C" int mobgt_token_ln_fwd(const float* enc, const float* ln_w, const float* ln_b, float* y, float* mean, float* rstd, int G,
                                  int T, int C, float eps, void* stream) {
    if (G <= 0 || T <= 0 || C <= 0 || C > 1024) return MOBGT_EBADDIM;
    hipLaunchKernelGGL(token_ln_fwd_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, enc, ln_w, ln_b, y, mean, rstd, T, C, eps);
    return (int)hipGetLastError();
}

extern "C" int mobgt_token_ln_bwd(const float* dy, const float* enc, const float* mean, const float* rstd, const float* ln_w,
                                  float* denc, float* dgamma, float* dbeta, int G, int T, int C, void* stream) {
    if (G <= 0 || T <= 0 || C <= 0 || C > 1024) return MOBGT_EBADDIM;
    hipLaunchKernelGGL(token_ln_bwd_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, dy, enc, mean, rstd, ln_w, denc, dgamma, dbeta, T, C);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// cross_entropy(logits, target, ignore_index) with mean reduction over the rows whose target is not ignore_index -- the stock
// variant's training loss (model.py:218-285 with the POI datasets' NLLLoss(ignore_index = 0) of data.py:76 / :98 on
// log-softmax outputs) -- value and gradient in ONE launch: torch runs log_softmax, nll_loss and their two backward kernels
// (24 us at 16 x 7 857).  One workgroup per row: the row stays in registers between the max, the sum and the gradient
// (V <= 256 x CE_PER: one read of the logits), the mean's sum over rows is one fence-free 64-bit atomic per row (arrival
// count + fixed point, as in csrc/skinny.hip).  The loss is >= 0.
namespace {
constexpr int CE_PER = 40;                         // logits per thread kept in registers: V <= 10 240
__device__ unsigned long long ce_cell = 0ull;
__device__ unsigned int ce_bad = 0u;

__device__ __forceinline__ float ce_block_max(float v, float* sh) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ float ce_block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ z, const int64_t* __restrict__ target,
                                                           int64_t ignore_index, float* __restrict__ dz, float* __restrict__ loss,
                                                           int G, int V) {
    __shared__ float sh[4];
    const int g = blockIdx.x;
    // the number of rows that count (every workgroup finds it for itself: G is a handful)
    int cnt = 0;
    for (int i = threadIdx.x; i < G; i += 256) cnt += target[i] != ignore_index;
    const float n_live = ce_block_sum((float)cnt, sh);
    const int64_t t = target[g];
    const bool live = t != ignore_index && t >= 0 && t < V;
    const float* zr = z + (int64_t)g * V;
    float v[CE_PER];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < CE_PER; ++k) {
        const int c = threadIdx.x + 256 * k;
        v[k] = c < V ? zr[c] : -INFINITY;
        m = fmaxf(m, v[k]);
    }
    m = ce_block_max(m, sh);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CE_PER; ++k) {
        v[k] = __expf(v[k] - m);                      // (0 beyond V)
        s += v[k];
    }
    s = ce_block_sum(s, sh);
    const float inv_s = 1.f / s, scale = (live && n_live > 0.f) ? 1.f / n_live : 0.f;
    if (dz) {
        float* dr = dz + (int64_t)g * V;
#pragma unroll
        for (int k = 0; k < CE_PER; ++k) {
            const int c = threadIdx.x + 256 * k;
            if (c < V) dr[c] = (v[k] * inv_s - (c == t ? 1.f : 0.f)) * scale;
        }
    }
    if (threadIdx.x == 0) {
        // -log softmax(z)[t] = m + log s - z[t]
        const float nll = live ? (m + logf(s) - zr[t]) : 0.f;
        const bool ok = nll >= 0.f && nll < 65536.f;                      // (false for NaN; 4095 rows x 2^16 x 2^24 < 2^52)
        if (!ok) {
            __hip_atomic_fetch_or(&ce_bad, nll != nll ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
        }
        const unsigned long long add = 1ull | ((ok ? (unsigned long long)((double)nll * 16777216.0) : 0ull) << 12);
        const unsigned long long old = __hip_atomic_fetch_add(&ce_cell, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old & 0xFFFull) == (unsigned long long)G - 1ull) {
            const unsigned long long tot = (old + add) >> 12;
            const unsigned int bad = __hip_atomic_exchange(&ce_bad, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (no live row: torch returns nan for the mean over nothing)
            *loss = bad ? ((bad & 2u) ? __builtin_nanf("") : INFINITY)
                        : (n_live > 0.f ? (float)((double)tot * (1.0 / 16777216.0) / (double)n_live) : __builtin_nanf(""));
            __hip_atomic_store(&ce_cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
}  // namespace

extern "C" int mobgt_cross_entropy(const float* logits, const int64_t* targets, int64_t ignore_index, float* dlogits, float* loss,
                                   int G, int V, void* stream) {
    if (G <= 0 || G > 4095 || V <= 0 || V > 256 * CE_PER || !loss) return MOBGT_EBADDIM;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, logits, targets, ignore_index, dlogits, loss, G, V);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Stand-alone dropout (nn.Dropout at the model's input / output / positional / GCN sites), same counter hash
// as everywhere else: y = keep(seed, salt, row, col) ? x / (1-p) : 0.  The backward is the same call on dy.
namespace {
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int row_len,
                                                      uint32_t thr, float inv_keep, uint64_t seed0,
                                                      const uint64_t* __restrict__ seed_dev, uint32_t salt) {
    const uint64_t seed = seed0 + (seed_dev ? *seed_dev : 0ull);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / row_len;
        const uint32_t c = (uint32_t)(i - r * row_len);
        const uint32_t rowh = dropout_row_hash(seed, (uint32_t)r ^ salt);
        y[i] = dropout_bits16(seed, rowh, c) >= thr ? x[i] * inv_keep : 0.f;
    }
}
}  // namespace

extern "C" int mobgt_dropout(const float* x, float* y, int64_t n, int row_len, float dropout_p, uint64_t seed,
                             const uint64_t* seed_dev, uint32_t salt, void* stream) {
    if (n <= 0) return 0;
    if (row_len <= 0) return MOBGT_EBADDIM;
    const uint32_t thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    const float inv_keep = thr ? 1.f / (1.f - (float)thr / 65536.f) : 1.f;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, row_len, thr, inv_keep, seed,
                       seed_dev, salt);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// AdamW over the trainer's flat parameter buffer (torch.optim.AdamW defaults of model_fqandtoyo.py:1599-1616:
// decoupled weight decay, bias-corrected moments, eps outside the square root), one pass that also refreshes the
// bf16 shadow copy the layer GEMMs read.  lr and the step count are DEVICE scalars, so a captured graph sees new
// values on every replay: t = *step_dev - step_base (the trainer's per-step counter, advanced once per step).
namespace {
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, bf16_t* __restrict__ shadow, int64_t n,
                                                         const float* __restrict__ lr_dev, const float* __restrict__ sched,
                                                         const int64_t* __restrict__ step_dev,
                                                         int64_t step_base, float beta1, float beta2, float eps, float wd) {
    const float t = (float)(*step_dev - step_base);
    float lr;
    if (sched) {            // PolynomialDecayLR (lr.py:17-31, power = 1) evaluated at its step_count = t + offset
        const float w = sched[0], tot = sched[1], peak = sched[2], end = sched[3], c = t + sched[4];
        lr = c <= w ? c / w * peak : (c >= tot ? end : (peak - end) * (1.f - (c - w) / (tot - w)) + end);
    } else {
        lr = *lr_dev;
    }
    const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        float4 pp = *reinterpret_cast<float4*>(p + i);
        const float4 gg = *reinterpret_cast<const float4*>(g + i);
        float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
        float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            M[k] = beta1 * M[k] + (1.f - beta1) * G[k];
            V[k] = beta2 * V[k] + (1.f - beta2) * G[k] * G[k];
            P[k] = P[k] * decay - step_size * M[k] / (sqrtf(V[k]) * inv_sqrt_bc2 + eps);
        }
        *reinterpret_cast<float4*>(p + i) = pp;
        *reinterpret_cast<float4*>(m + i) = mm;
        *reinterpret_cast<float4*>(v + i) = vv;
        if (shadow) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (bf16_t)P[k];
            *reinterpret_cast<bf16x4*>(shadow + i) = o;
        }
    } else {
        for (int64_t j = i; j < n; ++j) {
            const float gj = g[j];
            const float mj = beta1 * m[j] + (1.f - beta1) * gj;
            const float vj = beta2 * v[j] + (1.f - beta2) * gj * gj;
            const float pj = p[j] * decay - step_size * mj / (sqrtf(vj) * inv_sqrt_bc2 + eps);
            m[j] = mj; v[j] = vj; p[j] = pj;
            if (shadow) shadow[j] = (bf16_t)pj;
        }
    }
}
}  // namespace

extern "C" int mobgt_adamw_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                                int64_t n, const float* lr_dev, const float* sched, const int64_t* step_dev,
                                int64_t step_base, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (n <= 0) return 0;
    if (!lr_dev && !sched) return MOBGT_EBADDIM;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return MOBGT_EALIGN;
    if (shadow_bf16 && ((uintptr_t)shadow_bf16 & 7)) return MOBGT_EALIGN;
    const int64_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, reinterpret_cast<bf16_t*>(shadow_bf16), n, lr_dev, sched, step_dev, step_base, beta1, beta2, eps,
                       weight_decay);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Token assembly at the encoder input (model_fqandtoyo.py:1287-1298, 348-358, 1338-1347), one launch each way:
//   out[g, 0,   :] = drop_in( drop_pos( graph_token + pe[0] ) )
//   out[g, 1+n, :] = drop_in( drop_pos( nf[g,n,:] * real[g,n] + add[g,n,:] ) )
// drop_pos = LearnablePositionalEncoding's dropout (:358), drop_in = input_dropout (:1347); the masks are the very
// ones the stand-alone dropout launches drew (same salts, same row numbering: g*N+n and g for drop_pos, g*T+t for
// drop_in), so this replaces -- bit for bit -- a multiply, an add, a repeat, an add, three dropouts and a cat.
namespace {
struct TokParams {
    const float *nf, *real, *add, *token, *pe0;
    float* out;
    bf16_t* out16;                           // optional bf16 copy of out (the first layer's GEMM operand)
    const float* dout;
    float *d_nf, *d_add, *d_token;           // d_token [C]: accumulated with atomics (zero it first)
    int G, N, C;
    uint32_t thr_pos, thr_in;
    float keep_pos, keep_in;                 // 1 / (1 - p)
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt_nf, salt_tok, salt_in;
};

template <bool BWD>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const TokParams p) {
    const int T = p.N + 1;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per output row (g, t)
    if (row >= (int64_t)p.G * T) return;
    const int lane = threadIdx.x & 63;
    const int g = (int)(row / T), t = (int)(row - (int64_t)g * T);
    const uint64_t seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
    const bool tok = t == 0;
    const uint32_t r1 = tok ? (uint32_t)g : (uint32_t)(g * p.N + (t - 1));
    const uint32_t h1 = p.thr_pos ? dropout_row_hash(seed, r1 ^ (tok ? p.salt_tok : p.salt_nf)) : 0u;
    const uint32_t h2 = p.thr_in ? dropout_row_hash(seed, (uint32_t)row ^ p.salt_in) : 0u;
    const int64_t src = ((int64_t)g * p.N + (t - 1)) * p.C;
    const float rl = tok ? 1.f : p.real[(int64_t)g * p.N + (t - 1)];
    for (int c = lane; c < p.C; c += 64) {
        float scale = 1.f;
        if (p.thr_pos) scale = dropout_bits16(seed, h1, (uint32_t)c) >= p.thr_pos ? p.keep_pos : 0.f;
        if (p.thr_in) scale *= dropout_bits16(seed, h2, (uint32_t)c) >= p.thr_in ? p.keep_in : 0.f;
        if (!BWD) {
            const float v = tok ? p.token[c] + p.pe0[c] : p.nf[src + c] * rl + p.add[src + c];
            p.out[row * p.C + c] = v * scale;
            if (p.out16) p.out16[row * p.C + c] = (bf16_t)(v * scale);
        } else {
            const float d = p.dout[row * p.C + c] * scale;
            if (tok) {
                if (d != 0.f) atomicAdd(&p.d_token[c], d);
            } else {
                p.d_nf[src + c] = d * rl;
                p.d_add[src + c] = d;
            }
        }
    }
}

void fill_tok(TokParams& p, float p_pos, float p_in, uint64_t seed, const uint64_t* seed_dev, uint32_t s_nf, uint32_t s_tok,
              uint32_t s_in) {
    p.thr_pos = p_pos > 0.f ? dropout_threshold(p_pos) : 0u;
    p.thr_in = p_in > 0.f ? dropout_threshold(p_in) : 0u;
    p.keep_pos = p.thr_pos ? 1.f / (1.f - (float)p.thr_pos / 65536.f) : 1.f;
    p.keep_in = p.thr_in ? 1.f / (1.f - (float)p.thr_in / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt_nf = s_nf; p.salt_tok = s_tok; p.salt_in = s_in;
}
}  // namespace

extern "C" int mobgt_assemble_tokens_fwd(const float* nf, const float* real, const float* add, const float* token,
                                         const float* pe0, float* out, void* out_bf16, int G, int N, int C, float p_pos,
                                         float p_in, uint64_t seed, const uint64_t* seed_dev, uint32_t salt_nf,
                                         uint32_t salt_tok, uint32_t salt_in, void* stream) {
    if (G <= 0 || N < 0 || C <= 0) return MOBGT_EBADDIM;
    TokParams p = {};
    p.nf = nf; p.real = real; p.add = add; p.token = token; p.pe0 = pe0; p.out = out; p.G = G; p.N = N; p.C = C;
    p.out16 = reinterpret_cast<bf16_t*>(out_bf16);
    fill_tok(p, p_pos, p_in, seed, seed_dev, salt_nf, salt_tok, salt_in);
    const int64_t rows = (int64_t)G * (N + 1);
    hipLaunchKernelGGL(assemble_tokens_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int mobgt_assemble_tokens_bwd(const float* dout, const float* real, float* d_nf, float* d_add, float* d_token,
                                         int G, int N, int C, float p_pos, float p_in, uint64_t seed,
                                         const uint64_t* seed_dev, uint32_t salt_nf, uint32_t salt_tok, uint32_t salt_in,
                                         void* stream) {
    if (G <= 0 || N < 0 || C <= 0) return MOBGT_EBADDIM;
    TokParams p = {};
    p.dout = dout; p.real = real; p.d_nf = d_nf; p.d_add = d_add; p.d_token = d_token; p.G = G; p.N = N; p.C = C;
    fill_tok(p, p_pos, p_in, seed, seed_dev, salt_nf, salt_tok, salt_in);
    const int64_t rows = (int64_t)G * (N + 1);
    hipLaunchKernelGGL(assemble_tokens_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Head activation chain on the G graph-token rows (model_fqandtoyo.py:1353-1364):
//   out = dropout( ELU( LayerNorm( LeakyReLU_0.2(u) ) ) )
// one wave per row (C <= 512), forward and backward one launch each instead of four + five.
namespace {
struct HeadParams {
    const float *u, *w, *b;
    float *out, *mean, *rstd;
    const float* dout;
    float *du, *dgamma, *dbeta;           // dgamma / dbeta accumulated with atomics (zero them first)
    int R, C;
    float eps, slope, inv_keep;
    uint32_t thr;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt;
};

template <bool BWD>
__global__ __launch_bounds__(256) void head_act_kernel(const HeadParams p) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.R) return;
    const int lane = threadIdx.x & 63;
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    const uint32_t rowh = p.thr ? dropout_row_hash(seed, (uint32_t)r ^ p.salt) : 0u;
    const float invC = 1.f / (float)p.C;
    float a[MAXC_PER_LANE], uu[MAXC_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k) {
        const int c = lane + 64 * k;
        uu[k] = c < p.C ? p.u[(int64_t)r * p.C + c] : 0.f;
        a[k] = uu[k] > 0.f ? uu[k] : p.slope * uu[k];
        s += c < p.C ? a[k] : 0.f;
    }
    float mu, rs;
    if (!BWD) {
        mu = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC_PER_LANE; ++k) {
            const float d = lane + 64 * k < p.C ? a[k] - mu : 0.f;
            q += d * d;
        }
        rs = rsqrtf(wave_sum(q) * invC + p.eps);
        if (lane == 0) { p.mean[r] = mu; p.rstd[r] = rs; }
    } else {
        mu = p.mean[r];
        rs = p.rstd[r];
    }
    float g[MAXC_PER_LANE], xh[MAXC_PER_LANE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k) {
        const int c = lane + 64 * k;
        g[k] = 0.f; xh[k] = 0.f;
        if (c < p.C) {
            xh[k] = (a[k] - mu) * rs;
            const float z = xh[k] * p.w[c] + p.b[c];
            const float keep = p.thr ? (dropout_bits16(seed, rowh, (uint32_t)c) >= p.thr ? p.inv_keep : 0.f) : 1.f;
            if (!BWD) {
                p.out[(int64_t)r * p.C + c] = (z > 0.f ? z : expm1f(z)) * keep;
            } else {
                const float dz = p.dout[(int64_t)r * p.C + c] * keep * (z > 0.f ? 1.f : expf(z));      // ELU'(z) = e^z for z <= 0
                atomicAdd(&p.dgamma[c], dz * xh[k]);
                atomicAdd(&p.dbeta[c], dz);
                g[k] = dz * p.w[c];
                s1 += g[k];
                s2 += g[k] * xh[k];
            }
        }
    }
    if (!BWD) return;
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int k = 0; k < MAXC_PER_LANE; ++k) {
        const int c = lane + 64 * k;
        if (c < p.C) {
            const float da = rs * (g[k] - s1 - xh[k] * s2);
            p.du[(int64_t)r * p.C + c] = da * (uu[k] > 0.f ? 1.f : p.slope);
        }
    }
}
}  // namespace

extern "C" int mobgt_head_act_fwd(const float* u, const float* ln_w, const float* ln_b, float* out, float* mean, float* rstd,
                                  int R, int C, float eps, float slope, float dropout_p, uint64_t seed,
                                  const uint64_t* seed_dev, uint32_t salt, void* stream) {
    if (R <= 0) return 0;
    if (C <= 0 || C > 64 * MAXC_PER_LANE) return MOBGT_EBADDIM;
    HeadParams p = {};
    p.u = u; p.w = ln_w; p.b = ln_b; p.out = out; p.mean = mean; p.rstd = rstd; p.R = R; p.C = C; p.eps = eps; p.slope = slope;
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
    hipLaunchKernelGGL(head_act_kernel<false>, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int mobgt_head_act_bwd(const float* dout, const float* u, const float* ln_w, const float* ln_b, const float* mean,
                                  const float* rstd, float* du, float* dgamma, float* dbeta, int R, int C, float eps,
                                  float slope, float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt,
                                  void* stream) {
    if (R <= 0) return 0;
    if (C <= 0 || C > 64 * MAXC_PER_LANE) return MOBGT_EBADDIM;
    HeadParams p = {};
    p.dout = dout; p.u = u; p.w = ln_w; p.b = ln_b; p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd);
    p.du = du; p.dgamma = dgamma; p.dbeta = dbeta; p.R = R; p.C = C; p.eps = eps; p.slope = slope;
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
    hipLaunchKernelGGL(head_act_kernel<true>, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// y = dropout( LeakyReLU_slope( x + bias ) ) on [R,C] f32 (C % 4 == 0) -- the epilogue of a GraphConvolution inside
// GCN.forward (modelGNN.py:38-44, 62-71) -- and its backward, which also produces the bias gradient:
// dx = dy * mask / keep * (y > 0 ? 1 : slope)  (y carries the sign of the pre-activation wherever the mask kept it),
// dbias[c] += colsum(dx).  Thread = 4 consecutive columns, the 4 waves of a workgroup take different rows.
namespace {
struct BiasActParams {
    const float *x, *bias, *dy, *y_in;
    float *y, *dx, *dbias;
    int64_t R;
    int C, rows_per_wg;
    int lpr;                                 // lanes per row: 64, or C/4 for narrow matrices (a wave then walks 64/lpr rows at once)
    float slope, inv_keep;
    uint32_t thr;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt;
    bf16_t* yt;                              // forward, optional: y transposed as bf16 [C][ldt] (the bitmask adjacency product's
    int64_t ldt;                             // operand layout, csrc/maskgemm.hip: no transpose launch in front of it)
};

template <bool BWD>
__global__ __launch_bounds__(256) void bias_act_kernel(const BiasActParams p) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 64 / p.lpr, sub = lane / p.lpr;          // the GCN's 16- and 64-wide layers would leave 94 / 75 % of
    const int c = 256 * blockIdx.y + 4 * (lane % p.lpr);     // a wave idle with one row per wave
    const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_wg;
    const int nrow = (int)min((int64_t)p.rows_per_wg, p.R - r0);
    const uint64_t seed = p.thr ? p.seed + (p.seed_dev ? *p.seed_dev : 0ull) : 0ull;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (!BWD && p.bias && c < p.C) ld4<float>(p.bias + c, b4);
    if (c < p.C) {
        for (int rr = wave * rpw + sub; rr < nrow; rr += 4 * rpw) {
            const int64_t r = r0 + rr;
            const uint32_t rowh = p.thr ? dropout_row_hash(seed, (uint32_t)r ^ p.salt) : 0u;
            float v[4], o[4];
            if (!BWD) {
                ld4<float>(p.x + r * p.C + c, v);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = v[i] + b4[i];
                    const float a = t > 0.f ? t : p.slope * t;
                    const float keep = p.thr ? (dropout_bits16(seed, rowh, (uint32_t)(c + i)) >= p.thr ? p.inv_keep : 0.f) : 1.f;
                    o[i] = a * keep;
                }
                st4<float>(p.y + r * p.C + c, o);
                if (p.yt) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) p.yt[(int64_t)(c + i) * p.ldt + r] = (bf16_t)o[i];
                }
            } else {
                float y4[4];
                ld4<float>(p.dy + r * p.C + c, v);
                ld4<float>(p.y_in + r * p.C + c, y4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float keep = p.thr ? (dropout_bits16(seed, rowh, (uint32_t)(c + i)) >= p.thr ? p.inv_keep : 0.f) : 1.f;
                    o[i] = v[i] * keep * (y4[i] > 0.f ? 1.f : p.slope);
                    acc[i] += o[i];
                }
                st4<float>(p.dx + r * p.C + c, o);
            }
        }
    }
    if (!BWD || !p.dbias) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][4 * lane + i] = acc[i];
    __syncthreads();
    const int cc = 256 * blockIdx.y + threadIdx.x;
    if (cc < p.C && threadIdx.x < 4 * p.lpr) {
        float t = 0.f;
        for (int sr = 0; sr < rpw; ++sr) {                  // the lanes (sub-rows) that own this column
            const int e = 4 * (sr * p.lpr) + threadIdx.x;
            t += red[0][e] + red[1][e] + red[2][e] + red[3][e];
        }
        atomicAdd(&p.dbias[cc], t);
    }
}

int launch_bias_act(BiasActParams& p, bool bwd, float dropout_p, hipStream_t st) {
    if (p.R <= 0) return 0;
    if (p.C <= 0 || (p.C & 3)) return MOBGT_EBADDIM;
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.lpr = 64;
    const int q = p.C / 4;
    if (q < 64 && (64 % q) == 0) p.lpr = q;
    p.rows_per_wg = bwd ? pick_rows(p.R) : pick_rows_stream(p.R);
    if (p.rows_per_wg < 4 * (64 / p.lpr)) p.rows_per_wg = 4 * (64 / p.lpr);       // one sweep of the four waves
    const dim3 grid((unsigned)((p.R + p.rows_per_wg - 1) / p.rows_per_wg), (unsigned)((p.C + 255) / 256)), block(256);
    if (bwd) hipLaunchKernelGGL(bias_act_kernel<true>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(bias_act_kernel<false>, grid, block, 0, st, p);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int mobgt_bias_act_fwd(const float* x, const float* bias, float* y, int64_t R, int C, float slope, float dropout_p,
                                  uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* stream) {
    BiasActParams p = {};
    p.x = x; p.bias = bias; p.y = y; p.R = R; p.C = C; p.slope = slope; p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
    return launch_bias_act(p, false, dropout_p, (hipStream_t)stream);
}

extern "C" int mobgt_bias_act_fwd_t(const float* x, const float* bias, float* y, void* y_t_bf16, int64_t ld_t, int64_t R, int C,
                                    float slope, float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt,
                                    void* stream) {
    if (y_t_bf16 && ld_t < R) return MOBGT_EBADDIM;
    BiasActParams p = {};
    p.x = x; p.bias = bias; p.y = y; p.R = R; p.C = C; p.slope = slope; p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
    p.yt = reinterpret_cast<bf16_t*>(y_t_bf16); p.ldt = ld_t;
    return launch_bias_act(p, false, dropout_p, (hipStream_t)stream);
}

extern "C" int mobgt_bias_act_bwd(const float* dy, const float* y, float* dx, float* dbias, int64_t R, int C, float slope,
                                  float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* stream) {
    BiasActParams p = {};
    p.dy = dy; p.y_in = y; p.dx = dx; p.dbias = dbias; p.R = R; p.C = C; p.slope = slope; p.seed = seed; p.seed_dev = seed_dev;
    p.salt = salt;
    return launch_bias_act(p, true, dropout_p, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// Start of a training step: zero the flat gradient buffer and the zero arena, advance the step counter (dropout
// seeds, AdamW's t) -- one launch instead of two fills and an add.
namespace {
__global__ __launch_bounds__(256) void step_prologue_kernel(float4* __restrict__ a, int64_t na, float4* __restrict__ b, int64_t nb,
                                                            int64_t hole0, int64_t hole1, int64_t* __restrict__ counter) {
    const int64_t n = na + nb;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (i < na) {
            if (i < hole0 || i >= hole1) a[i] = z;
        } else {
            b[i - na] = z;
        }
    }
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) *counter += 1;
}
}  // namespace

extern "C" int mobgt_step_prologue(float* zero_a, int64_t n_a, float* zero_b, int64_t n_b, int64_t* counter, void* stream) {
    return mobgt_step_prologue_skip(zero_a, n_a, 0, 0, zero_b, n_b, counter, stream);
}

extern "C" int mobgt_step_prologue_skip(float* zero_a, int64_t n_a, int64_t skip_begin, int64_t skip_end, float* zero_b, int64_t n_b,
                                        int64_t* counter, void* stream) {
    if (n_a < 0 || n_b < 0 || (n_a & 3) || (n_b & 3)) return MOBGT_EBADDIM;
    if (skip_begin < 0 || skip_end < skip_begin || skip_end > n_a) return MOBGT_EBADDIM;
    const int64_t hole0 = (skip_begin + 3) / 4, hole1 = skip_end / 4;           // whole float4s inside the hole only
    if (((uintptr_t)zero_a | (uintptr_t)zero_b) & 15) return MOBGT_EALIGN;
    const int64_t n4 = (n_a + n_b) / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(step_prologue_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(zero_a), n_a / 4, reinterpret_cast<float4*>(zero_b), n_b / 4, hole0,
                       hole1 > hole0 ? hole1 : hole0, counter);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Input of the classifier head (model_fqandtoyo.py:1239-1240, 1353-1358 for q = 0): row g = [ encoder output of the
// graph token | user_embedding[user[g] + user_offset] ], and its backward: d(enc) [G,T,C] is zero except the token
// rows, the user rows are added into the table gradient (pre-zeroed by the caller).  Replaces a cast, a subtraction,
// an index_select and a cat forward; two fills, two copies and the embedding backward.
namespace {
struct HeadInParams {
    const float* enc; float* x3; const float* table; const void* user; int user_dtype; int64_t user_offset;
    const float* dx3; float* denc; float* dtable;
    int G, T, C, U; int64_t n_rows;
};
__device__ __forceinline__ int64_t head_user(const HeadInParams& p, int g) {
    const int64_t u = p.user_dtype == MOBGT_I32 ? (int64_t)reinterpret_cast<const int32_t*>(p.user)[g]
                                                 : reinterpret_cast<const int64_t*>(p.user)[g];
    return u + p.user_offset;
}
__global__ __launch_bounds__(256) void head_input_fwd_kernel(const HeadInParams p) {
    const int W = p.C + p.U;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.G * W) return;
    const int g = i / W, c = i - g * W;
    float v;
    if (c < p.C) {
        v = p.enc[(int64_t)g * p.T * p.C + c];
    } else {
        const int64_t u = head_user(p, g);
        v = (u >= 0 && u < p.n_rows) ? p.table[u * p.U + (c - p.C)] : 0.f;
    }
    p.x3[i] = v;
}
__global__ __launch_bounds__(256) void head_input_bwd_kernel(const HeadInParams p) {
    const int W = p.C + p.U;
    const int64_t n1 = (int64_t)p.G * p.T * p.C;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n1) {
        const int64_t row = i / p.C;
        const int c = (int)(i - row * p.C);
        const int64_t g = row / p.T;
        p.denc[i] = (row - g * p.T == 0) ? p.dx3[g * W + c] : 0.f;
    } else if (i < n1 + (int64_t)p.G * p.U) {
        const int64_t j = i - n1;
        const int g = (int)(j / p.U), c = (int)(j - (int64_t)g * p.U);
        const int64_t u = head_user(p, g);
        if (u >= 0 && u < p.n_rows) atomicAdd(&p.dtable[u * p.U + c], p.dx3[(int64_t)g * W + p.C + c]);
    }
}
}  // namespace

extern "C" int mobgt_head_input_fwd(const float* enc, const void* user, int user_dtype, int64_t user_offset, const float* table,
                                    int64_t n_rows, float* x3, int G, int T, int C, int U, void* stream) {
    if (G <= 0 || T <= 0 || C <= 0 || U <= 0) return MOBGT_EBADDIM;
    if (user_dtype != MOBGT_I64 && user_dtype != MOBGT_I32) return MOBGT_EDTYPE;
    HeadInParams p = {enc, x3, table, user, user_dtype, user_offset, nullptr, nullptr, nullptr, G, T, C, U, n_rows};
    hipLaunchKernelGGL(head_input_fwd_kernel, dim3((G * (C + U) + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int mobgt_head_input_bwd(const float* dx3, const void* user, int user_dtype, int64_t user_offset, float* denc,
                                    float* dtable, int64_t n_rows, int G, int T, int C, int U, void* stream) {
    if (G <= 0 || T <= 0 || C <= 0 || U <= 0) return MOBGT_EBADDIM;
    if (user_dtype != MOBGT_I64 && user_dtype != MOBGT_I32) return MOBGT_EDTYPE;
    HeadInParams p = {nullptr, nullptr, nullptr, user, user_dtype, user_offset, dx3, denc, dtable, G, T, C, U, n_rows};
    const int64_t n = (int64_t)G * T * C + (int64_t)G * U;
    hipLaunchKernelGGL(head_input_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int mobgt_stock_tail_bwd(const float* dy, const void* x, const void* in_degree, const void* out_degree, int idx_dtype,
                                    int deg_dtype, float* d_atom, float* d_indeg, float* d_outdeg, float* d_graph_token, int G, int N, int C,
                                    int64_t n_atom, int64_t n_in, int64_t n_out, int64_t padding_idx, float dropout_p, uint64_t seed,
                                    const uint64_t* seed_dev, uint32_t salt, const float* d_hop_table, const float* edge_encoder,
                                    const float* edge_dis_encoder, float* d_edge_encoder, float* d_edge_dis_encoder, int D, int n_edge,
                                    int H, int fp16_roundtrip, void* stream) {
    StockTokParams p;
    int rc = stock_tok_fill(p, x, in_degree, out_degree, idx_dtype, deg_dtype, G, N, C, n_atom, n_in, n_out, padding_idx, dropout_p, seed,
                            seed_dev, salt);
    if (rc) return rc;
    if ((uintptr_t)dy & 15) return MOBGT_EALIGN;
    if (D <= 0 || n_edge <= 0 || H != 8 || n_edge > 2048) return MOBGT_EBADDIM;
    if (((uintptr_t)d_hop_table | (uintptr_t)edge_dis_encoder) & 15) return MOBGT_EALIGN;
    p.dy = dy; p.d_atom = d_atom; p.d_indeg = d_indeg; p.d_outdeg = d_outdeg; p.d_gtok = d_graph_token;
    const int64_t total = (int64_t)G * (N + 1) * (C / 4);
    const int tok_blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    const mobgt_hop::HopBwd hp = {d_hop_table, edge_encoder, edge_dis_encoder, d_edge_encoder, d_edge_dis_encoder, D, n_edge, fp16_roundtrip};
    const int hop_blocks = mobgt_hop::hop_bwd8_blocks(D, n_edge);
    const size_t lds = (size_t)2 * n_edge * 8 * sizeof(float);
    rc = (int)hipFuncSetAttribute((const void*)stock_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc) return rc;
    hipLaunchKernelGGL(stock_tail_kernel, dim3(hop_blocks + tok_blocks), dim3(256), lds, (hipStream_t)stream, p, tok_blocks, hp, hop_blocks);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Round 4: the sums over the library's split-K partial products, all of a step's in ONE launch.  Past 4 096 rows the
// encoder layers' weight gradients are batched GEMMs over row slices, [s, M, N] f32 partial products each followed by a
// `.sum(0)` launch -- 36 launches of ~5.5 us per S-BIG step for ~4 MB each.  Nothing but the optimizer reads a weight
// gradient, so the sums wait for the end of the backward pass (ops.flush_deferred_wgrads) and run as one grid that writes
// the gradients' sinks: dst[e] = sum_k src[k * numel + e].
namespace {
constexpr int PSUM_MAX = 48;
struct PsumJobs {
    const float* src[PSUM_MAX];
    float* dst[PSUM_MAX];
    int64_t numel[PSUM_MAX];                       // per slice, % 4 == 0
    int s[PSUM_MAX];
    int first_block[PSUM_MAX + 1];                 // 256 threads x 4 elements per block
};
__global__ __launch_bounds__(256) void partial_sum_multi_kernel(const PsumJobs jobs, int njobs) {
    int job = 0;
    const int b = (int)blockIdx.x;
    while (job + 1 < njobs && b >= jobs.first_block[job + 1]) ++job;
    const int64_t e = ((int64_t)(b - jobs.first_block[job]) * 256 + threadIdx.x) * 4;
    const int64_t n = jobs.numel[job];
    if (e >= n) return;
    const float* src = jobs.src[job] + e;
    const int S = jobs.s[job];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 4 <= S; k += 4) {                   // four slices' loads in flight
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(k + u) * n);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; k < S; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)k * n);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(jobs.dst[job] + e) = acc;
}
}  // namespace

extern "C" int mobgt_partial_sum_multi(int n, const float* const* src, float* const* dst, const int* s, const int64_t* numel,
                                       void* stream) {
    if (n <= 0) return 0;
    if (n > PSUM_MAX) return MOBGT_EBADDIM;
    static PsumJobs jobs;                          // (by value into the launch; 1.3 KB)
    jobs = PsumJobs{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (s[i] <= 0 || numel[i] <= 0 || (numel[i] & 3)) return MOBGT_EBADDIM;
        if (((uintptr_t)src[i] | (uintptr_t)dst[i]) & 15) return MOBGT_EALIGN;
        jobs.src[i] = src[i]; jobs.dst[i] = dst[i]; jobs.s[i] = s[i]; jobs.numel[i] = numel[i];
        jobs.first_block[i] = blocks;
        blocks += (int)((numel[i] / 4 + 255) / 256);
    }
    jobs.first_block[n] = blocks;
    hipLaunchKernelGGL(partial_sum_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, jobs, n);
    return (int)hipGetLastError();
}
