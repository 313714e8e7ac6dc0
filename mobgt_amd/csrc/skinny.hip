// y = x W^T + b for a handful of rows (G <= 16) and a wide output: the classifier head of MobGT,
// out_proj = nn.Linear(448, P+1) on the G graph tokens (model_fqandtoyo.py:1394; f32).  A GEMM with M = 16 is a
// bandwidth problem -- every weight is used 16 times -- that the library runs at 0.5 TB/s (28 us forward, 29 + 26 us
// for the two backward products at V = 7857); these kernels stream W once per product.
//   forward : one wave per output column v; lane l keeps its slice of all G rows of x in registers, reads its slice
//             of W[v,:] as 16-byte pieces, and the G partial dot products are combined by a butterfly reduce-scatter.
//   dx      : a workgroup owns 8 columns k of dx and streams the 32-byte column block W[:, k0:k0+8] over all v
//             (dy staged through LDS in chunks); 32 v-lanes per column are combined in LDS.  No atomics.
//   dW, db  : a workgroup owns 16 rows v: dW[v,k] = sum_g dy[g,v] x[g,k] with x in registers, coalesced stores.
#include "common.h"
#include "mobgt_hip.h"

namespace {

constexpr int GMAX = 16;
constexpr int KCH = 2;                       // K <= 512: lane l owns k = 4l + 256c .. +3, c < KCH

__device__ __forceinline__ void reduce16(float (&v)[GMAX], int lane) {
    // afterwards lane l holds in v[0] the wave-wide sum of value (l >> 2) & 15
    int w = GMAX / 2;
#pragma unroll
    for (int bit = 32; w >= 1; bit >>= 1, w >>= 1) {
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int t = 0; t < w; ++t) {
            const float send = up ? v[t] : v[t + w];
            const float keep = up ? v[t + w] : v[t];
            v[t] = keep + __shfl_xor(send, bit, 64);
        }
    }
    v[0] += __shfl_xor(v[0], 2, 64);
    v[0] += __shfl_xor(v[0], 1, 64);
}

__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int G, int K, int V) {
    const int lane = threadIdx.x & 63;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    float xr[GMAX][KCH][4];
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int k = 4 * lane + 256 * c;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g < G && k < K) t = *reinterpret_cast<const float4*>(x + (int64_t)g * K + k);
            xr[g][c][0] = t.x; xr[g][c][1] = t.y; xr[g][c][2] = t.z; xr[g][c][3] = t.w;
        }
    // the row of W for the NEXT column is requested before the current one is consumed (a wave handles ~8 columns
    // one after the other: without the prefetch every column paid a full memory round trip)
    float4 wv[KCH], wn[KCH];
    auto load_row = [&](int v, float4 (&dst)[KCH]) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int k = 4 * lane + 256 * c;
            dst[c] = (v < V && k < K) ? *reinterpret_cast<const float4*>(w + (int64_t)v * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_row(wave_id, wv);
    for (int v = wave_id; v < V; v += n_waves) {
        load_row(v + n_waves, wn);
        float acc[GMAX];
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KCH; ++c)
                s += xr[g][c][0] * wv[c].x + xr[g][c][1] * wv[c].y + xr[g][c][2] * wv[c].z + xr[g][c][3] * wv[c].w;
            acc[g] = s;
        }
        reduce16(acc, lane);
        const int g = (lane >> 2) & 15;
        if ((lane & 3) == 0 && g < G) y[(int64_t)g * V + v] = acc[0] + (b ? b[v] : 0.f);
#pragma unroll
        for (int c = 0; c < KCH; ++c) wv[c] = wn[c];
    }
}

constexpr int DX_KB = 8;                     // columns of dx per workgroup
constexpr int DX_VL = 32;                    // v-lanes per column (256 threads)
constexpr int DX_CHUNK = 512;                // rows of dy staged per round

constexpr int DX_SPLIT = 8;                  // workgroups along V per column block (f32 atomics into the zeroed dx)

__global__ __launch_bounds__(256) void skinny_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                        float* __restrict__ dx, int G, int K, int V) {
    __shared__ float sdy[GMAX][DX_CHUNK];
    __shared__ float red[DX_VL][DX_KB][GMAX + 1];
    const int kk = threadIdx.x & (DX_KB - 1), vl = threadIdx.x / DX_KB;
    const int k = blockIdx.x * DX_KB + kk;
    float acc[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) acc[g] = 0.f;
    const int vper = ((V + DX_SPLIT - 1) / DX_SPLIT + DX_CHUNK - 1) / DX_CHUNK * DX_CHUNK;
    const int vbeg = blockIdx.y * vper, vstop = min(V, vbeg + vper);
    for (int v0 = vbeg; v0 < vstop; v0 += DX_CHUNK) {
        __syncthreads();
        for (int e = threadIdx.x; e < GMAX * DX_CHUNK; e += 256) {
            const int g = e / DX_CHUNK, j = e - g * DX_CHUNK;
            sdy[g][j] = (g < G && v0 + j < vstop) ? dy[(int64_t)g * V + v0 + j] : 0.f;
        }
        __syncthreads();
        float wv[DX_CHUNK / DX_VL];                                    // this thread's 16 rows of the chunk, all in flight
#pragma unroll
        for (int i = 0; i < DX_CHUNK / DX_VL; ++i) {
            const int v = v0 + vl + DX_VL * i;
            wv[i] = (k < K && v < vstop) ? w[(int64_t)v * K + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < DX_CHUNK / DX_VL; ++i) {
#pragma unroll
            for (int g = 0; g < GMAX; ++g) acc[g] += sdy[g][vl + DX_VL * i] * wv[i];
        }
    }
#pragma unroll
    for (int g = 0; g < GMAX; ++g) red[vl][kk][g] = acc[g];
    __syncthreads();
    if (threadIdx.x < DX_KB * GMAX) {
        const int g = threadIdx.x / DX_KB, k2 = threadIdx.x % DX_KB;
        float s = 0.f;
#pragma unroll 8
        for (int l = 0; l < DX_VL; ++l) s += red[l][k2][g];
        const int kc = blockIdx.x * DX_KB + k2;
        if (g < G && kc < K) atomicAdd(&dx[(int64_t)g * K + kc], s);
    }
}

// dx on the matrix cores.  dx[g][k] = sum_v dy[g][v] W[v][k]: G <= 16 rows is exactly one v_mfma_f32_16x16x4_f32 tile, and
// the f32 MFMA rate (256 FLOP/clk/CU) wants >= ~100 CUs on it.  A workgroup owns ONE 16-column tile of dx and an eighth of
// V: 8 x K/16 workgroups (224 at K = 448), each streaming its [V/8 x 16] column block of W by 16-byte loads (a 64-byte run
// per weight row: the load unit runs at full rate, unlike the 4-byte column walks of skinny_dx_kernel) through LDS; its four
// waves split the rows, meet in LDS, and 256 f32 atomics per workgroup fold the eight V-slices (dx zeroed by the caller).
// A row-owning layout (64 rows x all K per workgroup) was tried first: 123 x 7168 atomics made it 18 us.
constexpr int DXM_ROWS = 256;                // rows of W per staged chunk
constexpr int DXM_VSPLIT = 8;
// (round 4) past 32 k classes: 32 slices.  Eight slices are 224 workgroups with one 16 KB chunk each in flight -- 3.6 MB, a
// quarter of what 8 TB/s x ~2 us of latency needs -- and at V = 100 001 (S-BIG) each walks 49 chunks: 115 us for 358 MB.
inline int dxm_vsplit(int V) { return V >= 32768 ? 32 : DXM_VSPLIT; }
typedef __attribute__((ext_vector_type(4))) float f32x4_;

__device__ __forceinline__ void skinny_dx_mfma_body(const float* __restrict__ dy, const float* __restrict__ w,
                                                    float* __restrict__ dx, int G, int K, int V, int bx, int by, int vsplit) {
    __shared__ __attribute__((aligned(16))) float wl[DXM_ROWS][16];            // W[v0 + r][k0 .. k0 + 15]
    __shared__ __attribute__((aligned(16))) float dyl[GMAX][DXM_ROWS + 4];     // dy[g][v0 + r]
    __shared__ float part[4][16][17];
    const int k0 = bx * 16;
    const int vper = ((V + vsplit - 1) / vsplit + 3) & ~3;
    const int vbeg = by * vper, vend = min(V, vbeg + vper);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
    float4 wr[DXM_ROWS * 4 / 256];
    float dr[GMAX * DXM_ROWS / 256];
    auto fetch = [&](int v0) {                                                 // the next chunk rides in registers during the MFMAs
        const int nv = min(DXM_ROWS, vend - v0);
#pragma unroll
        for (int u = 0; u < DXM_ROWS * 4 / 256; ++u) {                         // 4 lanes x 16 bytes per weight row
            const int e = threadIdx.x + 256 * u, r = e >> 2, c = (e & 3) * 4;
            wr[u] = r < nv ? *reinterpret_cast<const float4*>(w + (int64_t)(v0 + r) * K + k0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < GMAX * DXM_ROWS / 256; ++u) {
            const int e = threadIdx.x + 256 * u, g = e / DXM_ROWS, r = e % DXM_ROWS;
            dr[u] = (g < G && r < nv) ? dy[(int64_t)g * V + v0 + r] : 0.f;
        }
    };
    if (vbeg < vend) fetch(vbeg);
    for (int v0 = vbeg; v0 < vend; v0 += DXM_ROWS) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < DXM_ROWS * 4 / 256; ++u) {
            const int e = threadIdx.x + 256 * u;
            *reinterpret_cast<float4*>(&wl[e >> 2][(e & 3) * 4]) = wr[u];
        }
#pragma unroll
        for (int u = 0; u < GMAX * DXM_ROWS / 256; ++u) {
            const int e = threadIdx.x + 256 * u;
            dyl[e / DXM_ROWS][e % DXM_ROWS] = dr[u];
        }
        __syncthreads();
        if (v0 + DXM_ROWS < vend) fetch(v0 + DXM_ROWS);
        const int rb = wave * (DXM_ROWS / 4);                                   // this wave's 64 rows of the chunk
#pragma unroll
        for (int s_ = 0; s_ < DXM_ROWS / 16; ++s_)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dyl[j][rb + 4 * s_ + q], wl[rb + 4 * s_ + q][j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) part[wave][4 * q + v][j] = acc[v];
    __syncthreads();
    {
        const int g = threadIdx.x >> 4, c = threadIdx.x & 15;
        if (g < G) atomicAdd(&dx[(int64_t)g * K + k0 + c], part[0][g][c] + part[1][g][c] + part[2][g][c] + part[3][g][c]);
    }
}

__global__ __launch_bounds__(256) void skinny_dx_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             float* __restrict__ dx, int G, int K, int V) {
    skinny_dx_mfma_body(dy, w, dx, G, K, V, blockIdx.x, blockIdx.y, (int)gridDim.y);
}

constexpr int DW_ROWS = 16;                  // rows v of dW per workgroup

__device__ __forceinline__ void skinny_dw_body(const float* __restrict__ dy, const float* __restrict__ x,
                                               float* __restrict__ dw, float* __restrict__ db, int G, int K, int V, int bx) {
    __shared__ float sdy[DW_ROWS][GMAX];
    const int v0 = bx * DW_ROWS;
    {
        const int j = threadIdx.x / GMAX, g = threadIdx.x % GMAX;         // 256 = 16 x 16
        sdy[j][g] = (g < G && v0 + j < V) ? dy[(int64_t)g * V + v0 + j] : 0.f;
    }
    float xr[GMAX][KCH];
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int k = threadIdx.x + 256 * c;
            xr[g][c] = (g < G && k < K) ? x[(int64_t)g * K + k] : 0.f;
        }
    __syncthreads();
    for (int j = 0; j < DW_ROWS && v0 + j < V; ++j) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const int k = threadIdx.x + 256 * c;
            if (k < K) {
                float s = 0.f;
#pragma unroll
                for (int g = 0; g < GMAX; ++g) s += sdy[j][g] * xr[g][c];
                dw[(int64_t)(v0 + j) * K + k] = s;
            }
        }
    }
    if (db && threadIdx.x < DW_ROWS && v0 + (int)threadIdx.x < V) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < GMAX; ++g) s += sdy[threadIdx.x][g];
        db[v0 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void skinny_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        float* __restrict__ dw, float* __restrict__ db, int G, int K, int V) {
    skinny_dw_body(dy, x, dw, db, G, K, V, blockIdx.x);
}

// Both gradients of the classifier in ONE launch: they share nothing but dy, each is a ~8 us launch of a few hundred
// workgroups, and neither depends on the other -- the first K/16 x 8 workgroups run the dx body, the rest the dW body.
__global__ __launch_bounds__(256) void skinny_bwd_both_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ w, float* __restrict__ dx,
                                                              float* __restrict__ dw, float* __restrict__ db, int G, int K, int V,
                                                              int vsplit) {
    const int ndx = (K / 16) * vsplit;
    if ((int)blockIdx.x < ndx) skinny_dx_mfma_body(dy, w, dx, G, K, V, blockIdx.x % (K / 16), blockIdx.x / (K / 16), vsplit);
    else skinny_dw_body(dy, x, dw, db, G, K, V, blockIdx.x - ndx);
}

__global__ __launch_bounds__(256) void skinny_zero_kernel(float* __restrict__ p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// The classifier and its loss in ONE launch (training): logits = x W^T + b on the matrix cores, and GradientTailLoss
// (model_fqandtoyo.py:545-550, elementwise on sigmoid(logit): csrc/layer.hip gtl_kernel's formulas) applied in the epilogue --
// d loss / d logits and the loss's partial sums leave the kernel, the logits only when asked for.  The library GEMM (9.2 us at
// G = 16, K = 320, V = 7856) + the loss kernel (9.1 us) were two launches that are mostly ramp.
// A wave owns 16 output columns: C[g][v] = sum_k x[g][k] W[v][k] is v_mfma_f32_16x16x4_f32 with A = x (rows g), B = W^T
// (columns v); lane (j = lane & 15, q = lane >> 4) streams W[v0 + j][16 s + 4 q .. + 3], s < K / 16 -- per load instruction the
// four q of a weight row read 64 contiguous bytes (a first version gave each lane 64 contiguous bytes, i.e. 64 scattered
// 16-byte pieces per instruction: 17 us) -- all K / 16 pieces in flight at once; x waits in LDS and is read in the same k order.
constexpr int FG_WAVES = 2;                  // 32 columns per workgroup: 246 workgroups at V = 7856
constexpr int FG_MAXB = 1024;               // workgroups per launch (a wave walks over its 16-column tiles)
// The loss's cross-workgroup sum WITHOUT a fence (an agent-scope release writes this XCD's whole L2 back: ~10 us, which is most
// of what gtl_kernel's 9.1 us were): every workgroup makes ONE 64-bit atomic add that carries its arrival (bits 0-11) and its
// partial sum in fixed point (bits 12-63, 2^-24 units: integer addition, so the total does not depend on the arrival order);
// the workgroup whose add returns the last count holds the complete sum in that very return value.  246 adds on ONE address cost
// 3.5 us (measured), so there are two levels: eight cells 256 bytes apart (workgroup b -> cell b % 8) whose last arrivers add
// their cell's total into a ninth; its last arriver writes the loss, and every last arriver re-arms its cell.  Partial sums are
// >= 0; one that is not finite or >= 2^18 (a diverged run) raises fg_bad and the loss reads +inf -- so the field cannot
// overflow (1024 x 2^18 x 2^24 = 2^52).
__device__ unsigned long long fg_cell[9 * 32];
__device__ unsigned int fg_bad = 0u;

template <int KT>
__global__ __launch_bounds__(64 * FG_WAVES) void skinny_fwd_gtl_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                      const float* __restrict__ b, const int64_t* __restrict__ target,
                                                                      int64_t target_offset, float* __restrict__ y,
                                                                      float* __restrict__ dz, float* __restrict__ loss, int G, int V,
                                                                      float alpha) {
    constexpr int K = 64 * KT, LDX = K + 4;
    __shared__ __attribute__((aligned(16))) float xs[GMAX][LDX];
    __shared__ float part[FG_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    float lsum = 0.f;
    const float inv_n = 1.f / ((float)G * (float)V);
    int64_t tgt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tgt[i] = (target && 4 * q + i < G) ? target[4 * q + i] + target_offset : -1;
    bool first = true;
    for (int v0 = ((int)blockIdx.x * FG_WAVES + wave) * 16; v0 < V || first; v0 += (int)gridDim.x * FG_WAVES * 16) {
    const int v = v0 + j;
    // this lane's weights: all in flight before anything else
    float4 wr[4 * KT];
    {
        const float* wrow = w + (int64_t)(v < V ? v : 0) * K + 4 * q;
#pragma unroll
        for (int s_ = 0; s_ < 4 * KT; ++s_)
            wr[s_] = v < V ? *reinterpret_cast<const float4*>(wrow + 16 * s_) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float bv = (b && v < V) ? b[v] : 0.f;
    if (first) {                                      // (every wave passes here exactly once: its first tile may lie beyond V)
        for (int e = threadIdx.x; e < GMAX * (K / 4); e += 64 * FG_WAVES) {
            const int g = e / (K / 4), c = e % (K / 4);
            *reinterpret_cast<float4*>(&xs[g][4 * c]) = g < G ? *reinterpret_cast<const float4*>(x + (int64_t)g * K + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        first = false;
    }
    f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < 4 * KT; ++s_) {
        const float4 a = *reinterpret_cast<const float4*>(&xs[j][16 * s_ + 4 * q]);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wr[s_].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wr[s_].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wr[s_].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wr[s_].w, acc, 0, 0, 0);
    }
    // lane (j, q) holds C[4 q + i][v0 + j], i < 4
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = 4 * q + i;
        if (g >= G || v >= V) continue;
        const float z = acc[i] + bv;
        if (y) y[(int64_t)g * V + v] = z;
        if (!target) continue;                        // (the plain forward: mobgt_skinny_linear_fwd_mfma)
        const float p = 1.f / (1.f + __expf(-z));
        const float pq = 1.f - p;
        float l, d;
        if (tgt[i] == v) {
            const float lp = logf(p);
            l = -alpha * pq * lp;
            d = alpha * p * pq * lp - alpha * pq * pq;
        } else {
            const float lq = logf(pq);
            l = -p * lq;
            d = -p * pq * lq + p * p;
        }
        lsum += l;
        dz[(int64_t)g * V + v] = d * inv_n;
    }
    }
    if (!target) return;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
    if (lane == 0) part[wave] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < FG_WAVES; ++k) s += part[k];
        const bool ok = s >= 0.f && s < 262144.f;                         // (false for NaN)
        if (!ok) {
            __hip_atomic_fetch_or(&fg_bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();                                              // (the rare path may pay for the ordering)
        }
        const unsigned long long add = 1ull | ((ok ? (unsigned long long)(s * 16777216.f) : 0ull) << 12);
        const unsigned int nb = gridDim.x, c = blockIdx.x & 7u;
        const unsigned int n_c = (nb - c + 7u) >> 3;                      // workgroups that report to cell c
        const unsigned long long old = __hip_atomic_fetch_add(&fg_cell[32 * c], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old & 0xFFFull) == (unsigned long long)n_c - 1ull) {
            __hip_atomic_store(&fg_cell[32 * c], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long add2 = 1ull | (((old + add) >> 12) << 12);
            const unsigned long long old2 = __hip_atomic_fetch_add(&fg_cell[32 * 8], add2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((old2 & 0xFFFull) == (unsigned long long)(nb < 8u ? nb : 8u) - 1ull) {
                const unsigned long long tot = (old2 + add2) >> 12;
                const unsigned int bad = __hip_atomic_exchange(&fg_bad, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *loss = bad ? INFINITY : (float)((double)tot * (1.0 / 16777216.0) * (double)inv_n);
                __hip_atomic_store(&fg_cell[32 * 8], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int KT>
int launch_fwd_gtl(const float* x, const float* w, const float* b, const int64_t* target, int64_t target_offset, float* y, float* dz,
                   float* loss, int G, int V, float alpha, hipStream_t st) {
    int blocks = (V + 16 * FG_WAVES - 1) / (16 * FG_WAVES);
    if (blocks > FG_MAXB) blocks = FG_MAXB;
    hipLaunchKernelGGL((skinny_fwd_gtl_kernel<KT>), dim3(blocks), dim3(64 * FG_WAVES), 0, st, x, w, b, target, target_offset, y, dz, loss,
                       G, V, alpha);
    return (int)hipGetLastError();
}

int check_dims(int G, int K, int V) {
    if (G <= 0 || G > GMAX || K <= 0 || K > 256 * KCH || (K & 3) || V <= 0) return MOBGT_EBADDIM;
    return 0;
}

}  // namespace

extern "C" int mobgt_skinny_linear_fwd(const float* x, const float* w, const float* b, float* y, int G, int K, int V,
                                       void* stream) {
    const int rc = check_dims(G, K, V);
    if (rc) return rc;
    if (((uintptr_t)x | (uintptr_t)w) & 15) return MOBGT_EALIGN;
    const int blocks = (V + 31) / 32 < 256 ? (V + 31) / 32 : 256;        // ~8 columns per wave: x is loaded once per wave
    hipLaunchKernelGGL(skinny_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, b, y, G, K, V);
    return (int)hipGetLastError();
}

extern "C" int mobgt_skinny_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                                       int G, int K, int V, void* stream) {
    const int rc = check_dims(G, K, V);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        // dx is accumulated across the DX_SPLIT workgroups of a column block: zero it first (a kernel, not a memset node)
        hipLaunchKernelGGL(skinny_zero_kernel, dim3((G * K + 255) / 256), dim3(256), 0, st, dx, G * K);
        hipLaunchKernelGGL(skinny_dx_kernel, dim3((K + DX_KB - 1) / DX_KB, DX_SPLIT), dim3(256), 0, st, dy, w, dx, G, K, V);
    }
    if (dw) hipLaunchKernelGGL(skinny_dw_kernel, dim3((V + DW_ROWS - 1) / DW_ROWS), dim3(256), 0, st, dy, x, dw, db, G, K, V);
    return (int)hipGetLastError();
}

/* dx = dy @ w only (dx must be ZERO on entry: f32 atomics), on the matrix cores; K % 16 == 0. */
extern "C" int mobgt_skinny_linear_dx(const float* dy, const float* w, float* dx, int G, int K, int V, void* stream) {
    const int rc = check_dims(G, K, V);
    if (rc) return rc;
    if ((K & 15) || ((uintptr_t)w & 15)) return MOBGT_EBADDIM;
    hipLaunchKernelGGL(skinny_dx_mfma_kernel, dim3(K / 16, dxm_vsplit(V)), dim3(256), 0, (hipStream_t)stream, dy, w, dx, G, K, V);
    return (int)hipGetLastError();
}

/* dx = dy @ w (dx ZERO on entry, K % 16 == 0) and dw = dy^T x (+ db = column sums of dy) in one launch. */
extern "C" int mobgt_skinny_linear_bwd_both(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                                            int G, int K, int V, void* stream) {
    const int rc = check_dims(G, K, V);
    if (rc) return rc;
    if ((K & 15) || ((uintptr_t)w & 15) || !dx || !dw) return MOBGT_EBADDIM;
    const int vsplit = dxm_vsplit(V);
    const int blocks = (K / 16) * vsplit + (V + DW_ROWS - 1) / DW_ROWS;
    hipLaunchKernelGGL(skinny_bwd_both_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, w, dx, dw, db, G, K, V, vsplit);
    return (int)hipGetLastError();
}

/* logits = x w^T + b, loss = GradientTailLoss(logits, target + target_offset, alpha) and dlogits = d loss / d logits in one
 * launch (K % 64 == 0, K <= 448; logits may be NULL).  Not re-entrant across streams (one ticket). */
namespace {
int fwd_gtl_dispatch(const float* x, const float* w, const float* b, const int64_t* targets, int64_t target_offset, float* logits,
                     float* dlogits, float* loss, int G, int K, int V, float alpha, void* stream);
}

/* y = x w^T + b alone, by the same kernel (K % 64 == 0, K <= 448): one pass over w on the matrix cores. */
extern "C" int mobgt_skinny_linear_fwd_mfma(const float* x, const float* w, const float* b, float* y, int G, int K, int V, void* stream) {
    const int rc = check_dims(G, K, V);
    if (rc) return rc;
    if ((K & 63) || K > 448 || !y) return MOBGT_EBADDIM;
    if (((uintptr_t)x | (uintptr_t)w) & 15) return MOBGT_EALIGN;
    return fwd_gtl_dispatch(x, w, b, nullptr, 0, y, nullptr, nullptr, G, K, V, 0.f, stream);
}

extern "C" int mobgt_skinny_linear_gtl(const float* x, const float* w, const float* b, const int64_t* targets, int64_t target_offset,
                                       float* logits, float* dlogits, float* loss, int G, int K, int V, float alpha, void* stream) {
    const int rc = check_dims(G, K, V);
    if (rc) return rc;
    if ((K & 63) || K > 448 || !dlogits || !loss || !targets) return MOBGT_EBADDIM;
    if (((uintptr_t)x | (uintptr_t)w) & 15) return MOBGT_EALIGN;
    return fwd_gtl_dispatch(x, w, b, targets, target_offset, logits, dlogits, loss, G, K, V, alpha, stream);
}

namespace {
int fwd_gtl_dispatch(const float* x, const float* w, const float* b, const int64_t* targets, int64_t target_offset, float* logits,
                     float* dlogits, float* loss, int G, int K, int V, float alpha, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (K / 64) {
        case 1: return launch_fwd_gtl<1>(x, w, b, targets, target_offset, logits, dlogits, loss, G, V, alpha, st);
        case 2: return launch_fwd_gtl<2>(x, w, b, targets, target_offset, logits, dlogits, loss, G, V, alpha, st);
        case 3: return launch_fwd_gtl<3>(x, w, b, targets, target_offset, logits, dlogits, loss, G, V, alpha, st);
        case 4: return launch_fwd_gtl<4>(x, w, b, targets, target_offset, logits, dlogits, loss, G, V, alpha, st);
        case 5: return launch_fwd_gtl<5>(x, w, b, targets, target_offset, logits, dlogits, loss, G, V, alpha, st);
        case 6: return launch_fwd_gtl<6>(x, w, b, targets, target_offset, logits, dlogits, loss, G, V, alpha, st);
        default: return launch_fwd_gtl<7>(x, w, b, targets, target_offset, logits, dlogits, loss, G, V, alpha, st);
    }
}
}  // namespace
