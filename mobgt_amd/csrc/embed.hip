// Embedding gathers for the node features (graphormer/model.py:193-203, model_fqandtoyo.py:1259-1264,
// 1288-1298): out[r,:] = sum_t table_t[idx_t[r],:], one wave per output row, 16-byte lanes, and the
// matching scatter-add backward (f32 atomics; rows equal to the table's padding index are skipped,
// which is nn.Embedding(padding_idx=0)'s gradient rule).
#include "common.h"
#include "mobgt_hip.h"
#include "front_body.h"

namespace {

constexpr int MAXT = 4;

struct EmbedParams {
    const float* tables[MAXT];
    float* d_tables[MAXT];
    const void* idx[MAXT];
    int64_t skip[MAXT];
    int n_tables;
    float* out;
    const float* dout;
    int64_t R, ld;
    int C;
};

template <typename TI>
__global__ __launch_bounds__(256) void gather_sum_kernel(const EmbedParams p) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.R) return;
    const int lane = threadIdx.x & 63;
    int64_t rows[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) rows[t] = t < p.n_tables ? (int64_t)reinterpret_cast<const TI*>(p.idx[t])[r] : -1;
    for (int c = lane * 4; c < p.C; c += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < p.n_tables && rows[t] >= 0) {
                const float4 v = *reinterpret_cast<const float4*>(p.tables[t] + rows[t] * p.C + c);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        *reinterpret_cast<float4*>(p.out + r * p.ld + c) = acc;
    }
}

template <typename TI>
__global__ __launch_bounds__(256) void scatter_add_kernel(const EmbedParams p) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.R) return;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t >= p.n_tables) break;
        const int64_t row = (int64_t)reinterpret_cast<const TI*>(p.idx[t])[r];
        if (row < 0 || row == p.skip[t]) continue;
        float* dst = p.d_tables[t] + row * p.C;
        for (int c = lane; c < p.C; c += 64) atomicAdd(dst + c, p.dout[r * p.ld + c]);
    }
}

// Concatenation instead of the sum: out[r, off_t : off_t + W_t] = table_t[idx_t[r], :] (zeros where idx < 0), and the
// matching scatter -- `[poi ; time]` of model_fqandtoyo.py:1262-1268 in one launch each way.
struct ConcatParams {
    const float* tables[MAXT];
    float* d_tables[MAXT];
    const void* idx[MAXT];
    int64_t skip[MAXT];
    int width[MAXT], coff[MAXT];
    int n_tables;
    float* out;
    const float* dout;
    int64_t R, ld;
};

template <typename TI, bool BWD>
__global__ __launch_bounds__(256) void gather_concat_kernel(const ConcatParams p) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.R) return;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t >= p.n_tables) break;
        const int64_t row = (int64_t)reinterpret_cast<const TI*>(p.idx[t])[r];
        const int W = p.width[t];
        if (!BWD) {
            for (int c = lane * 4; c < W; c += 256) {
                const float4 v = row >= 0 ? *reinterpret_cast<const float4*>(p.tables[t] + row * W + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(p.out + r * p.ld + p.coff[t] + c) = v;
            }
        } else {
            if (row < 0 || row == p.skip[t]) continue;
            float* dst = p.d_tables[t] + row * W;
            for (int c = lane; c < W; c += 64) atomicAdd(dst + c, p.dout[r * p.ld + p.coff[t] + c]);
        }
    }
}


// Several gathers of one position list in ONE launch, each with its own destination (model_fqandtoyo.py:1259-1298: the
// [poi ; time] rows, the category rows and the additive degree / frequency / positional rows were three launches forward
// and three backward): job t copies table_t[idx_t[r], :] to buf_t[r, coff_t : coff_t + W_t] (zeros where idx < 0), or ADDS it
// there when accum_t (jobs run in order inside the wave that owns row r: a sum of tables is a copy followed by adds).
// Backward: buf_t is the gradient buffer, scattered with f32 atomics into d_table_t (rows equal to skip_t excepted).
constexpr int MAXJ = 8;
constexpr int MAXFOLD = 3;
struct MultiParams {
    const float* tables[MAXJ];
    float* d_tables[MAXJ];
    const void* idx[MAXJ];
    int64_t skip[MAXJ];
    int width[MAXJ], coff[MAXJ], accum[MAXJ];
    float* buf[MAXJ];
    int64_t ld[MAXJ];
    int n;
    int64_t R;
    const float* fold[MAXFOLD];             // forward, optional: n_fold more tables of job fold_job's width, read at that job's
    int n_fold, fold_job;                   // row and added (in order) to its value before it is stored / accumulated
    const float* extra_src;                 // backward, optional: a [W] vector added to ROW 0 of d_tables[extra_job] (the
    int extra_job;                          // graph-token row's share of pe[0]'s gradient, model_fqandtoyo.py:1338-1342)
};

template <typename TI, bool BWD>
__global__ __launch_bounds__(256) void gather_multi_kernel(const MultiParams p) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (BWD && p.extra_src && blockIdx.x == 0 && threadIdx.x < 64 && p.d_tables[p.extra_job])
        for (int c = lane; c < p.width[p.extra_job]; c += 64) atomicAdd(p.d_tables[p.extra_job] + c, p.extra_src[c]);
    if (r >= p.R) return;
#pragma unroll
    for (int t = 0; t < MAXJ; ++t) {
        if (t >= p.n) break;
        const int64_t row = (int64_t)reinterpret_cast<const TI*>(p.idx[t])[r];
        const int W = p.width[t];
        float* b = p.buf[t] + r * p.ld[t] + p.coff[t];
        if (!BWD) {
            const bool f1 = p.n_fold > 0 && t == p.fold_job, f2 = p.n_fold > 1 && t == p.fold_job, f3 = p.n_fold > 2 && t == p.fold_job;
            for (int c = lane * 4; c < W; c += 256) {
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 v = row >= 0 ? *reinterpret_cast<const float4*>(p.tables[t] + row * W + c) : z4;
                const float4 o1 = (f1 && row >= 0) ? *reinterpret_cast<const float4*>(p.fold[0] + row * W + c) : z4;
                const float4 o2 = (f2 && row >= 0) ? *reinterpret_cast<const float4*>(p.fold[1] + row * W + c) : z4;
                const float4 o3 = (f3 && row >= 0) ? *reinterpret_cast<const float4*>(p.fold[2] + row * W + c) : z4;
                if (f1) { v.x += o1.x; v.y += o1.y; v.z += o1.z; v.w += o1.w; }
                if (f2) { v.x += o2.x; v.y += o2.y; v.z += o2.z; v.w += o2.w; }
                if (f3) { v.x += o3.x; v.y += o3.y; v.z += o3.z; v.w += o3.w; }
                if (p.accum[t]) {
                    const float4 o = *reinterpret_cast<const float4*>(b + c);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *reinterpret_cast<float4*>(b + c) = v;
            }
        } else {
            if (row < 0 || row == p.skip[t] || !p.d_tables[t]) continue;
            float* dst = p.d_tables[t] + row * W;
            for (int c = lane; c < W; c += 64) atomicAdd(dst + c, b[c]);
        }
    }
}

// Run-length variant (C <= 512): a wave walks RUN_ROWS consecutive rows and keeps the sum of a run of EQUAL indices
// in registers, flushing with atomics only when the index changes.  Along a trajectory the degree rows (and the
// shared frequency row) repeat for long stretches: at G*N = 12.5 k rows the plain kernel spent 120 us per call
// serialising thousands of atomics on a handful of table rows.
constexpr int RUN_ROWS = 16;
constexpr int RUN_MAXK = 8;

// (round 4) the RUN_ROWS gradient rows are the same for every table: they are read ONCE, all in flight together, into
// registers (NK <= 4: 64 of them), and so are the wave's indices -- the first form read row after row, table after table, each
// read behind the previous one's add: 16 x n_tables dependent round trips, 115 us at S-BIG for 13 MB.
template <typename TI, int NK>
__global__ __launch_bounds__(256) void scatter_add_runs_kernel(const EmbedParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RUN_ROWS;
    if (r0 >= p.R) return;
    const int nrow = (int)min((int64_t)RUN_ROWS, p.R - r0);
    float src[RUN_ROWS][NK];
#pragma unroll
    for (int rr = 0; rr < RUN_ROWS; ++rr) {
        const float* from = p.dout + (r0 + (rr < nrow ? rr : 0)) * p.ld;
#pragma unroll
        for (int k = 0; k < NK; ++k) src[rr][k] = (rr < nrow && lane + 64 * k < p.C) ? from[lane + 64 * k] : 0.f;
    }
#pragma unroll 1
    for (int t = 0; t < p.n_tables; ++t) {
        const TI* idx = reinterpret_cast<const TI*>(p.idx[t]);
        // lane rr holds the index of row rr (one load for the wave)
        const int64_t mine = lane < nrow ? (int64_t)idx[r0 + lane] : -1;
        int64_t cur = -1;
        float acc[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) acc[k] = 0.f;
#pragma unroll
        for (int rr = 0; rr <= RUN_ROWS; ++rr) {
            int64_t row = -1;
            bool live = false;
            if (rr < RUN_ROWS) {
                row = __shfl(mine, rr);
                live = rr < nrow && row >= 0 && row != p.skip[t];
            }
            if (rr < RUN_ROWS && !live) continue;
            if (row != cur) {                                   // (rr == RUN_ROWS: final flush)
                if (cur >= 0) {
                    float* dst = p.d_tables[t] + cur * p.C;
#pragma unroll
                    for (int k = 0; k < NK; ++k)
                        if (lane + 64 * k < p.C) atomicAdd(dst + lane + 64 * k, acc[k]);
                }
                cur = row;
#pragma unroll
                for (int k = 0; k < NK; ++k) acc[k] = 0.f;
            }
            if (rr < RUN_ROWS) {
#pragma unroll
                for (int k = 0; k < NK; ++k) acc[k] += src[rr][k];
            }
        }
    }
}

}  // namespace

extern "C" int mobgt_embed_gather_sum(const float* const* tables_host, const void* const* idx_host, int n_tables,
                                      float* out, int64_t R, int C, int64_t ld_out, int idx_dtype, void* stream) {
    if (n_tables < 1 || n_tables > MAXT || C <= 0 || C % 4 != 0 || ld_out % 4 != 0) return MOBGT_EBADDIM;
    if (R <= 0) return 0;
    EmbedParams p = {};
    for (int t = 0; t < n_tables; ++t) { p.tables[t] = tables_host[t]; p.idx[t] = idx_host[t]; }
    p.n_tables = n_tables; p.out = out; p.R = R; p.C = C; p.ld = ld_out;
    const dim3 grid((unsigned)((R + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (idx_dtype == MOBGT_I64) hipLaunchKernelGGL(gather_sum_kernel<int64_t>, grid, block, 0, st, p);
    else if (idx_dtype == MOBGT_I32) hipLaunchKernelGGL(gather_sum_kernel<int32_t>, grid, block, 0, st, p);
    else if (idx_dtype == MOBGT_I16) hipLaunchKernelGGL(gather_sum_kernel<int16_t>, grid, block, 0, st, p);
    else return MOBGT_EDTYPE;
    return (int)hipGetLastError();
}

extern "C" int mobgt_embed_scatter_add(float* const* d_tables_host, const void* const* idx_host,
                                       const int64_t* skip_idx_host, int n_tables, const float* dout, int64_t R, int C,
                                       int64_t ld_dout, int idx_dtype, void* stream) {
    if (n_tables < 1 || n_tables > MAXT || C <= 0) return MOBGT_EBADDIM;
    if (R <= 0) return 0;
    EmbedParams p = {};
    for (int t = 0; t < n_tables; ++t) {
        p.d_tables[t] = d_tables_host[t]; p.idx[t] = idx_host[t]; p.skip[t] = skip_idx_host ? skip_idx_host[t] : -1;
    }
    p.n_tables = n_tables; p.dout = dout; p.R = R; p.C = C; p.ld = ld_dout;
    hipStream_t st = (hipStream_t)stream;
    if (C <= 64 * RUN_MAXK && R >= 4096) {            // (few rows: atomics do not pile up, and 16 sequential rows per wave cost latency)
        const dim3 grid((unsigned)((R + 4 * RUN_ROWS - 1) / (4 * RUN_ROWS))), block(256);
        const bool narrow = C <= 256;
        if (idx_dtype == MOBGT_I64) {
            if (narrow) hipLaunchKernelGGL((scatter_add_runs_kernel<int64_t, 4>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((scatter_add_runs_kernel<int64_t, RUN_MAXK>), grid, block, 0, st, p);
        } else if (idx_dtype == MOBGT_I32) {
            if (narrow) hipLaunchKernelGGL((scatter_add_runs_kernel<int32_t, 4>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((scatter_add_runs_kernel<int32_t, RUN_MAXK>), grid, block, 0, st, p);
        } else if (idx_dtype == MOBGT_I16) {
            if (narrow) hipLaunchKernelGGL((scatter_add_runs_kernel<int16_t, 4>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((scatter_add_runs_kernel<int16_t, RUN_MAXK>), grid, block, 0, st, p);
        } else return MOBGT_EDTYPE;
        return (int)hipGetLastError();
    }
    const dim3 grid((unsigned)((R + 3) / 4)), block(256);
    if (idx_dtype == MOBGT_I64) hipLaunchKernelGGL(scatter_add_kernel<int64_t>, grid, block, 0, st, p);
    else if (idx_dtype == MOBGT_I32) hipLaunchKernelGGL(scatter_add_kernel<int32_t>, grid, block, 0, st, p);
    else if (idx_dtype == MOBGT_I16) hipLaunchKernelGGL(scatter_add_kernel<int16_t>, grid, block, 0, st, p);
    else return MOBGT_EDTYPE;
    return (int)hipGetLastError();
}

// ---- index derivation for the node features ---------------------------------------------------------------
namespace {
using mobgt_front::NodeIndexParams;
__global__ __launch_bounds__(256) void node_index_kernel(const NodeIndexParams p) {
    __shared__ int s_cnt[4];
    mobgt_front::node_index_body(p, (int)blockIdx.x, s_cnt);
}
}  // namespace

extern "C" int mobgt_node_index(const void* x, int x_dtype, int64_t xs_g, int64_t xs_n, const float* time_normal, int64_t ts_g,
                                int64_t ts_n, const int64_t* poi2cat, const void* in_degree, const void* out_degree,
                                int deg_dtype, int64_t* idx, float* real, int G, int N, int rows_only, void* stream) {
    if (G <= 0 || N <= 0) return 0;
    if (in_degree && deg_dtype != MOBGT_I64 && deg_dtype != MOBGT_I32 && deg_dtype != MOBGT_I16) return MOBGT_EDTYPE;
    if (x_dtype != MOBGT_I64 && x_dtype != MOBGT_I32) return MOBGT_EDTYPE;
    NodeIndexParams p = {x, x_dtype, xs_g, xs_n, time_normal, ts_g, ts_n, poi2cat, in_degree, out_degree, deg_dtype, idx, real, G, N,
                         rows_only};
    hipLaunchKernelGGL(node_index_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

namespace {
template <bool BWD>
int launch_concat(const ConcatParams& p, int idx_dtype, hipStream_t st) {
    const dim3 grid((unsigned)((p.R + 3) / 4)), block(256);
    if (idx_dtype == MOBGT_I64) hipLaunchKernelGGL((gather_concat_kernel<int64_t, BWD>), grid, block, 0, st, p);
    else if (idx_dtype == MOBGT_I32) hipLaunchKernelGGL((gather_concat_kernel<int32_t, BWD>), grid, block, 0, st, p);
    else if (idx_dtype == MOBGT_I16) hipLaunchKernelGGL((gather_concat_kernel<int16_t, BWD>), grid, block, 0, st, p);
    else return MOBGT_EDTYPE;
    return (int)hipGetLastError();
}
int fill_concat(ConcatParams& p, const void* const* idx_host, const int* widths, int n_tables, int64_t R, int64_t ld) {
    if (n_tables < 1 || n_tables > MAXT || ld % 4 != 0) return MOBGT_EBADDIM;
    int off = 0;
    for (int t = 0; t < n_tables; ++t) {
        if (widths[t] <= 0 || widths[t] % 4 != 0) return MOBGT_EBADDIM;
        p.idx[t] = idx_host[t]; p.width[t] = widths[t]; p.coff[t] = off;
        off += widths[t];
    }
    if (off > ld) return MOBGT_EBADDIM;
    p.n_tables = n_tables; p.R = R; p.ld = ld;
    return 0;
}
}  // namespace

extern "C" int mobgt_embed_gather_concat(const float* const* tables_host, const void* const* idx_host, const int* widths_host,
                                         int n_tables, float* out, int64_t R, int64_t ld_out, int idx_dtype, void* stream) {
    ConcatParams p = {};
    const int rc = fill_concat(p, idx_host, widths_host, n_tables, R, ld_out);
    if (rc) return rc;
    if (R <= 0) return 0;
    for (int t = 0; t < n_tables; ++t) p.tables[t] = tables_host[t];
    p.out = out;
    return launch_concat<false>(p, idx_dtype, (hipStream_t)stream);
}

extern "C" int mobgt_embed_scatter_concat(float* const* d_tables_host, const void* const* idx_host, const int64_t* skip_idx_host,
                                          const int* widths_host, int n_tables, const float* dout, int64_t R, int64_t ld_dout,
                                          int idx_dtype, void* stream) {
    ConcatParams p = {};
    const int rc = fill_concat(p, idx_host, widths_host, n_tables, R, ld_dout);
    if (rc) return rc;
    if (R <= 0) return 0;
    for (int t = 0; t < n_tables; ++t) { p.d_tables[t] = d_tables_host[t]; p.skip[t] = skip_idx_host ? skip_idx_host[t] : -1; }
    p.dout = dout;
    return launch_concat<true>(p, idx_dtype, (hipStream_t)stream);
}

// ---- evaluation: rank of the target class in each row of the logits --------------------------------------
namespace {

// One workgroup per row.  Everything get_acc (model_fqandtoyo.py:48-90: "is the target among the top-k indices,
// and at which position") and MRR_metric (:122-131: position of the target in the descending argsort) need is
// the number of classes ranked ahead of the target: strictly larger scores, plus -- among equal scores -- the
// classes a given sort puts first (lower index for a stable descending top-k; HIGHER index for the reference's
// reversed ascending argsort).  Both counts are returned; for distinct scores they coincide.
__global__ __launch_bounds__(256) void target_rank_kernel(const float* __restrict__ scores, const int64_t* __restrict__ target,
                                                          int32_t* __restrict__ rank, int64_t V) {
    __shared__ int s_g[4], s_lo[4], s_hi[4];
    const int64_t row = blockIdx.x;
    const int64_t t = target[row];
    const float* s = scores + row * V;
    int greater = 0, tie_lo = 0, tie_hi = 0;
    if (t >= 0 && t < V) {
        const float ts = s[t];
        for (int64_t c = threadIdx.x; c < V; c += 256) {
            const float v = s[c];
            greater += v > ts;
            tie_lo += (v == ts) & (c < t);
            tie_hi += (v == ts) & (c > t);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        greater += __shfl_xor(greater, o, 64);
        tie_lo += __shfl_xor(tie_lo, o, 64);
        tie_hi += __shfl_xor(tie_hi, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_g[threadIdx.x >> 6] = greater; s_lo[threadIdx.x >> 6] = tie_lo; s_hi[threadIdx.x >> 6] = tie_hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = s_g[0] + s_g[1] + s_g[2] + s_g[3];
        const bool ok = t >= 0 && t < V;
        rank[2 * row] = ok ? g + s_lo[0] + s_lo[1] + s_lo[2] + s_lo[3] : -1;
        rank[2 * row + 1] = ok ? g + s_hi[0] + s_hi[1] + s_hi[2] + s_hi[3] : -1;
    }
}

}  // namespace

extern "C" int mobgt_target_rank(const float* scores, const int64_t* target, int32_t* rank, int64_t G, int64_t V,
                                 void* stream) {
    if (G <= 0 || V <= 0) return 0;
    hipLaunchKernelGGL(target_rank_kernel, dim3((unsigned)G), dim3(256), 0, (hipStream_t)stream, scores, target, rank, V);
    return (int)hipGetLastError();
}

// ---- rows of a bf16 matrix, gathered AND transposed in one pass ---------------------------------------------
// out_rows [R, C] = a[rows[j], :],  out_t [C, R] = out_rows^T.  64 x 64 tiles through LDS, 16-byte lanes both ways.
// (torch: index_select 6 us + a generic strided copy for the transpose 34 us at R = 608, C = 7856.)
namespace {
__global__ __launch_bounds__(256) void gather_rows_t_kernel(const uint16_t* __restrict__ a, int64_t ld, const int64_t* __restrict__ rows,
                                                            uint16_t* __restrict__ out_rows, uint16_t* __restrict__ out_t, int R, int C) {
    __shared__ uint16_t tile[64][72];                                   // [j][c], 144-byte pitch
    const int j0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int jl = pass * 32 + (t >> 3), cl = (t & 7) * 8;          // 8 lanes x 16 B per row
        const int j = j0 + jl, c = c0 + cl;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (j < R && c < C) {                                             // C % 8 == 0: a piece is in or out as a whole
            v = *reinterpret_cast<const uint4*>(a + rows[j] * ld + c);
            *reinterpret_cast<uint4*>(out_rows + (int64_t)j * C + c) = v;
        }
        *reinterpret_cast<uint4*>(&tile[jl][cl]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int cl = pass * 32 + (t >> 3), jl = (t & 7) * 8;          // 8 lanes x 16 B (= 8 j) per output row c
        const int c = c0 + cl, j = j0 + jl;
        if (c < C && j < R) {                                             // R % 8 == 0
            uint16_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = tile[jl + i][cl];
            *reinterpret_cast<uint4*>(out_t + (int64_t)c * R + j) = *reinterpret_cast<const uint4*>(v);
        }
    }
}
}  // namespace

extern "C" int mobgt_gather_rows_t(const void* a, int64_t ld, const int64_t* rows, void* out_rows, void* out_t, int R,
                                   int C, void* stream) {
    if (R <= 0 || C <= 0) return 0;
    if ((R & 7) || (C & 7) || (ld & 7)) return MOBGT_EBADDIM;
    if (((uintptr_t)a | (uintptr_t)out_rows | (uintptr_t)out_t) & 15) return MOBGT_EALIGN;
    hipLaunchKernelGGL(gather_rows_t_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint16_t*>(a), ld, rows, reinterpret_cast<uint16_t*>(out_rows),
                       reinterpret_cast<uint16_t*>(out_t), R, C);
    return (int)hipGetLastError();
}

/* n <= 8 gathers of one position list in one launch (see gather_multi_kernel) + up to three FOLDED tables (accum == 2: added
 * to the job in front of them, forward only); backward != 0: the scatter-add of the gradient buffers into d_tables (null
 * entries skipped). */
extern "C" int mobgt_embed_gather_multi(int n, const float* const* tables, float* const* d_tables, const void* const* idx,
                                        const int64_t* skip, const int* width, const int* coff, const int* accum,
                                        float* const* buf, const int64_t* ld, int64_t R, int idx_dtype, int backward,
                                        const float* extra_row0, int extra_job, void* stream) {
    if (n < 1 || n > MAXJ + MAXFOLD) return MOBGT_EBADDIM;
    if (extra_row0 && (extra_job < 0 || extra_job >= n || !backward)) return MOBGT_EBADDIM;
    if (R <= 0) return 0;
    MultiParams p = {};
    int k = 0;
    for (int t = 0; t < n; ++t) {
        if (accum && accum[t] == 2) {                       // folded into job k - 1
            if (backward || !tables || !tables[t] || k == 0 || p.n_fold >= MAXFOLD || (p.n_fold && p.fold_job != k - 1) ||
                width[t] != p.width[k - 1] || extra_row0)
                return MOBGT_EBADDIM;
            if ((uintptr_t)tables[t] & 15) return MOBGT_EALIGN;
            p.fold[p.n_fold++] = tables[t];
            p.fold_job = k - 1;
            continue;
        }
        if (k >= MAXJ) return MOBGT_EBADDIM;
        if (width[t] <= 0 || (width[t] & 3) || (coff[t] & 3) || (ld[t] & 3) || !buf[t]) return MOBGT_EBADDIM;
        if (((uintptr_t)buf[t] | (uintptr_t)(tables ? tables[t] : nullptr)) & 15) return MOBGT_EALIGN;
        p.tables[k] = tables ? tables[t] : nullptr;
        p.d_tables[k] = d_tables ? d_tables[t] : nullptr;
        p.idx[k] = idx[t]; p.skip[k] = skip ? skip[t] : -1;
        p.width[k] = width[t]; p.coff[k] = coff[t]; p.accum[k] = accum ? accum[t] : 0;
        p.buf[k] = buf[t]; p.ld[k] = ld[t];
        ++k;
    }
    n = k;
    p.n = n; p.R = R; p.extra_src = extra_row0; p.extra_job = extra_job;
    const dim3 grid((unsigned)((R + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (idx_dtype == MOBGT_I64) {
        if (backward) hipLaunchKernelGGL((gather_multi_kernel<int64_t, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gather_multi_kernel<int64_t, false>), grid, block, 0, st, p);
    } else if (idx_dtype == MOBGT_I32) {
        if (backward) hipLaunchKernelGGL((gather_multi_kernel<int32_t, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gather_multi_kernel<int32_t, false>), grid, block, 0, st, p);
    } else return MOBGT_EDTYPE;
    return (int)hipGetLastError();
}
