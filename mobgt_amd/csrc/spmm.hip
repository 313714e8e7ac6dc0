// GraphConvolution's adjacency product (graphormer/modelGNN.py:38-44: `torch.spmm(adj, support)`) for a POI graph held as
// CSR.  The dense normalised adjacency the reference builds ((D+I)^-1 (A+I), model_fqandtoyo.py:481-486) has P^2 entries:
// 123 MB in bf16 at P = 7 856 (streamed four times per step) and simply does not exist at P = 100 000 (S-BIG, BASELINE
// configs[4]: 20 GB in bf16, 80 GB in the reference's float64 construction).  With ~30 neighbours per POI the product is
// a gather of ~30 rows of `support` per output row.
//
//   spmm      out[i, :]  = bias + sum_e val[e] * B[col[e], :]       e in [rowptr[r], rowptr[r+1]),  r = rows ? rows[i] : i
//   spmm_t    dB[col[e], :] += val[e] * g[i, :]                      the transposed product for a row SUBSET (atomics;
//             mobgt_spmm_csr_t_rows_gather below: the same product as a gather over the CSR of the transpose);
//             the full transposed product is `spmm` on the stored CSR of the transpose.
//
// One wave per output row, 16-byte lanes over the feature axis (C = 64 / 128: 256 / 512 contiguous bytes per gathered
// row), the neighbour loop unrolled by four so that four independent row reads are in flight.  HBM/L2-bound gather;
// f32 in, f32 accumulate, f32 out (the dense path's bf16 rounding of the adjacency does not arise).
#include "common.h"
#include "mobgt_hip.h"

namespace {

struct SpmmParams {
    const int64_t* rowptr;     // [n_rows_of_A + 1]
    const int32_t* col;        // [nnz]
    const float* val;          // [nnz]
    const int64_t* rows;       // [R] row subset of A, or null (then R = n_rows_of_A)
    const float* B;            // [n_cols_of_A, C] (ldb)
    const float* bias;         // [C] or null
    float* out;                // spmm: [R, C] (ldo);  spmm_t: dB [n_cols_of_A, C] (ldo), accumulated
    const float* g;            // spmm_t: [R, C] (ldg)
    int64_t R, ldb, ldo, ldg;
    int C;
};

__global__ __launch_bounds__(256) void spmm_kernel(const SpmmParams p) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.R) return;
    const int lane = threadIdx.x & 63;
    const int64_t r = p.rows ? p.rows[i] : i;
    const int64_t e0 = p.rowptr[r], e1 = p.rowptr[r + 1];
    for (int c = lane * 4; c < p.C; c += 256) {
        float4 acc = p.bias ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t e = e0;
        for (; e + 4 <= e1; e += 4) {
            float4 v[4];
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w[u] = p.val[e + u];
                v[u] = *reinterpret_cast<const float4*>(p.B + (int64_t)p.col[e + u] * p.ldb + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc.x = fmaf(w[u], v[u].x, acc.x); acc.y = fmaf(w[u], v[u].y, acc.y);
                acc.z = fmaf(w[u], v[u].z, acc.z); acc.w = fmaf(w[u], v[u].w, acc.w);
            }
        }
        for (; e < e1; ++e) {
            const float w = p.val[e];
            const float4 v = *reinterpret_cast<const float4*>(p.B + (int64_t)p.col[e] * p.ldb + c);
            acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
        }
        *reinterpret_cast<float4*>(p.out + i * p.ldo + c) = acc;
    }
}

// The same product for narrow feature rows (C = 64 / 128: LPR = C / 4 = 16 / 32 lanes cover a row): the wave's 64 / LPR lane
// groups gather DIFFERENT neighbours at once and the groups' sums meet in a shuffle at the end.  With one neighbour per load
// instruction three quarters (C = 64) of every instruction's lanes were idle, and the load unit takes its 16 clocks per wave
// instruction whatever it carries: 3 M instructions per product at P = 100 000 -- col, val and the row, per edge -- were the
// kernel's time (round 5: 124 us per product).  Here col / val / row are one instruction each per 64 / LPR edges.
template <int LPR>
__global__ __launch_bounds__(256) void spmm_narrow_kernel(const SpmmParams p) {
    constexpr int NGR = 64 / LPR;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.R) return;
    const int lane = threadIdx.x & 63, grp = lane / LPR, c = (lane % LPR) * 4;
    const int64_t r = p.rows ? p.rows[i] : i;
    const int64_t e0 = p.rowptr[r], e1 = p.rowptr[r + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t e = e0 + grp;
    for (; e + 3 * NGR < e1; e += 4 * NGR) {             // four edges per group in flight
        float4 v[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            w[u] = p.val[e + u * NGR];
            v[u] = *reinterpret_cast<const float4*>(p.B + (int64_t)p.col[e + u * NGR] * p.ldb + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc.x = fmaf(w[u], v[u].x, acc.x); acc.y = fmaf(w[u], v[u].y, acc.y);
            acc.z = fmaf(w[u], v[u].z, acc.z); acc.w = fmaf(w[u], v[u].w, acc.w);
        }
    }
    for (; e < e1; e += NGR) {
        const float w = p.val[e];
        const float4 v = *reinterpret_cast<const float4*>(p.B + (int64_t)p.col[e] * p.ldb + c);
        acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (grp == 0) {
        if (p.bias) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + c);
            acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
        }
        *reinterpret_cast<float4*>(p.out + i * p.ldo + c) = acc;
    }
}

__global__ __launch_bounds__(256) void spmm_t_rows_kernel(const SpmmParams p) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.R) return;
    const int lane = threadIdx.x & 63;
    const int64_t r = p.rows ? p.rows[i] : i;
    const int64_t e0 = p.rowptr[r], e1 = p.rowptr[r + 1];
    for (int c = lane; c < p.C; c += 64) {
        const float gv = p.g[i * p.ldg + c];
        if (gv == 0.f) continue;
        for (int64_t e = e0; e < e1; ++e) atomicAdd(p.out + (int64_t)p.col[e] * p.ldo + c, p.val[e] * gv);
    }
}

// The row-subset transposed product WITHOUT atomics (the scatter above adds ~400 MB per S-BIG step through the memory-side
// atomic units: 306 us).  With the CSR of the transpose, destination row j gathers over its in-edges (i -> j) and keeps
// those whose source i is one of the subset's rows; `rows` may name a row several times, so the subset is first threaded
// into per-row lists: head[i] = last k with rows[k] == i (or -1), nxt[k] = the previous such k.
//   link   : nxt[k] = atomicExch(&head[rows[k]], k)          (head is all -1 on entry)
//   gather : db[j,:] = sum over in-edges e of j, over the list k of t_col[e]:  t_val[e] * g[k,:]      (every row written)
//   unlink : head[rows[k]] = -1
__global__ __launch_bounds__(256) void rows_link_kernel(const int64_t* __restrict__ rows, int R, int* __restrict__ head,
                                                        int* __restrict__ nxt, int unlink) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= R) return;
    if (unlink) head[rows[k]] = -1;
    else nxt[k] = atomicExch(&head[rows[k]], k);
}

struct SpmmTgParams {
    const int64_t* t_rowptr; const int32_t* t_col; const float* t_val;
    const int* head; const int* nxt;
    const float* g; float* out;
    int64_t P, ldg, ldo;
    int C;
};

__global__ __launch_bounds__(256) void spmm_t_gather_kernel(const SpmmTgParams p) {
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= p.P) return;
    const int lane = threadIdx.x & 63;
    const int64_t e0 = p.t_rowptr[j], e1 = p.t_rowptr[j + 1];
    // C <= 512: up to two float4 per lane
    float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    for (int64_t eb = e0; eb < e1; eb += 64) {
        const int64_t e = eb + lane;
        int k = -1;
        float v = 0.f;
        if (e < e1) { k = p.head[p.t_col[e]]; v = p.t_val[e]; }
        unsigned long long hits = __ballot(k >= 0);
        while (hits) {
            const int l = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            int kk = __shfl(k, l, 64);
            const float vv = __shfl(v, l, 64);
            while (kk >= 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = lane * 4 + 256 * u;
                    if (c < p.C) {
                        const float4 gv = *reinterpret_cast<const float4*>(p.g + (int64_t)kk * p.ldg + c);
                        acc[u].x = fmaf(vv, gv.x, acc[u].x); acc[u].y = fmaf(vv, gv.y, acc[u].y);
                        acc[u].z = fmaf(vv, gv.z, acc[u].z); acc[u].w = fmaf(vv, gv.w, acc[u].w);
                    }
                }
                kk = p.nxt[kk];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = lane * 4 + 256 * u;
        if (c < p.C) *reinterpret_cast<float4*>(p.out + j * p.ldo + c) = acc[u];
    }
}

}  // namespace

extern "C" int mobgt_spmm_csr_t_rows_gather(const int64_t* t_rowptr, const int32_t* t_col, const float* t_val, const int64_t* rows,
                                            int* head, int* nxt, const float* g, int64_t ldg, float* db, int64_t ld_db, int64_t P,
                                            int64_t R, int C, void* stream) {
    if (P <= 0) return 0;
    if (C <= 0 || (C & 3) || C > 512 || (ldg & 3) || (ld_db & 3) || R > 0x7fffffff) return MOBGT_EBADDIM;
    if (((uintptr_t)g | (uintptr_t)db) & 15) return MOBGT_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (R <= 0) {      // an empty row subset: the contract is "every row of db is written", so the result is all zeros
        if (ld_db == C) return (int)hipMemsetAsync(db, 0, (size_t)P * C * sizeof(float), st);
        return (int)hipMemset2DAsync(db, (size_t)ld_db * sizeof(float), 0, (size_t)C * sizeof(float), (size_t)P, st);
    }
    const dim3 lgrid((unsigned)((R + 255) / 256)), block(256);
    hipLaunchKernelGGL(rows_link_kernel, lgrid, block, 0, st, rows, (int)R, head, nxt, 0);
    SpmmTgParams p = {};
    p.t_rowptr = t_rowptr; p.t_col = t_col; p.t_val = t_val; p.head = head; p.nxt = nxt; p.g = g; p.out = db;
    p.P = P; p.ldg = ldg; p.ldo = ld_db; p.C = C;
    hipLaunchKernelGGL(spmm_t_gather_kernel, dim3((unsigned)((P + 3) / 4)), block, 0, st, p);
    hipLaunchKernelGGL(rows_link_kernel, lgrid, block, 0, st, rows, (int)R, head, nxt, 1);
    return (int)hipGetLastError();
}

extern "C" int mobgt_spmm_csr(const int64_t* rowptr, const int32_t* col, const float* val, const int64_t* rows,
                              const float* b, int64_t ldb, const float* bias, float* out, int64_t ld_out, int64_t R, int C,
                              void* stream) {
    if (R <= 0) return 0;
    if (C <= 0 || (C & 3) || (ldb & 3) || (ld_out & 3)) return MOBGT_EBADDIM;
    if (((uintptr_t)b | (uintptr_t)out | (uintptr_t)bias) & 15) return MOBGT_EALIGN;
    SpmmParams p = {};
    p.rowptr = rowptr; p.col = col; p.val = val; p.rows = rows; p.B = b; p.bias = bias; p.out = out;
    p.R = R; p.ldb = ldb; p.ldo = ld_out; p.C = C;
    const dim3 grid((unsigned)((R + 3) / 4)), block(256);
    if (C == 64) hipLaunchKernelGGL(spmm_narrow_kernel<16>, grid, block, 0, (hipStream_t)stream, p);
    else if (C == 128) hipLaunchKernelGGL(spmm_narrow_kernel<32>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(spmm_kernel, grid, block, 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int mobgt_spmm_csr_t_rows(const int64_t* rowptr, const int32_t* col, const float* val, const int64_t* rows,
                                     const float* g, int64_t ldg, float* db, int64_t ld_db, int64_t R, int C, void* stream) {
    if (R <= 0) return 0;
    if (C <= 0) return MOBGT_EBADDIM;
    SpmmParams p = {};
    p.rowptr = rowptr; p.col = col; p.val = val; p.rows = rows; p.g = g; p.out = db;
    p.R = R; p.ldg = ldg; p.ldo = ld_db; p.C = C;
    hipLaunchKernelGGL(spmm_t_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
