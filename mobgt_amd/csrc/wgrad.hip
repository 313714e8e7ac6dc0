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
#include <cstdlib>
#include "common.h"
#include "mobgt_hip.h"
#include "gemm_body.h"
#include "wgrad_body.h"
#include "hop_body.h"
#include <stdlib.h>

namespace {

using namespace mobgt_wgrad;

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
    // Optional passengers behind the weight-gradient workgroups: the hop table's backward (csrc/hop_body.h), which needs nothing
    // the group produces and whose results nothing but the optimizer reads
    mobgt_hop::HopBwd hop;
    int n_hop, n_wg;                  // hop workgroups; first workgroup that is not a weight-gradient one (n_hop > 0)
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
    if (grp.n_hop > 0 && bid >= grp.n_wg) {
        mobgt_hop::hop_table_bwd8_body(grp.hop, bid - grp.n_wg, lds);
        return;
    }
    int q = 0;
#pragma unroll
    for (int t = 1; t < WG_GROUP; ++t)
        if (t < grp.n && bid >= grp.first[t]) q = t;
    const int local = bid - grp.first[q];
    if (grp.p[q].in_f32) wgrad_body<true, NWAVE>(grp.p[q], local % grp.tiles[q], local / grp.tiles[q], grp.splits[q], lds);
    else wgrad_body<false, NWAVE>(grp.p[q], local % grp.tiles[q], local / grp.tiles[q], grp.splits[q], lds);
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
    grp.n_hop = 0; grp.n_wg = 0;
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

// Heterogeneous group: up to 32 weight-gradient problems with their OWN row counts, operand dtypes and activation masks in one
// launch (round 3: the leaf weight gradients of a train step -- FuseEmbeddings' Linears, the distance GCN's three layers -- were
// six launches of ~9.5 us each, every one a handful of workgroups walking a few thousand rows; nothing but the optimizer
// reads their results, so the trainer collects them and issues them together at the end of the backward pass).
extern "C" int mobgt_linear_wgrad_multi_hop(int n, const void* const* g, const int64_t* ldg, const void* const* x, const int64_t* ldx,
                                            const float* const* g_mask, const float* const* x_mask, const float* mask_vals,
                                            float* const* dw, const int64_t* ldw, float* const* db, const int* db_of_x,
                                            const int64_t* R, const int* M, const int* N, const int* in_f32,
                                            // the arguments of mobgt_hop_table_bwd (with_hop != 0; H = 8, n_edge <= 256: its 16 n_edge floats of staging live in the group kernel's LDS)
                                            int with_hop, const float* d_table, const float* edge_encoder, const float* edge_dis_encoder,
                                            float* d_edge_encoder, float* d_edge_dis_encoder, int D, int n_edge, int fp16_roundtrip,
                                            void* stream) {
    if (n < 1 || n > WG_GROUP) return MOBGT_EBADDIM;
    WgradGroup grp;
    grp.n = n;
    grp.n_tail = 0;
    grp.n_hop = 0; grp.n_wg = 0;
    int total = 0;
    // (16-wave workgroups whenever any problem is long: a short problem then simply has idle waves)
    int nwave = 8;
    for (int q = 0; q < n; ++q)
        if (R[q] > SHORT_R) nwave = 16;
    for (int q = 0; q < n; ++q) {
        if (in_f32[q] != 0 && in_f32[q] != 1) return MOBGT_EDTYPE;
        if ((g_mask[q] || x_mask[q]) && !in_f32[q]) return MOBGT_EDTYPE;          // masks exist for f32 operands only
        // Workgroups per problem: 1024 slots shared by the problems, at least 64 each -- fill_problem caps a problem at one
        // workgroup per 16-wave slab of rows, i.e. a 7 856-row problem gets its 16 splits and every wave ONE 32-row step: the
        // launch is load -> MFMA -> reduce -> atomics once, whatever the row count.  (Round 4 probe, docs/NOTEBOOK.md, the
        // S-FSQ step's six problems together: 512 slots 24.4 us, 1024 slots 19.7, more: no change; FEWER workgroups -- one round
        // of resident ones, 256 -- 30 us: the chain of 32-row steps per wave is what costs, not the second round.)
        const int slots = 1024;
        const int rc = fill_problem(grp.p[q], g[q], ldg[q], x[q], ldx[q], dw[q], ldw[q], db[q], R[q], M[q], N[q],
                                    slots / n > 64 ? slots / n : 64, &grp.tiles[q], &grp.splits[q], in_f32[q], nwave);
        if (rc) return rc;
        if ((((uintptr_t)g_mask[q] | (uintptr_t)x_mask[q]) & 7)) return MOBGT_EALIGN;
        grp.p[q].gmask = g_mask[q]; grp.p[q].xmask = x_mask[q];
        grp.p[q].mpos = mask_vals[3 * q]; grp.p[q].mneg = mask_vals[3 * q + 1]; grp.p[q].mzero = mask_vals[3 * q + 2];
        grp.p[q].db_x = db_of_x[q];
        grp.first[q] = total;
        total += grp.tiles[q] * grp.splits[q];
    }
    for (int q = n; q <= WG_GROUP; ++q) grp.first[q] = total;
    for (int q = n; q < WG_GROUP; ++q) { grp.tiles[q] = 1; grp.splits[q] = 1; grp.p[q] = grp.p[0]; }
    if (with_hop) {
        if (D <= 0 || n_edge <= 0 || n_edge > 256 || (((uintptr_t)d_table | (uintptr_t)edge_dis_encoder) & 15)) return MOBGT_EBADDIM;
        grp.hop = mobgt_hop::HopBwd{d_table, edge_encoder, edge_dis_encoder, d_edge_encoder, d_edge_dis_encoder, D, n_edge, fp16_roundtrip};
        grp.n_hop = mobgt_hop::hop_bwd8_blocks(D, n_edge);
        grp.n_wg = total;
        total += grp.n_hop;
    }
    if (nwave == 8) hipLaunchKernelGGL(wgrad_group_kernel<8>, dim3(total), dim3(8 * 64), 0, (hipStream_t)stream, grp);
    else hipLaunchKernelGGL(wgrad_group_kernel<16>, dim3(total), dim3(16 * 64), 0, (hipStream_t)stream, grp);
    return (int)hipGetLastError();
}

extern "C" int mobgt_linear_wgrad_multi(int n, const void* const* g, const int64_t* ldg, const void* const* x, const int64_t* ldx,
                                        const float* const* g_mask, const float* const* x_mask, const float* mask_vals,
                                        float* const* dw, const int64_t* ldw, float* const* db, const int* db_of_x,
                                        const int64_t* R, const int* M, const int* N, const int* in_f32, void* stream) {
    return mobgt_linear_wgrad_multi_hop(n, g, ldg, x, ldx, g_mask, x_mask, mask_vals, dw, ldw, db, db_of_x, R, M, N, in_f32, 0, nullptr,
                                        nullptr, nullptr, nullptr, nullptr, 0, 0, 0, stream);
}

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
