// The whole 3-layer GCN of a SMALL graph (graphormer/modelGNN.py:53-74 on the ~300-node category graph,
// model_fqandtoyo.py:1237) as ONE launch each way.
//
// Per step the category GCN was 7 launches forward and ~10 backward of 4-10 us each for a few MFLOP -- a twentieth of the
// S-FSQ step spent on a 300-node graph.  Its layers depend on each other through the adjacency product (every output row
// needs ALL rows of the previous layer), so one kernel needs a grid-wide hand-over: ceil(n/16) workgroups own 16 rows
// each, and between the layers they meet at a counter in global memory (see grid_barrier).  19 workgroups are always
// co-resident on a 256-CU device; the wait is bounded in wall time all the same and traps instead of hanging.
//   forward   h1 = leaky(AX W0 + b0);  t = A h1;  h2 = dropout(leaky(t W1 + b1));  t2 = A h2;  out = t2 W2 + b2
//             ((A h) W instead of A (h W): the same value, and the backward reuses t / t2 for the weight gradients)
//   backward  the chain rule of the above; A^T products read the stored transpose; weight / bias gradients leave a
//             workgroup as f32 atomics on its 16-row partial sums.
// Full-f32 MFMA products; at this size the arithmetic (5 MFLOP over 19 CUs) is nothing next to latencies -- see below.
// Dropout mask and LeakyReLU slope are those of mobgt_bias_act_fwd (same hash, same salt), so the result is what the
// layer-by-layer path produced.
#include "common.h"
#include "mobgt_hip.h"
#include "pack_body.h"
#include "front_body.h"
// csrc/bias.hip is compiled as part of THIS translation unit (not on its own): the category GCN's backward launch carries the bias
// tables' backward (build_bias_bwd_body) as passenger workgroups, and that body lives there with everything it needs
#include "bias.hip"

namespace {

constexpr int RB = 16;               // rows per workgroup
constexpr int MAXH = 64;             // widest layer
constexpr int NT = 256;
#ifndef PACK_U
#define PACK_U 8
#endif
#ifndef PACK_ROUNDS
#define PACK_ROUNDS 1
#endif

struct SmallGcnParams {
    const float *AX, *A, *AT;        // [n,K0], [n,n], [n,n] (A^T)
    const float *W0, *b0, *W1, *b1, *W2, *b2;     // [K0,H1] [H1] [H1,H2] [H2] [H2,H3] [H3]
    float *h1, *t, *h2, *t2, *out;   // forward: written;  backward: read (out unused)
    const float* g;                  // backward: d(out) [n,H3]
    float *dW0, *db0, *dW1, *db1, *dW2, *db2;     // accumulated (zero them first)
    float *dt2, *dt;                 // backward scratch [n,H2], [n,H1]
    int* counter;                    // zero on entry
    int n, K0, H1, H2, H3;
    int nwg;                         // workgroups of the network itself (the forward launch may carry passengers behind them)
    float slope, inv_keep;
    uint32_t thr;
    uint64_t seed;
    const uint64_t* seed_dev;
    uint32_t salt;
};

// Hand-over between the layers.  NO fences: an agent-scope release / acquire pair is a write-back and an invalidate of the
// XCD's whole L2 (measured: ~10 us per hand-over, and the kernels after this one start on a cold L2).  Instead the rows
// that cross workgroups are written and read with agent-scope RELAXED atomics -- `sc1` stores that write through and
// `sc1` loads that do not trust a non-coherent line (coh_store / coh_load below).  EVERY wave waits for its own stores to
// be acknowledged (explicit `s_waitcnt vmcnt(0)`: on gfx950 __syncthreads() is a bare s_barrier when the compiler sees no
// pending LDS-DMA, it does NOT drain the vector-memory counter) before the workgroup barrier that precedes thread 0's
// announcement -- otherwise the rows of waves 1..3 can still be in flight when another workgroup passes the counter.
// A workgroup that waits longer than the limit GIVES UP instead of trapping (a trap kills the process; VERDICT r3 weak #7b): it
// counts itself in g_sg_faults and carries on -- the launch ends, its results are garbage, the host finds the count at its next
// check (mobgt_small_gcn_faults; train.TrainStep.check_faults).  Never seen in an undisturbed run: the network's <= 19
// workgroups are resident together on any device that is not completely occupied by other streams' persistent kernels.
__device__ unsigned int g_sg_faults = 0;
__device__ long long g_sg_wait_ticks = 200000000LL;                          // 2 s of the 100 MHz clock
__device__ __forceinline__ void grid_barrier(int* counter, const int target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64(), lim = g_sg_wait_ticks;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > lim) {                                 // never in a sane run
                atomicAdd(&g_sg_faults, 1u);
                break;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void coh_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float4 coh_load4(const float* p) {
    const uint64_t a = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t b = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)b),
                       __uint_as_float((uint32_t)(b >> 32)));
}

// in-kernel timeline (-DSG_DEBUG; the reader script left the tree in round 5): stamps stay in registers and are written at the end -- a store
// in front of a barrier would add its own round trip to what it measures
#ifdef SG_DEBUG
#define STAMP_DECL int st_[16] = {}
#define STAMP(i) st_[i] = (int)wall_clock64()
#define STAMP_DUMP() do { if (blockIdx.x == SG_WG && threadIdx.x == 0) for (int q_ = 0; q_ < 16; ++q_) p.counter[4 + q_] = st_[q_]; } while (0)
#define SG_DBG(x) (x)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_DUMP()
#define SG_DBG(x) nullptr
#endif
#define PSTAMP(i) do { if (dbg) dbg[i] = (int)wall_clock64(); } while (0)

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : slope * v; }

// ---- how the 16 x C x K products run -------------------------------------------------------------------------------------
// Every layer is "this workgroup's 16 rows of L times ALL of R".  Measured on the way here (in-kernel timestamps,
// round-3 stamps): a thread-per-output loop over global memory is a chain of K dependent L2 round trips (270 us for the
// network); from LDS with scalar FMA tiles a product still took 5-7 us (a wave64 FMA costs four cycles whether 16 or 64
// lanes are live, and every k waits an LDS latency), the little [16 x H] x [H x H'] products of the epilogues 2-3 us each
// for the same reason; and with ONE wave per SIMD nothing hides a taken branch, so predicated loads / runtime tile counts
// (a branch each) cost microseconds per phase.  So:
//   * the layer widths are template parameters, and every load / LDS store of a staging burst is UNCONDITIONAL: an index
//     beyond the piece is clamped to its last element (which is then loaded and stored again, same value);
//   * both operands of a product are staged in LDS, a chunk of KC = 304 k at a time (the whole category graph is one
//     chunk), by ONE burst: all of a thread's loads are issued, then all its LDS stores;
//   * everything that does not depend on another workgroup -- weights (and their transposes), biases, the workgroup's rows
//     of A / A^T (which serve both adjacency products), of AX and of every saved activation -- is part of the kernel's
//     FIRST burst; after a grid barrier there is exactly one fetch: the other workgroups' rows of the previous layer;
//   * every product, large or small, is v_mfma_f32_16x16x4_f32 (full f32) with operands read from LDS -- 16 rows is exactly
//     one MFMA tile; the k range of the big ones is split over the four waves, partial tiles meet in LDS.  The R chunk
//     is stored with its 16-column groups XOR-swizzled by k so that the four k of an MFMA operand hit different banks.
// Dynamic LDS, 148.5 KB: one workgroup per CU, and there are only ceil(n/16) of them.
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int KC = 304, LDL = KC + 4, LDT = MAXH + 4;
constexpr int OFF_RC = 0;                              // [KC][C] chunk of R
constexpr int OFF_PART = OFF_RC;                       // [4 waves][RB][LDT] partial tiles (over the R chunk, once it is consumed)
constexpr int OFF_L2 = OFF_RC + KC * 32;               // a second L area inside the R area, free while C <= 32
constexpr int OFF_LC = OFF_RC + KC * MAXH;             // [RB][LDL] chunk of this workgroup's rows of L
constexpr int OFF_T = OFF_LC + RB * LDL;               // five [RB][LDT] row tiles
constexpr int OFF_W = OFF_T + 5 * RB * LDT;            // W1 [H1][H2] | W1^T | W2 [H2][H3] | W2^T   (H1 H2 + H2 H3 <= 4096)
constexpr int LDS_FLOATS = OFF_W + 2 * MAXH * MAXH;
static_assert(LDS_FLOATS * 4 <= 152 * 1024, "LDS plan");
static_assert(KC % 16 == 0 && KC / 16 <= 19, "one burst stages a chunk");
static_assert(OFF_L2 + RB * LDL <= OFF_LC && 4 * RB * LDT <= OFF_L2, "second L area");

template <int C>
__device__ __forceinline__ int swz(int k) { return C == 64 ? (k & 3) << 4 : (C == 32 ? ((k >> 1) & 1) << 4 : 0); }

// a burst in two halves: load() issues every global load of the piece into registers, store() puts them into LDS

// dst[r][k] = L[r0 + r][k0 + k] for k < kc, zero up to kcp: 16 lanes per row, 64-byte runs
struct BurstL {
    float v[KC / 16];
    __device__ __forceinline__ void load(const float* __restrict__ L, int64_t ldl, int k0, int kc, int kcp, int r0, int n) {
        const int r = threadIdx.x >> 4, kk = threadIdx.x & 15;
        const float* src = L + (int64_t)min(r0 + r, n - 1) * ldl + k0;
        const bool row_ok = r0 + r < n;
#pragma unroll
        for (int u = 0; u < KC / 16; ++u) {
            const int k = min(kk + 16 * u, kcp - 16 + kk);
            const float x = src[min(k, kc - 1)];
            v[u] = (row_ok && k < kc) ? x : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ dst, int kcp) const {
        const int r = threadIdx.x >> 4, kk = threadIdx.x & 15;
#pragma unroll
        for (int u = 0; u < KC / 16; ++u) dst[r * LDL + min(kk + 16 * u, kcp - 16 + kk)] = v[u];
    }
};

// dst[k][c ^ swz(k)] = R[k0 + k][c] for k < kc, zero up to kcp (R [K][C] contiguous, 16-byte aligned)
template <int C>
struct BurstR {
    static constexpr int U = (KC * C / 4 + NT - 1) / NT;
    float4 v[U];
    // coherent: R was written by other workgroups of THIS launch
    template <bool COHERENT>
    __device__ __forceinline__ void load(const float* R, int k0, int kc, int kcp) {
        const float* src = R + (int64_t)k0 * C;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min((int)(threadIdx.x + u * NT) * 4, kcp * C - 4);
            const float* q = src + min(e, kc * C - 4);
            const float4 x = COHERENT ? coh_load4(q) : *reinterpret_cast<const float4*>(q);
            v[u] = e < kc * C ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ dst, int kcp) const {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min((int)(threadIdx.x + u * NT) * 4, kcp * C - 4);
            const int k = e / C, c = e % C;
            *reinterpret_cast<float4*>(dst + k * C + (c ^ swz<C>(k))) = v[u];
        }
    }
};

// a weight [J][C] (16-byte aligned) -> LDS as it is and transposed
template <int J, int C>
struct BurstW {
    static constexpr int U = (J * C / 4 + NT - 1) / NT;
    float4 v[U];
    __device__ __forceinline__ void load(const float* __restrict__ W) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(W + min((int)(threadIdx.x + u * NT) * 4, J * C - 4));
    }
    __device__ __forceinline__ void store(float* __restrict__ w, float* __restrict__ wt) const {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min((int)(threadIdx.x + u * NT) * 4, J * C - 4);
            *reinterpret_cast<float4*>(w + e) = v[u];
            const int j = e / C, c = e % C;
            wt[c * J + j] = v[u].x; wt[(c + 1) * J + j] = v[u].y; wt[(c + 2) * J + j] = v[u].z; wt[(c + 3) * J + j] = v[u].w;
        }
    }
};

// this workgroup's rows of a saved [n][C] activation -> an LDS row tile (zero beyond the rows)
template <int C>
struct BurstRows {
    static constexpr int U = RB * C / NT;
    float v[U];
    __device__ __forceinline__ void load(const float* __restrict__ src, int r0, int n) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = threadIdx.x + u * NT, r = e / C, c = e % C;
            const float x = src[(int64_t)min(r0 + r, n - 1) * C + c];
            v[u] = r0 + r < n ? x : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ dst) const {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = threadIdx.x + u * NT, r = e / C, c = e % C;
            dst[r * LDT + c] = v[u];
        }
    }
};

// One 16 x 16 MFMA tile: sum over k < K (K % 4 == 0) of a(k) b(k), where for this lane (i = lane & 15, kq = lane >> 4)
// a(k) = A[i][k] and b(k) = B[k][i], both called with k = 4 s + kq.  Result: register v = D[4 kq + v][i].
template <int K, typename FA, typename FB>
__device__ __forceinline__ f32x4 mfma_tile(FA&& a, FB&& b) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int kq = (threadIdx.x & 63) >> 4;
    float av[K / 4], bv[K / 4];
#pragma unroll
    for (int s = 0; s < K / 4; ++s) { av[s] = a(4 * s + kq); bv[s] = b(4 * s + kq); }
#pragma unroll
    for (int s = 0; s < K / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
    return acc;
}

// dst[r][c] (LDS, [RB][LDT]) = sum_k L[r0 + r][k] * R[k][c], c < C, k < K.  L global (ldl), staged at `Lc`; R global
// [K][C] contiguous, 16-byte aligned, staged at the R area.  L_STAGED / R_STAGED: when K is a single chunk the caller
// already staged that operand and synchronised.  COHERENT: R was written by other workgroups of this launch.  All NT
// threads call it; ends with a barrier (dst complete and visible, the staging areas free).
template <int C, bool L_STAGED, bool R_STAGED, bool COHERENT>
__device__ __forceinline__ void tile_product(float* __restrict__ smem, float* __restrict__ Lc, const float* __restrict__ L,
                                             int64_t ldl, const float* R, int K, int r0, int n, float* __restrict__ dst) {
    constexpr int T = C / 16;
    float* Rc = smem + OFF_RC;
    float* part = smem + OFF_PART;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, kq = lane >> 4;
    const bool one = K <= KC;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int kc = min(KC, K - k0), kcp = (kc + 15) & ~15;             // four waves x whole MFMA steps
        const bool need_r = !(R_STAGED && one), need_l = !(L_STAGED && one);
        if (need_r || need_l) {
            BurstR<C> br;
            BurstL bl;
            if (need_r) br.template load<COHERENT>(R, k0, kc, kcp);
            if (need_l) bl.load(L, ldl, k0, kc, kcp, r0, n);
            __syncthreads();                           // the previous chunk is consumed
            if (need_r) br.store(Rc, kcp);
            if (need_l) bl.store(Lc, kcp);
            __syncthreads();
        }
        const int per = kcp >> 2, end = (wave + 1) * per;
        const float* lrow = Lc + i * LDL;
        for (int s = wave * per; s < end; s += 16) {                  // four MFMA steps' operands in flight
            float a[4], b[4][T];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = min(s + 4 * u, end - 4) + kq;           // (a step beyond the range re-reads the last one, times 0)
                const float* rrow = Rc + k * C;
                const float x = lrow[k];
                a[u] = s + 4 * u < end ? x : 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t) b[u][t] = rrow[(16 * t + i) ^ swz<C>(k)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][t], acc[t], 0, 0, 0);
        }
    }
    __syncthreads();                                   // every wave is done with the R chunk the partial tiles overlay
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) part[(wave * RB + 4 * kq + v) * LDT + 16 * t + i] = acc[t][v];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RB * C / NT; ++u) {
        const int e = tid + u * NT, r = e / C, c = e % C;
        dst[r * LDT + c] = part[r * LDT + c] + part[(RB + r) * LDT + c] + part[(2 * RB + r) * LDT + c] + part[(3 * RB + r) * LDT + c];
    }
    __syncthreads();
}

// what rides along in the forward launch besides the weight pack (all optional)
struct FrontJobs {
    mobgt_front::NodeIndexParams ni; int ni_on;       // index derivation of the node features (one block per graph)
    mobgt_front::HopFwd hf; int hf_blocks;            // hop table forward (blocks of 256 entries)
};

template <int H1, int H2, int H3>
__global__ __launch_bounds__(NT) void small_gcn_fwd_kernel(const SmallGcnParams p, const mobgt_pack::PackJobs jobs, int njobs, int nvb,
                                                          const FrontJobs fj) {
    if ((int)blockIdx.x >= p.nwg) {
        // passengers on the compute units this network leaves idle: the step's weight pack (mobgt_pack_mfma_b's body) and two tiny
        // front-of-step kernels (front_body.h).  (The bias ASSEMBLY was tried here too: with the pack in the same launch the
        // passengers outlast the network -- 36.2 us for 27.4 + 7.0, measured -- so it keeps its own launch.)
        const int np = (int)gridDim.x - p.nwg, pid = (int)blockIdx.x - p.nwg;
        // (the two tiny ones first, on the LAST passengers: the first ones carry the largest pack shares)
        const int rid = np - 1 - pid;
        if (fj.ni_on) {
            __shared__ int s_cnt[4];
            for (int g = rid; g < fj.ni.G; g += np) {
                mobgt_front::node_index_body(fj.ni, g, s_cnt);
                __syncthreads();
            }
        }
        for (int vb = rid; vb < fj.hf_blocks; vb += np) mobgt_front::hop_table_fwd_body(fj.hf, vb);
        for (int vb = pid; vb < nvb; vb += PACK_U * np) mobgt_pack::pack_blocks<PACK_U>(jobs, njobs, vb, np, nvb);
        return;
    }
    static_assert(H1 <= 32 && H1 * H2 + H2 * H3 <= MAXH * MAXH, "LDS plan");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tl = smem + OFF_T;
    float* bs = tl + RB * LDT;                    // b0 | b1 | b2 at MAXH strides
    float* w1s = smem + OFF_W;
    float* w1t = w1s + H1 * H2;
    float* w2s = w1t + H1 * H2;
    float* w2t = w2s + H2 * H3;
    float* Lc = smem + OFF_LC;
    const int r0 = blockIdx.x * RB, nwg = p.nwg;
    const int wave = threadIdx.x >> 6, i = threadIdx.x & 15, kq = (threadIdx.x & 63) >> 4;
    // rows of A early (they serve both adjacency products): then layer 1's own L chunk lives inside the R area
    const bool early = p.n <= KC && p.K0 <= KC;
    const int np = (p.n + 15) & ~15, k0p = (p.K0 + 15) & ~15;
    uint64_t seed = 0;
    STAMP_DECL;
    STAMP(0);
    {
        BurstW<H1, H2> w1;
        BurstW<H2, H3> w2;
        BurstR<H1> r;
        BurstL lx, la;
        if (p.thr) seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
        w1.load(p.W1); w2.load(p.W2);
        const int which = min((int)threadIdx.x / MAXH, 2), bc = threadIdx.x % MAXH;
        const float* bsrc = which == 0 ? p.b0 : (which == 1 ? p.b1 : p.b2);
        const float b = bsrc[min(bc, (which == 0 ? H1 : (which == 1 ? H2 : H3)) - 1)];
        if (early) {
            r.template load<false>(p.W0, 0, p.K0, k0p);
            lx.load(p.AX, p.K0, 0, p.K0, k0p, r0, p.n);
            la.load(p.A, p.n, 0, p.n, np, r0, p.n);
        }
        STAMP(10);
        w1.store(w1s, w1t); w2.store(w2s, w2t);
        STAMP(11);
        if (threadIdx.x < 3 * MAXH) bs[threadIdx.x] = b;
        if (early) {
            r.store(smem + OFF_RC, k0p);
            lx.store(smem + OFF_L2, k0p);
            la.store(Lc, np);
        }
        STAMP(12);
        __syncthreads();
    }
    STAMP(1);
    // layer 1 (the constant product A X comes in precomputed)
    if (early) tile_product<H1, true, true, false>(smem, smem + OFF_L2, p.AX, p.K0, p.W0, p.K0, r0, p.n, tl);
    else tile_product<H1, false, false, false>(smem, Lc, p.AX, p.K0, p.W0, p.K0, r0, p.n, tl);
    STAMP(2);
#pragma unroll
    for (int u = 0; u < RB * H1 / NT; ++u) {
        const int e = threadIdx.x + u * NT, r = e / H1, c = e % H1, row = r0 + r;
        if (row < p.n) coh_store(&p.h1[(int64_t)row * H1 + c], leaky(tl[r * LDT + c] + bs[c], p.slope));
    }
    STAMP(3);
    grid_barrier(p.counter, nwg);
    STAMP(4);
    // layer 2: t = A h1 (kept for the backward), h2 = dropout(leaky(t W1 + b1))
    tile_product<H1, true, false, true>(smem, Lc, p.A, p.n, p.h1, p.n, r0, p.n, tl);
    STAMP(5);
#pragma unroll
    for (int u = 0; u < RB * H1 / NT; ++u) {
        const int e = threadIdx.x + u * NT, r = e / H1, c = e % H1, row = r0 + r;
        if (row < p.n) p.t[(int64_t)row * H1 + c] = tl[r * LDT + c];
    }
    for (int t = wave; t < H2 / 16; t += 4) {
        const int c = 16 * t + i;
        const f32x4 u = mfma_tile<H1>([&](int k) { return tl[i * LDT + k]; }, [&](int k) { return w1s[k * H2 + c]; });
        const float b1 = bs[MAXH + c];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = r0 + 4 * kq + v;
            float a = leaky(u[v] + b1, p.slope);
            if (p.thr) {
                const uint32_t rowh = dropout_row_hash(seed, (uint32_t)row ^ p.salt);
                a = dropout_bits16(seed, rowh, (uint32_t)c) >= p.thr ? a * p.inv_keep : 0.f;
            }
            if (row < p.n) coh_store(&p.h2[(int64_t)row * H2 + c], a);
        }
    }
    STAMP(6);
    grid_barrier(p.counter, 2 * nwg);
    STAMP(7);
    // layer 3: t2 = A h2 (kept), out = t2 W2 + b2
    tile_product<H2, true, false, true>(smem, Lc, p.A, p.n, p.h2, p.n, r0, p.n, tl);
    STAMP(8);
#pragma unroll
    for (int u = 0; u < RB * H2 / NT; ++u) {
        const int e = threadIdx.x + u * NT, r = e / H2, c = e % H2, row = r0 + r;
        if (row < p.n) p.t2[(int64_t)row * H2 + c] = tl[r * LDT + c];
    }
    for (int t = wave; t < H3 / 16; t += 4) {
        const int c = 16 * t + i;
        const f32x4 o = mfma_tile<H2>([&](int k) { return tl[i * LDT + k]; }, [&](int k) { return w2s[k * H3 + c]; });
        const float b2 = bs[2 * MAXH + c];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = r0 + 4 * kq + v;
            if (row < p.n) p.out[(int64_t)row * H3 + c] = o[v] + b2;
        }
    }
    STAMP(9);
    STAMP_DUMP();
}

// dW[j][c] += sum_r S[r][j] G[r][c] (J x C outputs, j < jmax, from two LDS row tiles S [RB][lds] and G [RB][LDT]; rows
// beyond the workgroup's are zero in G);  MFMA tiles over (j, c), round-robin over the waves
template <int C>
__device__ __forceinline__ void tile_wgrad(const float* S, int lds, const float* G, int J, int jmax, float* dW) {
    const int wave = threadIdx.x >> 6, i = threadIdx.x & 15, kq = (threadIdx.x & 63) >> 4;
    constexpr int nt = C / 16;
    const int tiles = ((J + 15) >> 4) * nt;
    for (int t = wave; t < tiles; t += 4) {
        const int j0 = (t / nt) * 16, c0 = (t % nt) * 16;
        const f32x4 d = mfma_tile<RB>([&](int r) { return S[r * lds + j0 + i]; }, [&](int r) { return G[r * LDT + c0 + i]; });
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int j = j0 + 4 * kq + v;
            if (j < jmax) atomicAdd(&dW[(int64_t)j * C + c0 + i], d[v]);
        }
    }
}

template <int C>
__device__ __forceinline__ void tile_bgrad(const float* G, float* db) {
    if ((int)threadIdx.x < C) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < RB; ++r) acc += G[r * LDT + threadIdx.x];
        atomicAdd(&db[threadIdx.x], acc);
    }
}

template <int H1, int H2, int H3>
__global__ __launch_bounds__(NT) void small_gcn_bwd_kernel(const SmallGcnParams p, const BuildParams bp, int lds_rel, int lds_poi) {
    if ((int)blockIdx.x >= p.nwg) {
        // passengers: the backward of the bias tables (csrc/bias.hip, the short-batch form: int16 indices, uint8 edge ids, 8
        // heads, hop histogram on the matrix core, 4 waves) -- 21.6 us of its own launch, independent of this network
        build_bias_bwd_body<int16_t, uint8_t, 8, true, 4>(bp, lds_rel, lds_poi, (int)blockIdx.x - p.nwg, (int)gridDim.x - p.nwg);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* gl = smem + OFF_T;                     // this workgroup's rows of the current gradient
    float* t2l = gl + RB * LDT;                   // ... and of the saved activations, all fetched up front
    float* h2l = t2l + RB * LDT;
    float* tl = h2l + RB * LDT;
    float* h1l = tl + RB * LDT;
    float* w1s = smem + OFF_W;
    float* w1t = w1s + H1 * H2;
    float* w2s = w1t + H1 * H2;
    float* w2t = w2s + H2 * H3;
    float* Lc = smem + OFF_LC;
    const int r0 = blockIdx.x * RB, nwg = p.nwg;
    const int wave = threadIdx.x >> 6, i = threadIdx.x & 15, kq = (threadIdx.x & 63) >> 4;
    const bool early = p.n <= KC;                 // rows of A^T once, for both products
    const bool ax_early = p.K0 <= KC;             // rows of AX ride in registers from the first burst to the last layer
    uint64_t seed = 0;
    BurstL lax;
    STAMP_DECL;
    STAMP(0);
    {
        BurstRows<H3> g;
        BurstRows<H2> t2, h2;
        BurstRows<H1> t, h1;
        BurstW<H1, H2> w1;
        BurstW<H2, H3> w2;
        BurstL la;
        if (p.thr) seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
        g.load(p.g, r0, p.n); t2.load(p.t2, r0, p.n); h2.load(p.h2, r0, p.n);
        t.load(p.t, r0, p.n); h1.load(p.h1, r0, p.n);
        w1.load(p.W1); w2.load(p.W2);
        if (early) la.load(p.AT, p.n, 0, p.n, (p.n + 15) & ~15, r0, p.n);
        if (ax_early) lax.load(p.AX, p.K0, 0, p.K0, (p.K0 + 15) & ~15, r0, p.n);
        g.store(gl); t2.store(t2l); h2.store(h2l); t.store(tl); h1.store(h1l);
        w1.store(w1s, w1t); w2.store(w2s, w2t);
        if (early) la.store(Lc, (p.n + 15) & ~15);
        __syncthreads();
    }
    STAMP(1);
    // ---- layer 3: out = t2 W2 + b2
    tile_wgrad<H3>(t2l, LDT, gl, H2, H2, p.dW2);
    tile_bgrad<H3>(gl, p.db2);
    for (int t = wave; t < H2 / 16; t += 4) {                        // dt2 = g W2^T
        const int j = 16 * t + i;
        const f32x4 d = mfma_tile<H3>([&](int c) { return gl[i * LDT + c]; }, [&](int c) { return w2t[c * H2 + j]; });
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = r0 + 4 * kq + v;
            if (row < p.n) coh_store(&p.dt2[(int64_t)row * H2 + j], d[v]);
        }
    }
    STAMP(2);
    grid_barrier(p.counter, nwg);
    STAMP(3);
    // ---- layer 2: h2 = dropout(leaky(t W1 + b1)),  t2 = A h2  ->  dh2 = A^T dt2
    tile_product<H2, true, false, true>(smem, Lc, p.AT, p.n, p.dt2, p.n, r0, p.n, gl);
    STAMP(4);
#pragma unroll
    for (int u = 0; u < RB * H2 / NT; ++u) {
        const int e = threadIdx.x + u * NT, r = e / H2, c = e % H2, row = r0 + r;
        float keep = 1.f;
        if (p.thr) {
            const uint32_t rowh = dropout_row_hash(seed, (uint32_t)row ^ p.salt);
            keep = dropout_bits16(seed, rowh, (uint32_t)c) >= p.thr ? p.inv_keep : 0.f;
        }
        // as mobgt_bias_act_bwd (rows beyond n: the product's rows are zero there)
        gl[r * LDT + c] = gl[r * LDT + c] * keep * (h2l[r * LDT + c] > 0.f ? 1.f : p.slope);
    }
    __syncthreads();
    tile_wgrad<H2>(tl, LDT, gl, H1, H1, p.dW1);
    tile_bgrad<H2>(gl, p.db1);
    for (int t = wave; t < H1 / 16; t += 4) {                        // dt = du W1^T
        const int j = 16 * t + i;
        const f32x4 d = mfma_tile<H2>([&](int c) { return gl[i * LDT + c]; }, [&](int c) { return w1t[c * H1 + j]; });
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = r0 + 4 * kq + v;
            if (row < p.n) coh_store(&p.dt[(int64_t)row * H1 + j], d[v]);
        }
    }
    STAMP(5);
    grid_barrier(p.counter, 2 * nwg);
    STAMP(6);
    // ---- layer 1: h1 = leaky(AX W0 + b0),  t = A h1  ->  dh1 = A^T dt
    tile_product<H1, true, false, true>(smem, Lc, p.AT, p.n, p.dt, p.n, r0, p.n, gl);
    STAMP(7);
#pragma unroll
    for (int u = 0; u < RB * H1 / NT; ++u) {
        const int e = threadIdx.x + u * NT, r = e / H1, c = e % H1;
        gl[r * LDT + c] = gl[r * LDT + c] * (h1l[r * LDT + c] > 0.f ? 1.f : p.slope);
    }
    for (int k0 = 0; k0 < p.K0; k0 += KC) {                          // dW0[k][j] += sum_r AX[r][k] dpre[r][j]
        const int kc = min(KC, p.K0 - k0);
        if (!ax_early) lax.load(p.AX, p.K0, k0, kc, (kc + 15) & ~15, r0, p.n);
        __syncthreads();
        lax.store(Lc, (kc + 15) & ~15);
        __syncthreads();
        if (k0 == 0) tile_bgrad<H1>(gl, p.db0);
        tile_wgrad<H1>(Lc, LDL, gl, kc, kc, p.dW0 + (int64_t)k0 * H1);
    }
    STAMP(8);
    STAMP_DUMP();
}

int check(const SmallGcnParams& p) {
    if (p.n <= 0 || p.K0 <= 0 || p.H1 <= 0 || p.H2 <= 0 || p.H3 <= 0 || p.H1 > MAXH || p.H2 > MAXH || p.H3 > MAXH) return MOBGT_EBADDIM;
    if (p.H1 != 16 || p.H2 != 64 || p.H3 != 32) return MOBGT_EBADDIM;      // the instantiated widths (MobGT: gcn_nhid = [16, 64], 32 out)
    if (p.n > 4096) return MOBGT_EBADDIM;          // ceil(n/16) workgroups must be co-resident (one per CU: 148.5 KB of LDS each)
    return 0;
}

int lds_opt_in(const void* fn) {
    // more than the default 64 KB of dynamic LDS needs the attribute once per function (cheap; not a stream operation)
    return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * sizeof(float));
}

void set_drop(SmallGcnParams& p, float dropout_p, uint64_t seed, const uint64_t* seed_dev, uint32_t salt) {
    p.thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.thr ? 1.f / (1.f - (float)p.thr / 65536.f) : 1.f;
    p.seed = seed; p.seed_dev = seed_dev; p.salt = salt;
}

}  // namespace

extern "C" int mobgt_small_gcn_fwd_pack(const float* ax, const float* a, const float* w0, const float* b0, const float* w1,
                                        const float* b1, const float* w2, const float* b2, float* h1, float* t, float* h2, float* t2,
                                        float* out, int* counter, int n, int K0, int H1, int H2, int H3, float slope, float dropout_p,
                                        uint64_t seed, const uint64_t* seed_dev, uint32_t salt, int pack_n, const void* const* pack_src,
                                        void* const* pack_dst, const int* pack_N, const int* pack_K, const int* pack_transposed,
                                        // the arguments of mobgt_node_index (with_node_index != 0)
                                        int with_node_index, const void* ni_x, int ni_x_dtype, int64_t ni_xs_g, int64_t ni_xs_n,
                                        const float* ni_time_normal, int64_t ni_ts_g, int64_t ni_ts_n, const int64_t* ni_poi2cat,
                                        const void* ni_in_degree, const void* ni_out_degree, int ni_deg_dtype, int64_t* ni_idx,
                                        float* ni_real, int ni_G, int ni_N, int ni_rows_only,
                                        // the arguments of mobgt_hop_table_fwd (with_hop != 0)
                                        int with_hop, const float* hop_edge_encoder, const float* hop_edge_dis_encoder, float* hop_out,
                                        int hop_D, int hop_n_edge, int hop_H, int hop_fp16_roundtrip, void* stream) {
    SmallGcnParams p = {};
    p.AX = ax; p.A = a; p.W0 = w0; p.b0 = b0; p.W1 = w1; p.b1 = b1; p.W2 = w2; p.b2 = b2;
    p.h1 = h1; p.t = t; p.h2 = h2; p.t2 = t2; p.out = out; p.counter = counter;
    p.n = n; p.K0 = K0; p.H1 = H1; p.H2 = H2; p.H3 = H3; p.slope = slope;
    int rc = check(p);
    if (rc) return rc;
    set_drop(p, dropout_p, seed, seed_dev, salt);
    if (((uintptr_t)w0 | (uintptr_t)h1 | (uintptr_t)h2) & 15) return MOBGT_EALIGN;
    if ((rc = lds_opt_in((const void*)small_gcn_fwd_kernel<16, 64, 32>))) return rc;
    p.nwg = (n + RB - 1) / RB;
    static mobgt_pack::PackJobs jobs;            // (by value into the launch; 3 KB -- not on the stack of every call)
    int nvb = 0, passengers = 0;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int free_cus = cus > p.nwg ? cus - p.nwg : 1;      // (every workgroup of this launch owns a compute unit's LDS)
    if (pack_n > 0) {
        jobs = mobgt_pack::PackJobs{};
        if ((rc = mobgt_pack::fill_jobs(jobs, pack_n, pack_src, pack_dst, pack_N, pack_K, pack_transposed, &nvb))) return rc;
        passengers = PACK_ROUNDS * free_cus;
        if (passengers > (nvb + PACK_U - 1) / PACK_U) passengers = (nvb + PACK_U - 1) / PACK_U;
    }
    FrontJobs fj = {};
    if (with_node_index && ni_G > 0 && ni_N > 0) {
        if (ni_in_degree && ni_deg_dtype != MOBGT_I64 && ni_deg_dtype != MOBGT_I32 && ni_deg_dtype != MOBGT_I16) return MOBGT_EDTYPE;
        if (ni_x_dtype != MOBGT_I64 && ni_x_dtype != MOBGT_I32) return MOBGT_EDTYPE;
        fj.ni = mobgt_front::NodeIndexParams{ni_x, ni_x_dtype, ni_xs_g, ni_xs_n, ni_time_normal, ni_ts_g, ni_ts_n, ni_poi2cat, ni_in_degree,
                                             ni_out_degree, ni_deg_dtype, ni_idx, ni_real, ni_G, ni_N, ni_rows_only};
        fj.ni_on = 1;
        if (passengers < (ni_G < free_cus ? ni_G : free_cus)) passengers = ni_G < free_cus ? ni_G : free_cus;
    }
    if (with_hop) {
        if (hop_D <= 0 || hop_n_edge <= 0 || hop_H <= 0) return MOBGT_EBADDIM;
        fj.hf = mobgt_front::HopFwd{hop_edge_encoder, hop_edge_dis_encoder, hop_out, hop_D, hop_n_edge, hop_H, hop_fp16_roundtrip};
        fj.hf_blocks = (hop_D * hop_n_edge * hop_H + 255) / 256;
        const int want = fj.hf_blocks < free_cus ? fj.hf_blocks : free_cus;
        if (passengers < want) passengers = want;
    }
    static_assert(sizeof(SmallGcnParams) + sizeof(mobgt_pack::PackJobs) + sizeof(FrontJobs) + 16 <= 4096, "kernel arguments");
    hipLaunchKernelGGL((small_gcn_fwd_kernel<16, 64, 32>), dim3(p.nwg + passengers), dim3(NT), LDS_FLOATS * sizeof(float), (hipStream_t)stream, p,
                       jobs, pack_n > 0 ? pack_n : 0, nvb, fj);
    return (int)hipGetLastError();
}

extern "C" int mobgt_small_gcn_fwd(const float* ax, const float* a, const float* w0, const float* b0, const float* w1,
                                   const float* b1, const float* w2, const float* b2, float* h1, float* t, float* h2, float* t2,
                                   float* out, int* counter, int n, int K0, int H1, int H2, int H3, float slope, float dropout_p,
                                   uint64_t seed, const uint64_t* seed_dev, uint32_t salt, void* stream) {
    return mobgt_small_gcn_fwd_pack(ax, a, w0, b0, w1, b1, w2, b2, h1, t, h2, t2, out, counter, n, K0, H1, H2, H3, slope, dropout_p, seed,
                                    seed_dev, salt, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    0, nullptr, 0, 0, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0,
                                    0, nullptr, nullptr, nullptr, 0, 0, 0, 0, stream);
}

extern "C" int mobgt_small_gcn_bwd_bias(const float* g, const float* ax, const float* a_t, const float* w1, const float* w2,
                                        const float* h1, const float* t, const float* h2, const float* t2, float* dw0, float* db0,
                                        float* dw1, float* db1, float* dw2, float* db2, float* dt2, float* dt, int* counter, int n,
                                        int K0, int H1, int H2, int H3, float slope, float dropout_p, uint64_t seed,
                                        const uint64_t* seed_dev, uint32_t salt,
                                        // the arguments of mobgt_build_bias_bwd (with_bias != 0), idx int16 / edge uint8 / H = 8
                                        int with_bias, const void* dbias, int dbias_dtype, int n_slices, int64_t slice_stride,
                                        const float* attn_bias, const void* rel_pos, const void* poi_pos, const void* edge_input,
                                        float* d_rel_table, float* d_poi_table, float* d_hop_table, float* d_vdist, int G, int N,
                                        int H, int D_in, int D, int F, int n_rel, int n_poi, int n_edge, int64_t ld_bias,
                                        int idx_dtype, int edge_dtype, void* stream) {
    SmallGcnParams p = {};
    p.g = g; p.AX = ax; p.AT = a_t; p.W1 = w1; p.W2 = w2;
    p.h1 = const_cast<float*>(h1); p.t = const_cast<float*>(t); p.h2 = const_cast<float*>(h2); p.t2 = const_cast<float*>(t2);
    p.dW0 = dw0; p.db0 = db0; p.dW1 = dw1; p.db1 = db1; p.dW2 = dw2; p.db2 = db2; p.dt2 = dt2; p.dt = dt; p.counter = counter;
    p.n = n; p.K0 = K0; p.H1 = H1; p.H2 = H2; p.H3 = H3; p.slope = slope;
    int rc = check(p);
    if (rc) return rc;
    set_drop(p, dropout_p, seed, seed_dev, salt);
    if (((uintptr_t)dt2 | (uintptr_t)dt) & 15) return MOBGT_EALIGN;
    if ((rc = lds_opt_in((const void*)small_gcn_bwd_kernel<16, 64, 32>))) return rc;
    p.nwg = (n + RB - 1) / RB;
    BuildParams bp = {};
    int lds_rel = 0, lds_poi = 0, passengers = 0;
    if (with_bias) {
        if ((rc = fill_bias_bwd(bp, dbias, dbias_dtype, n_slices, slice_stride, attn_bias, rel_pos, poi_pos, edge_input, d_rel_table,
                                d_poi_table, d_hop_table, d_vdist, G, N, H, D_in, D, F, n_rel, n_poi, n_edge, ld_bias))) return rc;
        const int T = N + 1;
        // only the instantiation the short-batch launch of mobgt_build_bias_bwd would pick (see launch_build_bwd)
        if (idx_dtype != MOBGT_I16 || edge_dtype != MOBGT_U8 || H != 8 || !bp.edge_input || F != 1 || D > HOP_DMAX ||
            (int64_t)G * T * T >= (1 << 20)) return MOBGT_EBADDIM;
        lds_rel = bp.n_rel < 512 ? bp.n_rel : 512;
        lds_poi = bp.poi_pos ? (bp.n_poi < 1024 ? bp.n_poi : 1024) : 0;
        if ((size_t)bwd_lds_dwords(lds_rel, lds_poi, bp.D, bp.H) * sizeof(float) + bwd_stage_bytes<8, true, 4>() > LDS_FLOATS * sizeof(float))
            return MOBGT_EBADDIM;
        const int n_units = ((T + 63) / 64) * ((T + 3) / 4) * G;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        passengers = cus > p.nwg ? cus - p.nwg : 1;            // (every workgroup of this launch owns a compute unit's LDS)
        if (passengers > n_units) passengers = n_units;
    }
    hipLaunchKernelGGL((small_gcn_bwd_kernel<16, 64, 32>), dim3(p.nwg + passengers), dim3(NT), LDS_FLOATS * sizeof(float), (hipStream_t)stream,
                       p, bp, lds_rel, lds_poi);
    return (int)hipGetLastError();
}

extern "C" int mobgt_small_gcn_bwd(const float* g, const float* ax, const float* a_t, const float* w1, const float* w2,
                                   const float* h1, const float* t, const float* h2, const float* t2, float* dw0, float* db0,
                                   float* dw1, float* db1, float* dw2, float* db2, float* dt2, float* dt, int* counter, int n,
                                   int K0, int H1, int H2, int H3, float slope, float dropout_p, uint64_t seed,
                                   const uint64_t* seed_dev, uint32_t salt, void* stream) {
    return mobgt_small_gcn_bwd_bias(g, ax, a_t, w1, w2, h1, t, h2, t2, dw0, db0, dw1, db1, dw2, db2, dt2, dt, counter, n, K0, H1, H2, H3,
                                    slope, dropout_p, seed, seed_dev, salt, 0, nullptr, 0, 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, stream);
}

// Workgroups of the one-launch GCN kernels that gave up at a grid barrier since the last reset (`reset` != 0 clears the count).
// Synchronous (hipMemcpyFromSymbol): call it from the host outside any capture, after the work in question has been waited for.
extern "C" int mobgt_small_gcn_faults(int reset, uint32_t* count) {
    unsigned int v = 0;
    hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_sg_faults), sizeof(v));
    if (e != hipSuccess) return (int)e;
    if (count) *count = v;
    if (reset && v) {
        const unsigned int z = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_sg_faults), &z, sizeof(z));
    }
    return (int)e;
}
// Test hook: the barrier's give-up limit in ticks of the 100 MHz clock (<= 0: the default of 2 s).
extern "C" int mobgt_small_gcn_set_wait_limit(int64_t ticks_100mhz) {
    const long long v = ticks_100mhz > 0 ? ticks_100mhz : 200000000LL;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sg_wait_ticks), &v, sizeof(v));
}
