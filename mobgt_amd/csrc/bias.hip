// Attention-bias assembly for MobGT on gfx950: fused table gathers + multi-hop edge reduce
// (graphormer/model.py:126-190, model_fqandtoyo.py:1143-1216), its backward (scatter of dBias into the
// table gradients), and the re-layout of a caller-supplied bias.
//
// One workgroup owns a 32x32 tile of (query node i, key node j) pairs of one graph and produces all H
// heads of it.  The hop table (sum over h' of edge_encoder x edge_dis_encoder, [D, n_edge, H] f32) is
// read through L1/L2 (it is a few tens of KB); the per-pair inputs (D*F hop indices, rel_pos, poi_pos,
// attn_bias) are read once, coalesced along j; the outputs are written twice -- row-major for the
// forward / dQ pass and transposed (through an LDS tile) for the dK/dV pass -- both coalesced.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "mobgt_hip.h"
#include "sgemm_body.h"

namespace {

constexpr int TILE = 32;
constexpr int BIAS_MAXH = 32;

typedef float hb2_t __attribute__((ext_vector_type(2)));
typedef float hb4_t __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ T to_out(float v);
template <> __device__ __forceinline__ float to_out<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t to_out<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ float ld_as_float(const float* p) { return *p; }
__device__ __forceinline__ float ld_as_float(const bf16_t* p) { return (float)*p; }

template <typename TI> __device__ __forceinline__ int ld_idx(const void* base, int64_t off) {
    return (int)reinterpret_cast<const TI*>(base)[off];
}

// ------------------------------------------------------------------------------------ bias_pack
template <typename TS, typename TB>
__global__ __launch_bounds__(256) void bias_pack_kernel(const TS* __restrict__ src, int64_t s_g, int64_t s_h, int64_t s_i,
                                                       int64_t s_j, TB* __restrict__ bias, TB* __restrict__ bias_t,
                                                       int H, int T, int64_t ld) {
    __shared__ float tile[TILE][TILE + 1];
    const int gh = blockIdx.z, g = gh / H, h = gh % H;
    const int i0 = blockIdx.y * TILE, j0 = blockIdx.x * TILE;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const TS* s = src + g * s_g + h * s_h;
    TB* b = bias + (int64_t)gh * T * ld;
    TB* bt = bias_t ? bias_t + (int64_t)gh * T * ld : nullptr;
#pragma unroll
    for (int r = 0; r < TILE; r += 8) {
        const int i = i0 + ty + r, j = j0 + tx;
        float v = -INFINITY;
        if (i < T && j < T) v = ld_as_float(s + i * s_i + j * s_j);
        tile[ty + r][tx] = v;
        if (i < T && j < ld) b[(int64_t)i * ld + j] = to_out<TB>(v);
    }
    if (!bt) return;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TILE; r += 8) {
        const int j = j0 + ty + r, i = i0 + tx;                  // transposed walk: contiguous along i
        if (j < T && i < ld) bt[(int64_t)j * ld + i] = to_out<TB>(tile[tx][ty + r]);
    }
}

// ------------------------------------------------------------------------------------ build_bias
struct BuildParams {
    const float* attn_bias;
    const void *rel_pos, *poi_pos, *edge_input;
    const float *rel_table, *poi_table, *hop_table, *vdist;
    void *bias, *bias_t;
    const void* dbias;            // f32, or n_slices bf16 slices (one per layer) `slice_stride` elements apart
    int dbias_bf16, n_slices;
    int64_t slice_stride;
    float *d_rel, *d_poi, *d_hop, *d_vdist;
    int G, N, H, D_in, D, F, n_rel, n_poi, n_edge;
    int64_t ld;
};

// SPD divisor of model.py:158-163: pad(0) -> 1, k > 1 -> k-1, clamp to [0, D]
__device__ __forceinline__ float spd_divisor(int rp, int D) {
    int s = rp == 0 ? 1 : rp;
    s = s > 1 ? s - 1 : s;
    if (D > 0) s = s < 0 ? 0 : (s > D ? D : s);
    return (float)s;
}

// (bx, by, bz: the tile; `fill`: bring the tables into LDS first -- a persistent workgroup of the long-batch form does so once)
template <typename TI, typename TE, typename TB, int HH, int RND>
__device__ __forceinline__ void build_bias_body(const BuildParams& p, const int bx, const int by, const int bz, const bool fill = true) {
    // RND = 1: one workgroup = one 8-row round of a 32x32 tile (blockIdx.y = 4 * tile row + round): a short batch
    // (16 graphs x 41 tokens = 64 tiles) still spreads over 256 workgroups.  RND = 4 (long batches): the four rounds of a
    // tile one after the other, so that a row of the transposed copy receives 64 contiguous bytes instead of 16 (with
    // parts compiled out at c5: the 16-byte transposed stores were 132 of the kernel's 442 us, the hop-row gathers 223)
    // (long batches keep the tile in the output's own 2-byte type -- rounded once, as the direct copy is: a third less LDS, four
    //  workgroups per compute unit instead of three; row pitch 17 dwords: the transposed reads of the four 8-row pieces fall
    //  on disjoint banks)
    constexpr bool NARROW = RND == 4 && sizeof(TB) == 2;
    typedef typename std::conditional<NARROW, TB, float>::type TT;
    __shared__ TT tile[HH][8 * RND][TILE + (NARROW ? 2 : 1)];
    // long batches: the hop-table rows of the small edge ids (transition counts < 16: nearly all of them) wait in LDS --
    // the L1 path then carries only the index loads and the stores, and a hop row arrives in ~64 cycles instead of ~500
    constexpr int HOPL = 16;
    __shared__ __attribute__((aligned(16))) float hop_s[RND == 4 ? 20 : 1][RND == 4 ? HOPL : 1][8];
    const bool hop_lds = RND == 4 && HH == 8 && p.edge_input && p.D <= 20;
    // ... and so do the leading rows of the two position tables (shortest-path lengths and distance bins are small numbers): a
    // row costs one LDS round trip instead of a dependent gather through the L1 path (80 of the 229 us left without the hop sum)
    constexpr int TABL = 128;
    __shared__ __attribute__((aligned(16))) float tab_s[RND == 4 ? 2 : 1][RND == 4 ? TABL : 1][8];
    const bool tab_lds = RND == 4 && HH == 8;
    if (tab_lds && fill) {
        for (int e = threadIdx.x; e < 2 * TABL * 2; e += 256) {
            const int t = e / (TABL * 2), id = (e >> 1) % TABL, half = e & 1;
            const float* tab = t ? p.poi_table : p.rel_table;
            const int rows = t ? p.n_poi : p.n_rel;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tab && id < rows) v = *reinterpret_cast<const float4*>(tab + (int64_t)id * 8 + 4 * half);
            *reinterpret_cast<float4*>(&tab_s[RND == 4 ? t : 0][RND == 4 ? id : 0][4 * half]) = v;
        }
        if (!hop_lds) __syncthreads();
    }
    if (hop_lds && fill) {
        for (int e = threadIdx.x; e < 20 * HOPL * 2; e += 256) {
            const int d = e / (HOPL * 2), id = (e >> 1) % HOPL, half = e & 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (d < p.D && id < p.n_edge) v = *reinterpret_cast<const float4*>(p.hop_table + ((int64_t)d * p.n_edge + id) * 8 + 4 * half);
            *reinterpret_cast<float4*>(&hop_s[RND == 4 ? d : 0][RND == 4 ? id : 0][4 * half]) = v;
        }
        __syncthreads();
    }
    const int g = bz;
    const int N = p.N, T = N + 1;
    const int i0 = (RND == 1 ? (by >> 2) : by) * TILE, j0 = bx * TILE;    // token coordinates
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float inv_f = 1.f / (float)p.F;
    TB* B = reinterpret_cast<TB*>(p.bias);
    TB* BT = reinterpret_cast<TB*>(p.bias_t);
    // (long batches: the direct copy leaves through the tile as well -- 16-byte stores, four lanes per 64-byte row piece)
    const bool b_via_lds = NARROW && BT != nullptr;
    // what a pair reads from HBM, all of it requested at the top of its round (the edge ids used to be requested behind the
    // position rows: 212 -> 185 us at c5; requesting them a whole round ahead on top of that: no gain, 26 registers)
    const bool fast_edge = sizeof(TE) == 1 && HH == 8 && p.F == 1 && p.D <= 20 && (p.D_in & 3) == 0 && p.edge_input;
    struct PairIn { float ab; int rp, pp; uint32_t w[5]; };
    auto fetch = [&](const int rd, PairIn& q) {
        const int r = RND == 1 ? (by & 3) * 8 : rd * 8;
        const int ti = i0 + ty + r, tj = j0 + tx;
        q.ab = 0.f; q.rp = 0; q.pp = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) q.w[k] = 0u;
        if (ti < T && tj < T) {
            q.ab = p.attn_bias[((int64_t)g * T + ti) * T + tj];
            if (ti >= 1 && tj >= 1) {
                const int64_t pair = ((int64_t)g * N + (ti - 1)) * N + (tj - 1);
                q.rp = ld_idx<TI>(p.rel_pos, pair);
                if (p.poi_pos) q.pp = ld_idx<TI>(p.poi_pos, pair);
                if (fast_edge) {
                    const uint8_t* eb = reinterpret_cast<const uint8_t*>(p.edge_input) + pair * p.D_in;
#pragma unroll
                    for (int k = 0; k < 5; ++k)
                        if (4 * k < p.D) q.w[k] = *reinterpret_cast<const uint32_t*>(eb + 4 * k);
                }
            }
        }
    };
#pragma unroll 1
    for (int rd = 0; rd < RND; ++rd) {
        const int r = RND == 1 ? (by & 3) * 8 : rd * 8;
        const int ti = i0 + ty + r, tj = j0 + tx;                // token indices
        float acc[HH];
        bool live = ti < T && tj < T;
        PairIn in;
        fetch(rd, in);
        const float ab = in.ab;
#pragma unroll
        for (int h = 0; h < HH; ++h) acc[h] = live ? 2.f * ab : -INFINITY;     // attn_bias counted twice (model.py:127,190)
        if (live && ti >= 1 && tj == 0) {
#pragma unroll
            for (int h = 0; h < HH; ++h) acc[h] += p.vdist[h];                 // virtual-token column (model.py:139-151)
        }
        if (live && ti >= 1 && tj >= 1) {
            const int64_t pair = ((int64_t)g * N + (ti - 1)) * N + (tj - 1);
            const int rp = in.rp;
            const float* rrow = tab_lds && rp < TABL ? &tab_s[0][RND == 4 ? rp : 0][0] : p.rel_table + (int64_t)rp * HH;
#pragma unroll
            for (int h = 0; h < HH; ++h) acc[h] += rrow[h];
            if (p.poi_pos) {
                const int pp = in.pp;
                const float* prow = tab_lds && pp < TABL ? &tab_s[RND == 4 ? 1 : 0][RND == 4 ? pp : 0][0] : p.poi_table + (int64_t)pp * HH;
#pragma unroll
                for (int h = 0; h < HH; ++h) acc[h] += prow[h];
            }
            if (p.edge_input) {
                float e[HH];
#pragma unroll
                for (int h = 0; h < HH; ++h) e[h] = 0.f;
                const int64_t ebase = pair * p.D_in * p.F;
                if (fast_edge) {
                    // The MobGT case.  The plain loop below is D dependent round trips (id -> table row): the ids
                    // arrive as 5 dwords, then the rows are requested five hops at a time (same summation order).
                    uint32_t w[5];
#pragma unroll
                    for (int k = 0; k < 5; ++k) w[k] = in.w[k];
                    // (the sums run on pairs of heads: v_pk_add_f32, same order of additions per head)
                    hb2_t e2[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) e2[q] = hb2_t{0.f, 0.f};
                    if (hop_lds && __all(((w[0] | w[1] | w[2] | w[3] | w[4]) & 0xF0F0F0F0u) == 0u)) {
                        // every id of the wave's pairs is small: all 20 rows from LDS, no per-hop branch (rows of hops >= D are zero)
#pragma unroll
                        for (int d0 = 0; d0 < 20; d0 += 5) {
                            hb4_t lo[5], hi[5];
#pragma unroll
                            for (int j = 0; j < 5; ++j) {
                                const int d = d0 + j;
                                const int idx = (int)((w[d >> 2] >> (8 * (d & 3))) & 0xffu);
                                const hb4_t* hrow = reinterpret_cast<const hb4_t*>(&hop_s[RND == 4 ? d : 0][RND == 4 ? idx : 0][0]);
                                lo[j] = hrow[0];
                                hi[j] = hrow[1];
                            }
#pragma unroll
                            for (int j = 0; j < 5; ++j) {
                                e2[0] += __builtin_shufflevector(lo[j], lo[j], 0, 1);
                                e2[1] += __builtin_shufflevector(lo[j], lo[j], 2, 3);
                                e2[2] += __builtin_shufflevector(hi[j], hi[j], 0, 1);
                                e2[3] += __builtin_shufflevector(hi[j], hi[j], 2, 3);
                            }
                            __builtin_amdgcn_sched_barrier(0);       // (five hops' rows in flight, not twenty: registers)
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            e[HH > 2 * q ? 2 * q : 0] = e2[q].x;
                            e[HH > 2 * q + 1 ? 2 * q + 1 : 0] = e2[q].y;
                        }
                    } else {
#pragma unroll
                        for (int d0 = 0; d0 < 20; d0 += 5) {
                            float4 lo[5], hi[5];
#pragma unroll
                            for (int j = 0; j < 5; ++j) {
                                const int d = d0 + j;
                                const int idx = (int)((w[d >> 2] >> (8 * (d & 3))) & 0xffu);
                                if (hop_lds && idx < HOPL) {
                                    const float4* hrow = reinterpret_cast<const float4*>(&hop_s[RND == 4 ? d : 0][RND == 4 ? idx : 0][0]);
                                    lo[j] = hrow[0];
                                    hi[j] = hrow[1];
                                } else {
                                    const float4* hrow = reinterpret_cast<const float4*>(p.hop_table + ((int64_t)(d < p.D ? d : 0) * p.n_edge + idx) * 8);
                                    lo[j] = hrow[0];
                                    hi[j] = hrow[1];
                                }
                            }
#pragma unroll
                            for (int j = 0; j < 5; ++j) {
                                if (d0 + j < p.D) {
                                    e[0] += lo[j].x; e[1] += lo[j].y; e[2] += lo[j].z; e[3] += lo[j].w;
                                    e[HH > 4 ? 4 : 0] += hi[j].x; e[HH > 5 ? 5 : 0] += hi[j].y; e[HH > 6 ? 6 : 0] += hi[j].z; e[HH > 7 ? 7 : 0] += hi[j].w;
                                }
                            }
                        }
                    }
                } else
                for (int d = 0; d < p.D; ++d) {
                    for (int f = 0; f < p.F; ++f) {
                        const int idx = ld_idx<TE>(p.edge_input, ebase + (int64_t)d * p.F + f);
                        const float* hrow = p.hop_table + ((int64_t)d * p.n_edge + idx) * HH;
#pragma unroll
                        for (int h = 0; h < HH; ++h) e[h] += hrow[h];
                    }
                }
                const float inv = inv_f / spd_divisor(rp, p.D);
#pragma unroll
                for (int h = 0; h < HH; ++h) acc[h] += e[h] * inv;
            }
        }
#pragma unroll
        for (int h = 0; h < HH; ++h) {
            tile[h][RND == 1 ? ty : ty + r][tx] = (TT)acc[h];
            if (ti < T && tj < p.ld && !b_via_lds) B[(((int64_t)g * HH + h) * T + ti) * p.ld + tj] = to_out<TB>(acc[h]);
        }
    }
    if (!BT) return;
    __syncthreads();
    if (b_via_lds) {
        for (int e = threadIdx.x; e < HH * TILE * 4; e += 256) {
            const int c8 = e & 3, row = (e >> 2) % TILE, h = e / (4 * TILE);
            const int ti = i0 + row;
            if (ti >= T) continue;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&tile[h][row][c8 * 8]);
            *reinterpret_cast<uint4*>(B + (((int64_t)g * HH + h) * T + ti) * p.ld + j0 + c8 * 8) = make_uint4(src[0], src[1], src[2], src[3]);
        }
    }
    // transposed copy: row tj of bias_t receives the 8 * RND consecutive queries of this workgroup, 8 per thread
    for (int e = threadIdx.x; e < HH * TILE * RND; e += 256) {
        const int piece = e % RND, c = (e / RND) % TILE, h = e / (RND * TILE);
        const int tj = j0 + c;
        if (tj >= T) continue;
        const int r = RND == 1 ? (by & 3) * 8 : 8 * piece;
        // ld is a multiple of 32 and i0 + r of 8: the 8 elements are in range and 16-byte aligned -> one (bf16) or
        // two (f32) 16-byte stores
        TB* dst = BT + (((int64_t)g * HH + h) * T + tj) * p.ld + i0 + r;
        float v8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v8[q] = (float)tile[h][(RND == 1 ? 0 : r) + q][c];
        store8(dst, v8);
    }
}

// --------------------------------------------------------------------------------- build_bias_bwd
// dbias[g,h,i,j] (f32, summed over layers) -> d_rel_table[rel_pos], d_poi_table[poi_pos],
// d_hop_table[d, edge_input[d], :] (scaled by 1/(F*spd)), d_vdist.
//
// A scatter-add whose keys are massively repeated (most pairs of a trajectory graph are "unreachable",
// padding or "no hop"), so plain atomics serialise on a handful of addresses.  Each wave therefore first
// combines lanes that share a key: leader's key -> ballot of equal keys -> butterfly reduce-scatter of the
// H head values over the 64 lanes (H-1 + log2(64/H) shuffles) -> H lanes issue one atomic each.  The
// per-workgroup tables live in LDS (rel, poi, and hop rows with a small edge id); the workgroup flushes
// its non-zero entries with global atomics once.
// atomic add on a pointer KNOWN to be LDS: a generic float* here compiles to flat_atomic_add_f32
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ void lds_add(float* p, float v) {
    __hip_atomic_fetch_add((lds_float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int HH>
__device__ __forceinline__ void wave_reduce_heads(float (&v)[HH], int lane) {
    // after this, lane l with (l & (64/HH - 1)) == 0 holds in v[0] the wave-wide sum of head l / (64/HH)
    int w = HH / 2;
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
#pragma unroll
    for (int bit = 64 / HH / 2; bit >= 1; bit >>= 1) v[0] += __shfl_xor(v[0], bit, 64);
}

// all 64 lanes must call; key < 0 = lane has nothing to add.  table row r, head h at table[r*HH + h].
// Adaptive: a key shared by >= 8 lanes of the wave ("unreachable", padding, "no hop", ...) is combined
// in registers and costs H atomics; keys held by only a few lanes (the distinct SPDs / distance bins along
// a trajectory) go straight to the atomic unit, where they do not collide anyway.
constexpr int MAX_LIGHT = 2;

// The LDS table is laid out [head][row] with an ODD row stride `lo_stride`: one instruction adds one head of up to 64
// different rows -- distinct banks unless rows collide mod 32 -- and the H lanes of a combined key hit H different banks.
// ([row][head] put the 64 lanes of an instruction on 4 banks: the LDS was busy half of the kernel's time.)
template <int HH>
__device__ __forceinline__ void wave_scatter_add(float* table_lo, int lo_rows, int lo_stride, float* table_hi, int key,
                                                 const float (&vals)[HH], int lane) {
    unsigned long long todo = __ballot(key >= 0);
    unsigned long long light = 0;
    // probing costs ~25 instructions per distinct key: stop after MAX_LIGHT keys that turned out to be rare --
    // whatever is still unprobed then goes to the atomic unit directly, like the rare keys themselves
    int misses = 0;
    for (int it = 0; it < 8 && todo && misses < MAX_LIGHT; ++it) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader, 64);
        const bool mine = key == k;
        const unsigned long long same = __ballot(mine) & todo;
        todo &= ~same;
        if (__popcll(same) < 8) { light |= same; ++misses; continue; }
        float r[HH];
#pragma unroll
        for (int h = 0; h < HH; ++h) r[h] = mine ? vals[h] : 0.f;
        wave_reduce_heads<HH>(r, lane);
        constexpr int STEP = 64 / HH;
        if ((lane & (STEP - 1)) == 0) {
            const int h = lane / STEP;
            // two explicit paths: a select of an LDS and a global pointer compiles to flat_atomic_add_f32
            if (k < lo_rows) lds_add(table_lo + h * lo_stride + k, r[0]);
            else atomicAdd(table_hi + (size_t)k * HH + h, r[0]);
        }
    }
    light |= todo;
    if ((light >> lane) & 1ull) {
        if (key < lo_rows) {
#pragma unroll
            for (int h = 0; h < HH; ++h) lds_add(table_lo + h * lo_stride + key, vals[h]);
        } else {
#pragma unroll
            for (int h = 0; h < HH; ++h) atomicAdd(table_hi + (size_t)key * HH + h, vals[h]);
        }
    }
}

// LDS FLOAT atomics are serial on gfx950: ds_add_f32 costs ~3 cycles per ACTIVE LANE (193 cycles per full wave instruction
// whatever the addresses) against 8 cycles for ds_add_u32 and 13 for ds_add_u64 (tools/micro/lds_rate.hip).  The rel / poi
// tables receive one add per lane, head and round (distinct SPDs / distance bins along a row: nothing to combine), and with
// float atomics the LDS was busy two thirds of the kernel's time.  They are therefore kept in 64-bit FIXED POINT: values
// scaled by 2^k with k chosen per launch from a strided sample of the gradient (sample maximum -> 2^34), added with
// ds_add_u64, converted back at the flush.  An addend is exact unless it is below 2^-10 of the sample maximum, then it is
// rounded to 2^-34 of it; one that is 2^12 times LARGER than the sample maximum (or not finite) does not fit the headroom
// left for 2^16 adds per entry and goes straight to the global f32 table, as do all of them when the sample is all zero.
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
constexpr float FX_LIMIT = 0x1p46f;
__device__ __forceinline__ unsigned long long to_fx(float x, float scale, bool& ok) {
    const float y = rintf(x * scale);
    ok = fabsf(y) < FX_LIMIT;                              // false for NaN / inf, and for every x under a NaN scale
    const float hi = floorf(y * 0x1p-32f);
    const float lo = fmaf(hi, -0x1p32f, y);                // in [0, 2^32), an integer
    return ((unsigned long long)(uint32_t)(int32_t)hi << 32) | (unsigned long long)(uint32_t)lo;
}
__device__ __forceinline__ void lds_add_fx(unsigned long long* p, unsigned long long v) {
    __hip_atomic_fetch_add((lds_u64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// wave_scatter_add for a fixed-point LDS table: fx[h] = to_fx(vals[h]), fx_ok = all of them fit
template <int HH>
__device__ __forceinline__ void wave_scatter_add_fx(unsigned long long* table_lo, int lo_rows, int lo_stride, float* table_hi,
                                                    int key, const float (&vals)[HH], const unsigned long long (&fx)[HH],
                                                    bool fx_ok, float scale, int lane) {
    unsigned long long todo = __ballot(key >= 0);
    unsigned long long light = 0;
    int misses = 0;
    for (int it = 0; it < 8 && todo && misses < MAX_LIGHT; ++it) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader, 64);
        const bool mine = key == k;
        const unsigned long long same = __ballot(mine) & todo;
        todo &= ~same;
        if (__popcll(same) < 8) { light |= same; ++misses; continue; }
        float r[HH];
#pragma unroll
        for (int h = 0; h < HH; ++h) r[h] = mine ? vals[h] : 0.f;
        wave_reduce_heads<HH>(r, lane);
        constexpr int STEP = 64 / HH;
        bool ok;
        const unsigned long long q = to_fx(r[0], scale, ok);
        if ((lane & (STEP - 1)) == 0) {
            const int h = lane / STEP;
            if (k < lo_rows && ok) lds_add_fx(table_lo + h * lo_stride + k, q);
            else atomicAdd(table_hi + (size_t)k * HH + h, r[0]);
        }
    }
    light |= todo;
    if ((light >> lane) & 1ull) {
        if (key < lo_rows && fx_ok) {
#pragma unroll
            for (int h = 0; h < HH; ++h) lds_add_fx(table_lo + h * lo_stride + key, fx[h]);
        } else {
#pragma unroll
            for (int h = 0; h < HH; ++h) atomicAdd(table_hi + (size_t)key * HH + h, vals[h]);
        }
    }
}

constexpr int HOP_LDS_ROWS = 16;      // edge ids < 16 (count <= 12) accumulate in LDS; rarer ids go to global

// Hop-table gradient on the matrix core (HOPMM: F == 1, 8 heads, D <= HOP_DMAX).  d_hop[d, e, h] is a histogram
// of 8-vectors keyed by the edge id of hop d: per hop slot, onehot(e)^T [16 ids x pairs] times ge [pairs x 8].
// One v_mfma_f32_16x16x32_bf16 contracts 32 pairs: the A operand is built from 8 staged edge bytes per lane
// (byte == my row ? 1 : 0), the B operand holds ge split into bf16 hi (columns 0-7) and lo (columns 8-15) parts,
// so the sum is exact to ~2^-17, and the [16 x 16] result of every hop slot stays in 4 registers per lane for the
// whole tile.  This replaces 20 ballot/shuffle key-combining rounds per 64 pairs (~5000 VALU instructions; the
// kernel ran at 54 % VALU utilisation with 68 % of wave time parked) by ~1300 + 40 MFMAs.  Edge ids >= 16 (a
// transition seen > 14 times in one trajectory) go to the atomic unit directly.
constexpr int HOP_DMAX = 20;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int HOP_STRIDE = HOP_LDS_ROWS + 1;
__host__ __device__ constexpr int odd(int n) { return n | 1; }
// LDS dwords of the per-workgroup tables: rel, poi (64-bit fixed point), vdist, hop rows, hop-length sums (f32) -- each
// [head][odd stride]
__host__ __device__ inline int bwd_lds_dwords(int lds_rel, int lds_poi, int D, int H) {
    return (2 * (odd(lds_rel) + odd(lds_poi)) + 1 + D * HOP_STRIDE + odd(D + 1)) * H;
}

// NW waves per workgroup, each wave owning ONE row x 64 columns of a unit (its row segments of the bf16 gradient slices are
// whole 128-byte lines: with 2 rows x 32 columns every line was fetched twice, by two workgroups): 4 waves for short
// batches (more units to spread), 8 for long ones (one workgroup per CU shares ONE set of tables among 8 waves: the
// 4-wave form needed 78 KB of LDS and 308 registers, i.e. ran one wave per SIMD with every latency exposed).
// (round 4) A tall-and-narrow small GEMM left by mobgt_front_sgemm_job rides in the SHORT-batch launch as extra z-slices of the
// grid: the distance GCN's first layer (parameters only) is as independent of the bias assembly as two launches can be, and
// both are a few hundred 256-thread workgroups.
struct SgemmRide {
    mobgt_sgemm::SgemmParams sg;
    int tiles, z0;                           // 16-row tiles; first z-slice that belongs to them
};
SgemmRide g_front_sg = {};                   // host side, one thread: the job the next short-batch launch takes along

template <typename TI, typename TE, typename TB, int HH, int RND>
__global__ __launch_bounds__(256) void build_bias_kernel(const BuildParams p, const SgemmRide ride) {
    if (RND == 1 && ride.tiles > 0 && (int)blockIdx.z >= ride.z0) {
        __shared__ float sg_part[4][16 * 17];
        const int tile = (((int)blockIdx.z - ride.z0) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
        if (tile < ride.tiles) mobgt_sgemm::sgemm_splitk_tile<4>(ride.sg, tile, sg_part);
        return;
    }
    if (RND == 4) {
        // long batches, a 1-D launch: the four tiles of a 2 x 2 group -- which share every 128-byte line of the two copies, 64
        // bytes each -- run on ONE XCD in adjacent dispatch slots (workgroups go round-robin over the 8 XCDs), so that its L2
        // merges the halves before the line is written back
        // ... and the workgroups are persistent (gridDim.x = a multiple of 32): the tables are brought into LDS once per
        // workgroup, not once per tile (18 KB x 10.8 k tiles at c5)
        const int nt = (int)(p.ld / TILE), nt2 = (nt + 1) / 2;
        const int lid = (int)blockIdx.x;
        const int xcd = lid & 7, slot = lid >> 3, sub = slot & 3;
        const int ngrp = nt2 * nt2 * p.G, nq = (int)gridDim.x >> 2;
        bool fill = true;
        for (int grp = (slot >> 2) * 8 + xcd; grp < ngrp; grp += nq) {
            const int bx = 2 * (grp % nt2) + (sub & 1), by = 2 * ((grp / nt2) % nt2) + (sub >> 1), bz = grp / (nt2 * nt2);
            if (bx < nt && by < nt) {                       // (workgroup-uniform)
                build_bias_body<TI, TE, TB, HH, RND>(p, bx, by, bz, fill);
                fill = false;
                __syncthreads();                            // (the tile is reused)
            }
        }
        return;
    }
    build_bias_body<TI, TE, TB, HH, RND>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

template <bool HOPMM, int NW> __host__ __device__ constexpr size_t bwd_hop_e_bytes() {
    return (sizeof(uint16_t) * (HOPMM ? NW : 1) * (HOPMM ? HOP_DMAX : 1) * 64 + 15) / 16 * 16;
}
// bytes of per-wave staging in front of the tables: hop_e | gr_s | wmax_s, a multiple of 16
template <int HH, bool HOPMM, int NW> __host__ __device__ constexpr size_t bwd_stage_bytes() {
    return bwd_hop_e_bytes<HOPMM, NW>() + sizeof(float) * NW * HH * 64 + (sizeof(float) * NW + 15) / 16 * 16;
}
// (`bid` of `nblk`: the workgroup's place among those that run this body -- the grid of build_bias_bwd_kernel, or the passenger
//  workgroups of the category GCN's backward launch, csrc/smallgcn.hip)
template <typename TI, typename TE, int HH, bool HOPMM, int NW>
__device__ __forceinline__ void build_bias_bwd_body(const BuildParams& p, int lds_rel, int lds_poi, const int bid, const int nblk) {
    // per-wave staging in front of the tables, all of it in the launch's DYNAMIC LDS (bwd_stage_bytes): as a passenger of the
    // category GCN's backward launch (csrc/smallgcn.hip) this body shares that launch's 148.5 KB, and static arrays would come
    // on top of them
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    typedef uint16_t (*hop_e_t)[HOPMM ? HOP_DMAX : 1][64];
    typedef float (*gr_s_t)[HH][64];
    hop_e_t hop_e = reinterpret_cast<hop_e_t>(smem_all);                                                // [wave][d][pair] one-hot
    // (the MFMA's B staging [half][col][pair] bf16 overlays the wave's gr_s tile: both are private to the wave, LDS runs a
    // wave's instructions in order, and gr_s is dead once gr[] has been read)
    gr_s_t gr_s = reinterpret_cast<gr_s_t>(reinterpret_cast<unsigned char*>(smem_all) + bwd_hop_e_bytes<HOPMM, NW>());    // [wave][head][col]
    static_assert(!HOPMM || sizeof(float) * HH * 64 >= sizeof(bf16_t) * 2 * 16 * 32, "hop_b overlays gr_s");
    float* wmax_s = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(smem_all) + bwd_hop_e_bytes<HOPMM, NW>() + sizeof(float) * NW * HH * 64);
    float* smem = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(smem_all) + bwd_stage_bytes<HH, HOPMM, NW>());
    f32x4_t hacc[HOPMM ? HOP_DMAX : 1];
#pragma unroll
    for (int d = 0; d < (HOPMM ? HOP_DMAX : 1); ++d) hacc[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int st_rel = odd(lds_rel), st_poi = odd(lds_poi), st_len = odd(p.D + 1);
    unsigned long long* s_rel = reinterpret_cast<unsigned long long*>(smem);      // [HH][st_rel] fixed point
    unsigned long long* s_poi = s_rel + (size_t)st_rel * HH;                      // [HH][st_poi] fixed point
    float* s_vd = reinterpret_cast<float*>(s_poi + (size_t)st_poi * HH);          // [HH]
    float* s_hop = s_vd + HH;                             // [D][HH][HOP_STRIDE]
    float* s_len = s_hop + (size_t)p.D * HOP_STRIDE * HH;     // [HH][st_len]: sums keyed by the number of real hops
    const int n_lds = bwd_lds_dwords(lds_rel, lds_poi, p.D, HH);
    for (int t = threadIdx.x; t < n_lds; t += blockDim.x) smem[t] = 0.f;

    const int N = p.N, T = N + 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float inv_f = 1.f / (float)p.F;
    // fixed-point scale from a strided sample of the gradient (the same 16 positions x threads in every workgroup: L2 hits
    // after the first; bf16: first and last layer slice, times the number of slices)
    // (short batches -- one unit per workgroup -- keep f32 LDS tables in the same memory: their few hundred adds do not
    // repay the sample's extra memory round trip and barrier: 25 vs 22 us at c2)
    constexpr bool FX = NW == 8;
    float* f_rel = reinterpret_cast<float*>(s_rel);       // !FX: [HH][st_rel] f32
    float* f_poi = f_rel + (size_t)st_rel * HH;           // !FX: [HH][st_poi] f32
    float fx_scale = __builtin_nanf(""), fx_inv = 0.f;     // NaN scale = no usable sample: to_fx says 'does not fit' for every addend
    if constexpr (!FX) __syncthreads();
    if constexpr (FX) {
        const int64_t total = (int64_t)p.G * HH * T * p.ld;
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int64_t at = (((int64_t)(threadIdx.x * 4 + q) * total) / (4 * NW * 64)) & ~(int64_t)7;
            if (at + 8 > total) at = (total - 8) & ~(int64_t)7;       // tiny batches: stay inside the buffer
            if (at < 0) continue;
            const int col = (int)(at % p.ld);              // (columns >= T are padding nobody has to have written)
            if (p.dbias_bf16) {
                const bf16_t* src = reinterpret_cast<const bf16_t*>(p.dbias) + at;
                const uint4 a = *reinterpret_cast<const uint4*>(src);
                const uint4 b = *reinterpret_cast<const uint4*>(src + (int64_t)(p.n_slices - 1) * p.slice_stride);
                const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (col + 2 * (i & 3) < T) m = fmaxf(m, fabsf(bf16_lo(w[i])));
                    if (col + 2 * (i & 3) + 1 < T) m = fmaxf(m, fabsf(bf16_hi(w[i])));
                }
            } else {
                const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.dbias) + at);
                const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.dbias) + at + 4);
                const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (col + i < T) m = fmaxf(m, fabsf(v[i]));
            }
        }
#pragma unroll
        for (int bit = 32; bit >= 1; bit >>= 1) m = fmaxf(m, __shfl_xor(m, bit, 64));
        if (lane == 0) wmax_s[wv] = m;
        __syncthreads();
        m = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) m = fmaxf(m, wmax_s[w]);
        if (p.dbias_bf16) m *= (float)p.n_slices;
        if (m > 0.f && m < INFINITY) {                     // (NaN fails both)
            int e = 34 - ilogbf(m);
            e = e > 120 ? 120 : (e < -120 ? -120 : e);
            fx_scale = ldexpf(1.f, e);
            fx_inv = ldexpf(1.f, -e);
        }
    }
    // (the barrier above also orders the zeroing of the tables before their first use)
    // A workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and flushes its LDS tables ONCE at the end:
    // with one tile per workgroup a long-trajectory batch (T = 785: 10 000 tiles) flushed ~100 M f32 atomics
    // (every tile touches most SPD / distance-bin rows) and spent ~3 ms doing so.
    // the unit of work is NW rows x 64 columns; consecutive units are neighbours along the row
    const int njt = (T + 63) / 64, nib = (T + NW - 1) / NW;
    const int n_units = njt * nib * p.G;
#pragma unroll 1
    for (int unit = bid; unit < n_units; unit += nblk) {
    const int g = unit / (njt * nib);
    const int j0 = (unit % njt) * 64;
    {
        const int ti = ((unit / njt) % nib) * NW + wv, tj = j0 + lane;
        // Every load of this round is issued up front, keyed on the index range only: the chain
        // attn_bias -> (live?) -> dBias -> rel_pos -> hop ids used to be 4-5 dependent memory round trips per round
        // at 3 waves per SIMD (~13 k cycles per round).
        const bool inr = ti >= 1 && ti < T && tj < T;
        const bool pairin = inr && tj >= 1;
        const int64_t pair = pairin ? ((int64_t)g * N + (ti - 1)) * N + (tj - 1) : 0;
        const float ab = inr ? p.attn_bias[((int64_t)g * T + ti) * T + tj] : -INFINITY;
        const int rp_raw = pairin ? ld_idx<TI>(p.rel_pos, pair) : 0;
        const int pp_raw = (pairin && p.poi_pos) ? ld_idx<TI>(p.poi_pos, pair) : 0;
        const int64_t ebase = pair * p.D_in * p.F;
        // a pair's hop ids: D_in bytes (4-byte aligned when D_in % 4 == 0) as dwords -- 5 loads instead of 20
        constexpr bool HOPW = HOPMM && sizeof(TE) == 1;
        uint32_t hop_w[HOPW ? HOP_DMAX / 4 : 1];
        int hop_id[(HOPMM && !HOPW) ? HOP_DMAX : 1];
        const bool hop_words = HOPW && (p.D_in & 3) == 0;
        if (HOPMM) {
            if (HOPW && hop_words) {
#pragma unroll
                for (int w = 0; w < HOP_DMAX / 4; ++w)
                    hop_w[w] = (4 * w < p.D && pairin)
                                   ? *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(p.edge_input) + ebase + 4 * w)
                                   : 0u;
            } else if (HOPW) {
#pragma unroll
                for (int w = 0; w < HOP_DMAX / 4; ++w) {
                    uint32_t v = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (4 * w + b < p.D && pairin) v |= (uint32_t)ld_idx<TE>(p.edge_input, ebase + 4 * w + b) << (8 * b);
                    hop_w[w] = v;
                }
            } else {
#pragma unroll
                for (int d = 0; d < HOP_DMAX; ++d) hop_id[d] = (d < p.D && pairin) ? ld_idx<TE>(p.edge_input, ebase + d) : 0;
            }
        }
        float gr[HH];
#pragma unroll
        for (int h = 0; h < HH; ++h) gr[h] = 0.f;
        if (p.dbias_bf16) {
            // bf16 layer slices: the wave's 64 pairs are 1 row x 64 columns x HH heads = HH row segments of 128 B per
            // slice -- ONE 16-byte-per-lane load instruction per slice (lane = (head, 8-column part)) instead of
            // HH two-byte loads per pair, and up to 12 slices in flight at once.  The f32 sums go through a
            // wave-private LDS tile to the lanes that own the pairs.
            const int sh = lane >> 3, part = lane & 7;
            float s8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s8[i] = 0.f;
            if (sh < HH && ti >= 1 && ti < T && j0 + 8 * part < p.ld) {
                const bf16_t* src = reinterpret_cast<const bf16_t*>(p.dbias) + (((int64_t)g * HH + sh) * T + ti) * p.ld + j0 + 8 * part;
                for (int l = 0; l < p.n_slices; l += 12) {
                    uint4 t[12];
#pragma unroll
                    for (int u = 0; u < 12; ++u)
                        t[u] = l + u < p.n_slices ? *reinterpret_cast<const uint4*>(src + (l + u) * p.slice_stride) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                    for (int u = 0; u < 12; ++u) {
                        s8[0] += bf16_lo(t[u].x); s8[1] += bf16_hi(t[u].x); s8[2] += bf16_lo(t[u].y); s8[3] += bf16_hi(t[u].y);
                        s8[4] += bf16_lo(t[u].z); s8[5] += bf16_hi(t[u].z); s8[6] += bf16_lo(t[u].w); s8[7] += bf16_hi(t[u].w);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (sh < HH) {
                float* dst = &gr_s[wv][sh][8 * part];
                *reinterpret_cast<float4*>(dst) = make_float4(s8[0], s8[1], s8[2], s8[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(s8[4], s8[5], s8[6], s8[7]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < HH; ++h) gr[h] = gr_s[wv][h][lane];
        } else if (inr) {
            const int64_t at0 = ((int64_t)g * HH * T + ti) * p.ld + tj;          // head 0; heads are T * ld apart
            const int64_t hs = (int64_t)T * p.ld;
#pragma unroll
            for (int h = 0; h < HH; ++h) gr[h] = reinterpret_cast<const float*>(p.dbias)[at0 + h * hs];
        }
        const bool live = inr && ab != -INFINITY;                 // -inf: probability 0, no gradient
        if (!live) {
#pragma unroll
            for (int h = 0; h < HH; ++h) gr[h] = 0.f;
        }
        // virtual-token column
        wave_scatter_add<HH>(s_vd, 1, 1, s_vd, (live && tj == 0) ? 0 : -1, gr, lane);
        const bool pairlive = live && tj >= 1;
        const int rp = pairlive ? rp_raw : 0;
        // row 0 of the index tables is nn.Embedding's padding_idx: it never receives a gradient, skip it
        if constexpr (FX) {
            unsigned long long gfx[HH];
            bool gfx_ok = true;
#pragma unroll
            for (int h = 0; h < HH; ++h) {
                bool ok;
                gfx[h] = to_fx(gr[h], fx_scale, ok);
                gfx_ok = gfx_ok && ok;
            }
            wave_scatter_add_fx<HH>(s_rel, lds_rel, st_rel, p.d_rel, (pairlive && rp != 0) ? rp : -1, gr, gfx, gfx_ok, fx_scale, lane);
            if (p.poi_pos) {
                const int pp = pairlive ? pp_raw : 0;
                wave_scatter_add_fx<HH>(s_poi, lds_poi, st_poi, p.d_poi, (pairlive && pp != 0) ? pp : -1, gr, gfx, gfx_ok, fx_scale, lane);
            }
        } else {
            wave_scatter_add<HH>(f_rel, lds_rel, st_rel, p.d_rel, (pairlive && rp != 0) ? rp : -1, gr, lane);
            if (p.poi_pos) {
                const int pp = pairlive ? pp_raw : 0;
                wave_scatter_add<HH>(f_poi, lds_poi, st_poi, p.d_poi, (pairlive && pp != 0) ? pp : -1, gr, lane);
            }
        }
        if (p.edge_input) {
            const float inv = inv_f / spd_divisor(rp, p.D);
            float ge[HH];
#pragma unroll
            for (int h = 0; h < HH; ++h) ge[h] = pairlive ? gr[h] * inv : 0.f;      // (the virtual-token column has no hops)
            if (HOPMM) {
                const int wave = threadIdx.x >> 6, half = lane >> 5, k = lane & 31;
                bf16_t (*hop_b)[16][32] = reinterpret_cast<bf16_t (*)[16][32]>(&gr_s[wave][0][0]);       // [half][col][pair]
                // the staging areas are private to the wave and LDS executes a wave's instructions in order, so only
                // the COMPILER has to be kept from moving this round's writes above the previous round's reads
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int h = 0; h < HH; ++h) {
                    const bf16_t hi = (bf16_t)ge[h];
                    hop_b[half][h][k] = hi;
                    hop_b[half][8 + h][k] = (bf16_t)(ge[h] - (float)hi);
                }
#pragma unroll
                for (int d = 0; d < HOP_DMAX; ++d) {
                    if (d < p.D) {
                        const int raw = HOPW ? (int)((hop_w[HOPW ? d / 4 : 0] >> (8 * (d & 3))) & 0xffu) : hop_id[HOPW ? 0 : d];
                        const int idx = pairlive ? raw : 0;
                        if (idx >= 16 && idx < p.n_edge) {         // rare id: straight to the atomic unit
#pragma unroll
                            for (int h = 0; h < HH; ++h) atomicAdd(&p.d_hop[((int64_t)d * p.n_edge + idx) * HH + h], ge[h]);
                        }
                        // staged as a 16-bit one-hot mask (bit e); 0 = "no row here"
                        hop_e[wave][d][lane] = idx < 16 ? (uint16_t)(1u << idx) : (uint16_t)0;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // A operand of lane (row m, k-group kq): [pair's id == m] for 8 pairs = bit m of 8 one-hot masks.
                // Rotating a dword of two masks left by 14 - m puts those two bits at positions 14 and 30, and
                // 0x4000 is bf16 2.0: two instructions per pair of pairs, the factor 2 is undone at the flush.
                const int m = lane & 15, kq = lane >> 4;
                const uint32_t rot = (uint32_t)(m - 14) & 31u;      // rotate RIGHT by m - 14 (mod 32)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const bf16x8 bop = *reinterpret_cast<const bf16x8*>(&hop_b[hf][m][8 * kq]);
#pragma unroll
                    for (int d = 0; d < HOP_DMAX; ++d) {
                        if (d < p.D) {
                            const uint4 eb = *reinterpret_cast<const uint4*>(&hop_e[wave][d][hf * 32 + 8 * kq]);
                            uint32_t w[4] = {eb.x, eb.y, eb.z, eb.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_alignbit(w[j], w[j], rot) & 0x40004000u;
                            hacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), bop, hacc[d], 0, 0, 0);
                        }
                    }
                }
                continue;
            }
            // Fast path (F == 1, the MobGT case): a pair's hop list is "L real hops, then zeros" (algos.pyx fills
            // the tail with -1, collator.py:87 shifts it to 0).  The zero tail goes to table row 0 of every
            // later hop slot -- a sum over pairs keyed by L alone -- so only the L real hops need a scatter.
            int L = 0;
            bool clean = true;
            if (p.F == 1 && pairlive) {
                bool tail = false;
                for (int d = 0; d < p.D; ++d) {
                    const int idx = ld_idx<TE>(p.edge_input, ebase + d);
                    if (idx == 0) tail = true;
                    else if (tail) clean = false;
                    else L = d + 1;
                }
            }
            if (p.F == 1 && !__any(!clean)) {
                wave_scatter_add<HH>(s_len, p.D + 1, st_len, s_len, pairlive ? L : -1, ge, lane);
                for (int d = 0; __any(d < L); ++d) {
                    const int idx = (pairlive && d < L) ? ld_idx<TE>(p.edge_input, ebase + d) : -1;
                    wave_scatter_add<HH>(s_hop + (size_t)d * HOP_STRIDE * HH, HOP_LDS_ROWS, HOP_STRIDE,
                                         p.d_hop + (int64_t)d * p.n_edge * HH, idx, ge, lane);
                }
            } else {
                for (int d = 0; d < p.D; ++d)
                    for (int f = 0; f < p.F; ++f) {
                        const int idx = pairlive ? ld_idx<TE>(p.edge_input, ebase + (int64_t)d * p.F + f) : -1;
                        wave_scatter_add<HH>(s_hop + (size_t)d * HOP_STRIDE * HH, HOP_LDS_ROWS, HOP_STRIDE,
                                             p.d_hop + (int64_t)d * p.n_edge * HH, idx, ge, lane);
                    }
            }
        }
    }
    }   // tiles
    if (HOPMM && p.edge_input) {
        // register v of lane (n = lane & 15, q = lane >> 4) holds row (edge id) 4q + v, column n: hi part of head n
        // for n < 8, lo part of head n - 8 otherwise
        const int n = lane & 15, q = lane >> 4;
#pragma unroll
        for (int d = 0; d < HOP_DMAX; ++d) {
            if (d < p.D) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float tot = 0.5f * (hacc[d][v] + __shfl_xor(hacc[d][v], 8, 64));   // one-hot entries are 2.0
                    if (n < 8 && tot != 0.f) lds_add(&s_hop[((size_t)d * HH + n) * HOP_STRIDE + 4 * q + v], tot);
                }
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < lds_rel * HH; t += blockDim.x) {
        const int at = (t % HH) * st_rel + t / HH;
        const float v = FX ? (float)(long long)s_rel[at] * fx_inv : f_rel[at];
        if (v != 0.f) atomicAdd(&p.d_rel[t], v);
    }
    if (p.d_poi)
        for (int t = threadIdx.x; t < lds_poi * HH; t += blockDim.x) {
            const int at = (t % HH) * st_poi + t / HH;
            const float v = FX ? (float)(long long)s_poi[at] * fx_inv : f_poi[at];
            if (v != 0.f) atomicAdd(&p.d_poi[t], v);
        }
    for (int t = threadIdx.x; t < HH; t += blockDim.x)
        if (s_vd[t] != 0.f) atomicAdd(&p.d_vdist[t], s_vd[t]);
    if (p.edge_input) {
        // zero tails: hop slot d receives, in table row 0, every pair with fewer than d+1 real hops
        for (int t = threadIdx.x; t < p.D * HH; t += blockDim.x) {
            const int d = t / HH, h = t - d * HH;
            float acc = 0.f;
            for (int l = 0; l <= d; ++l) acc += s_len[h * st_len + l];
            if (acc != 0.f) atomicAdd(&p.d_hop[(int64_t)d * p.n_edge * HH + h], acc);
        }
    }
    if (p.edge_input)
        for (int t = threadIdx.x; t < p.D * HOP_LDS_ROWS * HH; t += blockDim.x) {
            const int d = t / (HOP_LDS_ROWS * HH), rem = t - d * HOP_LDS_ROWS * HH;      // rem = row * HH + head
            const float v = s_hop[((size_t)d * HH + rem % HH) * HOP_STRIDE + rem / HH];
            if (v != 0.f && rem / HH < p.n_edge) atomicAdd(&p.d_hop[(int64_t)d * p.n_edge * HH + rem], v);
        }
}

template <typename TI, typename TE, int HH, bool HOPMM, int NW>
__global__ __launch_bounds__(NW * 64) void build_bias_bwd_kernel(const BuildParams p, int lds_rel, int lds_poi) {
    build_bias_bwd_body<TI, TE, HH, HOPMM, NW>(p, lds_rel, lds_poi, (int)blockIdx.x, (int)gridDim.x);
}

template <typename TI, typename TE, typename TB>
int launch_build(const BuildParams& p, hipStream_t st) {
    const int T = p.N + 1;
    const int nt = (int)((p.ld + TILE - 1) / TILE);
    // (tile rows cover all ld columns of the TRANSPOSED copy as well: its columns [T, ld) must read -inf -- the dK/dV pass
    // relies on it -- and ld = roundup(T, 64) can exceed roundup(T, 32))
    (void)T;
    const dim3 grid(nt, 4 * nt, p.G), block(256);
    if (p.H == 8 && (int64_t)p.G * T * T >= (1 << 20)) {
        const int nt2 = (nt + 1) / 2;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        int64_t want = (int64_t)nt2 * nt2 * p.G * 4;
        const int64_t resident = (sizeof(TB) == 2 ? 4 : 3) * (int64_t)cus;     // (35 KB of LDS per workgroup; 52 KB with an f32 tile)
        if (want > resident) want = resident;
        const dim3 grid4((unsigned)((want + 31) / 32 * 32));
        hipLaunchKernelGGL((build_bias_kernel<TI, TE, TB, 8, 4>), grid4, block, 0, st, p, SgemmRide{});
    } else if (p.H == 8) {
        SgemmRide ride = g_front_sg;                            // (taken by this launch)
        g_front_sg.tiles = 0;
        dim3 gr = grid;
        if (ride.tiles > 0) {
            ride.z0 = p.G;
            gr.z = (unsigned)(p.G + (ride.tiles + (int)(grid.x * grid.y) - 1) / (int)(grid.x * grid.y));
        }
        hipLaunchKernelGGL((build_bias_kernel<TI, TE, TB, 8, 1>), gr, block, 0, st, p, ride);
    } else if (p.H == 4) hipLaunchKernelGGL((build_bias_kernel<TI, TE, TB, 4, 1>), grid, block, 0, st, p, SgemmRide{});
    else return MOBGT_EBADDIM;
    return (int)hipGetLastError();
}

template <typename TI, typename TE>
int launch_build_b(const BuildParams& p, int bias_dtype, hipStream_t st) {
    if (bias_dtype == MOBGT_F32) return launch_build<TI, TE, float>(p, st);
    if (bias_dtype == MOBGT_BF16) return launch_build<TI, TE, bf16_t>(p, st);
    return MOBGT_EDTYPE;
}

int g_bwd_workgroups = 0;           // host side, one thread: 0 = one workgroup per compute unit (long-batch form)

template <typename TI, typename TE>
int launch_build_bwd(const BuildParams& p, hipStream_t st) {
    const int T = p.N + 1;
    const int64_t pairs = (int64_t)p.G * T * T;
    const int lds_rel = p.n_rel < 512 ? p.n_rel : 512;
    const int lds_poi = p.poi_pos ? (p.n_poi < 1024 ? p.n_poi : 1024) : 0;
    const size_t tables = (size_t)bwd_lds_dwords(lds_rel, lds_poi, p.D, p.H) * sizeof(float);
    size_t shm = 0;
    const bool hopmm = p.edge_input && p.F == 1 && p.H == 8 && p.D <= HOP_DMAX;
    // long batches: 8-wave workgroups, one per CU (each walks its units); short ones: 4-wave workgroups, up to 3 per CU
    // (more than 64 KB of dynamic LDS has to be asked for, per kernel)
#define BWD_LDS(K, HH_, HM_, NW_) do { shm = tables + bwd_stage_bytes<HH_, HM_, NW_>(); if (shm > 48 * 1024 && hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return MOBGT_EBADDIM; } while (0)
    static const int64_t long_from = (1 << 20);
    if (hopmm && pairs >= long_from) {
        const int n_units8 = ((T + 63) / 64) * ((T + 7) / 8) * p.G;
        // (one persistent workgroup per compute unit; a caller that runs this launch BESIDE other work -- ops._bias_bwd_beside: on a
        //  side stream under the tail of the backward pass -- asks for fewer, so that the other stream's kernels find free units:
        //  mobgt_build_bias_bwd_set_workgroups)
        const int wgs_cap = (g_bwd_workgroups > 0 && g_bwd_workgroups < 256) ? g_bwd_workgroups : 256;
        const dim3 grid(n_units8 < wgs_cap ? n_units8 : wgs_cap), block(512);
        BWD_LDS((build_bias_bwd_kernel<TI, TE, 8, true, 8>), 8, true, 8);
        hipLaunchKernelGGL((build_bias_bwd_kernel<TI, TE, 8, true, 8>), grid, block, shm, st, p, lds_rel, lds_poi);
        return (int)hipGetLastError();
    }
    const int n_units = ((T + 63) / 64) * ((T + 3) / 4) * p.G;
    const dim3 grid(n_units < 768 ? n_units : 768), block(256);
    if (hopmm) {
        BWD_LDS((build_bias_bwd_kernel<TI, TE, 8, true, 4>), 8, true, 4);
        hipLaunchKernelGGL((build_bias_bwd_kernel<TI, TE, 8, true, 4>), grid, block, shm, st, p, lds_rel, lds_poi);
    } else if (p.H == 8) {
        BWD_LDS((build_bias_bwd_kernel<TI, TE, 8, false, 4>), 8, false, 4);
        hipLaunchKernelGGL((build_bias_bwd_kernel<TI, TE, 8, false, 4>), grid, block, shm, st, p, lds_rel, lds_poi);
    } else if (p.H == 4) {
        BWD_LDS((build_bias_bwd_kernel<TI, TE, 4, false, 4>), 4, false, 4);
        hipLaunchKernelGGL((build_bias_bwd_kernel<TI, TE, 4, false, 4>), grid, block, shm, st, p, lds_rel, lds_poi);
    } else return MOBGT_EBADDIM;
#undef BWD_LDS
    return (int)hipGetLastError();
}

#define DISPATCH_IDX(FN, ...)                                                                      \
    do {                                                                                           \
        if (idx_dtype == MOBGT_I64 && edge_dtype == MOBGT_I64) return FN<int64_t, int64_t>(__VA_ARGS__); \
        if (idx_dtype == MOBGT_I64 && edge_dtype == MOBGT_U8) return FN<int64_t, uint8_t>(__VA_ARGS__);  \
        if (idx_dtype == MOBGT_I32 && edge_dtype == MOBGT_I32) return FN<int32_t, int32_t>(__VA_ARGS__); \
        if (idx_dtype == MOBGT_I16 && edge_dtype == MOBGT_U8) return FN<int16_t, uint8_t>(__VA_ARGS__);  \
        if (idx_dtype == MOBGT_I32 && edge_dtype == MOBGT_U8) return FN<int32_t, uint8_t>(__VA_ARGS__);  \
        return MOBGT_EDTYPE;                                                                       \
    } while (0)

}  // namespace

extern "C" int mobgt_bias_pack(const void* src, int src_dtype, int64_t s_g, int64_t s_h, int64_t s_i, int64_t s_j,
                               void* bias, void* bias_t, int bias_dtype, int G, int H, int T, int64_t ld_bias,
                               void* stream) {
    if (G <= 0 || H <= 0 || T <= 0) return MOBGT_EBADDIM;
    if (ld_bias % 32 != 0 || ld_bias < T) return MOBGT_EALIGN;
    const int nt = (int)(ld_bias / TILE);
    const dim3 grid(nt, nt, G * H), block(256);
    hipStream_t st = (hipStream_t)stream;
#define PACK(TS, TB)                                                                                              \
    hipLaunchKernelGGL((bias_pack_kernel<TS, TB>), grid, block, 0, st, reinterpret_cast<const TS*>(src), s_g, s_h, \
                       s_i, s_j, reinterpret_cast<TB*>(bias), reinterpret_cast<TB*>(bias_t), H, T, ld_bias)
    if (src_dtype == MOBGT_F32 && bias_dtype == MOBGT_F32) PACK(float, float);
    else if (src_dtype == MOBGT_F32 && bias_dtype == MOBGT_BF16) PACK(float, bf16_t);
    else if (src_dtype == MOBGT_BF16 && bias_dtype == MOBGT_F32) PACK(bf16_t, float);
    else if (src_dtype == MOBGT_BF16 && bias_dtype == MOBGT_BF16) PACK(bf16_t, bf16_t);
    else return MOBGT_EDTYPE;
#undef PACK
    return (int)hipGetLastError();
}

namespace {
// arguments of mobgt_build_bias -> BuildParams (validated); shared with mobgt_small_gcn_fwd_pack (csrc/smallgcn.hip)
int fill_bias_fwd(BuildParams& p, const float* attn_bias, const void* rel_pos, const void* poi_pos, const void* edge_input,
                  const float* rel_table, const float* poi_table, const float* hop_table, const float* vdist, void* bias, void* bias_t,
                  int G, int N, int H, int D_in, int D, int F, int n_rel, int n_poi, int n_edge, int64_t ld_bias) {
    if (G <= 0 || N <= 0 || H > BIAS_MAXH || D < 0 || D > D_in || F <= 0) return MOBGT_EBADDIM;
    if (ld_bias % 32 != 0 || ld_bias < N + 1) return MOBGT_EALIGN;
    p = BuildParams{};
    p.attn_bias = attn_bias; p.rel_pos = rel_pos; p.poi_pos = poi_pos; p.edge_input = D > 0 ? edge_input : nullptr;
    p.rel_table = rel_table; p.poi_table = poi_table; p.hop_table = hop_table; p.vdist = vdist;
    p.bias = bias; p.bias_t = bias_t;
    p.G = G; p.N = N; p.H = H; p.D_in = D_in; p.D = D; p.F = F; p.n_rel = n_rel; p.n_poi = n_poi; p.n_edge = n_edge;
    p.ld = ld_bias;
    return 0;
}
}  // namespace

/* Leaves  c = leaky_relu(a [M,K] @ b [k_b <= K rows, N] + bias)  (+ the result transposed in bf16, as mobgt_small_gemm_f32_act
 * writes it) as a job for the NEXT short-batch mobgt_build_bias launch (H = 8, G (N+1)^2 < 2^20) of this process, which runs it
 * in its split-K form as extra workgroups (round 4: the distance GCN's first layer, modelGNN.py:38-44 / 66-72 on the
 * precomputed A X, depends on parameters only).  Shapes of the split-K form only: N <= 16, K % 16 == 0, 128 <= K <= 512, a's rows
 * 16-byte aligned, M >= 1024 (MOBGT_EBADDIM otherwise).  a == NULL drops a pending job; mobgt_front_sgemm_pending() != 0: no
 * launch has taken it yet. */
extern "C" int mobgt_front_sgemm_job(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, int leaky,
                                     float slope, float* c, int64_t ldc, void* c_t_bf16, int64_t ld_t, int M, int N, int K, int k_b) {
    if (!a) { g_front_sg.tiles = 0; return 0; }
    if (M < 1024 || N <= 0 || N > 16 || K < 128 || K > 16 * 4 * 8 || (K & 15) || (lda & 3) || k_b <= 0 || k_b > K || !b || (!c && !c_t_bf16))
        return MOBGT_EBADDIM;
    if (((uintptr_t)a & 15) || (c_t_bf16 && ld_t < M)) return MOBGT_EALIGN;
    mobgt_sgemm::SgemmParams p = {};
    p.A = a; p.lda = lda; p.B = b; p.ldb = ldb; p.bias = bias; p.C = c; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.Kb = k_b;
    p.act = leaky; p.slope = slope; p.ct = reinterpret_cast<bf16_t*>(c_t_bf16); p.ldt = ld_t;
    g_front_sg.sg = p;
    g_front_sg.tiles = (M + 15) / 16;
    return 0;
}
extern "C" int mobgt_front_sgemm_pending(void) { return g_front_sg.tiles > 0; }

extern "C" int mobgt_build_bias(const float* attn_bias, const void* rel_pos, const void* poi_pos, const void* edge_input,
                                const float* rel_table, const float* poi_table, const float* hop_table,
                                const float* vdist, void* bias, void* bias_t, int G, int N, int H, int D_in, int D, int F,
                                int n_rel, int n_poi, int n_edge, int64_t ld_bias, int idx_dtype, int edge_dtype,
                                int bias_dtype, void* stream) {
    BuildParams p;
    const int rc = fill_bias_fwd(p, attn_bias, rel_pos, poi_pos, edge_input, rel_table, poi_table, hop_table, vdist, bias, bias_t, G, N, H,
                                 D_in, D, F, n_rel, n_poi, n_edge, ld_bias);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_IDX(launch_build_b, p, bias_dtype, st);
}

namespace {
// arguments of mobgt_build_bias_bwd -> BuildParams (validated); shared with mobgt_small_gcn_bwd_bias (csrc/smallgcn.hip)
int fill_bias_bwd(BuildParams& p, const void* dbias, int dbias_dtype, int n_slices, int64_t slice_stride, const float* attn_bias,
                  const void* rel_pos, const void* poi_pos, const void* edge_input, float* d_rel_table, float* d_poi_table,
                  float* d_hop_table, float* d_vdist, int G, int N, int H, int D_in, int D, int F, int n_rel, int n_poi, int n_edge,
                  int64_t ld_bias) {
    if (G <= 0 || N <= 0 || H > BIAS_MAXH || D < 0 || D > D_in || F <= 0 || n_slices < 1) return MOBGT_EBADDIM;
    if (dbias_dtype != MOBGT_F32 && dbias_dtype != MOBGT_BF16) return MOBGT_EDTYPE;
    p = BuildParams{};
    p.dbias = dbias; p.dbias_bf16 = dbias_dtype == MOBGT_BF16; p.n_slices = p.dbias_bf16 ? n_slices : 1;
    p.slice_stride = slice_stride; p.attn_bias = attn_bias; p.rel_pos = rel_pos; p.poi_pos = poi_pos;
    p.edge_input = D > 0 ? edge_input : nullptr;
    p.d_rel = d_rel_table; p.d_poi = d_poi_table; p.d_hop = d_hop_table; p.d_vdist = d_vdist;
    p.G = G; p.N = N; p.H = H; p.D_in = D_in; p.D = D; p.F = F; p.n_rel = n_rel; p.n_poi = n_poi; p.n_edge = n_edge;
    p.ld = ld_bias;
    return 0;
}
}  // namespace

extern "C" int mobgt_build_bias_bwd_set_workgroups(int n) {
    if (n < 0) return MOBGT_EBADDIM;
    g_bwd_workgroups = n;
    return 0;
}

extern "C" int mobgt_build_bias_bwd(const void* dbias, int dbias_dtype, int n_slices, int64_t slice_stride,
                                    const float* attn_bias, const void* rel_pos, const void* poi_pos,
                                    const void* edge_input, float* d_rel_table, float* d_poi_table, float* d_hop_table,
                                    float* d_vdist, int G, int N, int H, int D_in, int D, int F, int n_rel, int n_poi,
                                    int n_edge, int64_t ld_bias, int idx_dtype, int edge_dtype, void* stream) {
    BuildParams p;
    const int rc = fill_bias_bwd(p, dbias, dbias_dtype, n_slices, slice_stride, attn_bias, rel_pos, poi_pos, edge_input, d_rel_table,
                                 d_poi_table, d_hop_table, d_vdist, G, N, H, D_in, D, F, n_rel, n_poi, n_edge, ld_bias);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_IDX(launch_build_bwd, p, st);
}
