// Bias-fused multi-head attention for MobGT on gfx950 (MI355X): forward, dQ/dBias pass, dK/dV pass.
//
// Replaces graphormer/model.py:436-455 (== model_fqandtoyo.py:1687-1706) and its autograd.
//
// Orientation ("swapped QK^T"): every product is issued so that the softmax axis lies in a lane's
// registers.  With v_mfma_f32_32x32x16_bf16 computing S^T = K_tile (A) x Q^T (B), lane (n, hi) holds,
// thanks to the kappa() row permutation on the A operand, the 16 CONTIGUOUS keys key0+16*hi .. +15 of
// query row q0+n.  Consequences:
//   * the bias tile (the dominant HBM stream: H*T*T elements per graph per layer) is loaded as
//     16 contiguous elements per lane straight into the accumulator (bias is the MFMA's C operand:
//     the "+ attn_bias" of model.py:445 costs no VALU);
//   * row max / row sum are 16 in-register ops + one cross-half exchange (wave shuffle);
//   * P (and dS in the backward) is already the B operand of the following P.V product, no LDS trip.
// K and V^T (fwd), K/V/K^T (dQ pass), Q/dO and their transposes (dK/dV pass) are staged per 64-row
// chunk in LDS as bf16; fp32 inputs are rounded to bf16 while staging, accumulation is fp32.
//
// Consistent softmax (round 5).  The MFMA operands are bf16, so the probabilities that multiply V are P_b = bf16(2^(x - M))
// -- and a backward pass that re-forms P in f32, or takes rowsum(dO O) from differently rounded dO / O, computes
// dS_j = P_j (dP_j - delta) with a delta that is NOT sum_j P_j dP_j of the P and dP it uses.  The difference is 2^-9 of
// |dO||O| per row, un-cancelled, while dQ = sum_j dS_j K_j and dK rely on sum_j dS_j = 0 to cancel the component all key rows
// share (MobGT: the user embedding fused into every node of a trajectory): on real Gowalla trajectories the q / k weight
// gradients of the top layers came out 10-50 % off (tests/test_gpu_real.py, golden G8).  Now every pass works with the SAME
// probabilities and a delta that is exactly their dP-weighted mean:
//   * the running maximum M is an INTEGER in the log2 domain (ceil), so the online rescale 2^(M_old - M_new) is exact and
//     bf16 rounding commutes with it: P_b,j = bf16(2^(x_j - M)) for any integer M up to an exact power of two;
//   * the row sum l is the sum of the ROUNDED probabilities (v_dot2c_f32_bf16 with packed ones on the MFMA operand words), so
//     O = sum_j P_b,j V_b,j / sum_j P_b,j is a convex combination and sum_j P_j = 1 in the backward;
//   * the backward takes M' = rint(lse log2e), re-forms the forward's P_b,j = bf16(2^(x_j - M')) bit for bit (times the exact
//     2^(M - M')) and normalises with 2^(M' - lse log2e);
//   * delta = dO_b . O with the very bf16 dO values the dP product multiplies and O in full precision: f32 O, or bf16 O plus
//     the bf16 residual `out_lo` the forward writes beside it (2 more bytes per element: 3.5 % of the c5 traffic).
#include <type_traits>

#include "common.h"
#ifndef DB_AUX
#define DB_AUX 0              // cache policy bits of the dBias stores (timing experiments: 2 = nt)
#endif
#include "mobgt_hip.h"

namespace {

// Diagnostic build (-DATTN_STAMP, tools/attn_stamp.sh): the forward kernel sums, per wave, the shader cycles between fixed
// points of its chunk loop and writes the sums over the first LSE values of its rows.  Never defined in the shipped library.
#ifdef ATTN_STAMP
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        unsigned long long t_;                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        st_acc[k] += (uint32_t)(t_ - st_last);                                                     \
        st_last = t_;                                                                              \
    } while (0)
#else
#define STAMP(k)
#endif

// A/B switches for tools/attn_ab.sh (each variant is its own library; the shipped build uses the defaults)
#ifndef ATTN_DEADSKIP
#define ATTN_DEADSKIP 1       // waves whose 32 rows all lie beyond T skip the tile loop
#endif
// (round 4 measured and round 5 removed: dead-wave skipping and even tile dealing in the backward passes, a late prologue prefetch,
//  packed-f32 arithmetic around the exponentials, dropout masks in the shadow of the QK^T products -- all inside the +-2 us
//  run-to-run band at c5: profiles/r4_attn_ab.txt)

constexpr int KC = 64;        // keys (or queries, in the dK/dV pass) staged per LDS chunk = 2 MFMA tiles
constexpr int ROWP = 40;      // row-major tile row pitch in bf16 (32 + 8: odd multiple of 16 B)
constexpr int COLP = KC + 8;  // transposed tile row pitch in bf16 (144 B = 9 * 16 B)

struct AttnParams {
    const void *q, *k, *v, *bias, *bias_t, *out, *dout;
    void *o, *dq, *dk, *dv;
    float* lse;
    const float* lse_in;
    void* dbias;              // f32 (read-modify-write when `accumulate`) or bf16 (write-only slice of this layer)
    int dbias_bf16;
    float* delta;
    void* o_lo;               // forward, bf16 I/O: bf16 residual O - bf16(O) beside `o` (same layout), or null
    const void* out_lo;       // backward, bf16 I/O: that residual, or null (delta then sees the rounded O only)
    float* dq_acc;            // one-pass backward: [G, T, H * d] f32, the dQ sums; ZERO on entry, zero again when the call's last launch has run
    int G, H, T;
    int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv, ld_bias;
    float scale, inv_keep;
    uint32_t drop_thr;
    int thr_s;                // drop_thr - 32768: the signed 16-bit form the v2 keep rule compares against (common.h)
    uint64_t seed;
    const uint64_t* seed_dev;
    int accumulate;
    int n_first;              // attn_bwd_both_kernel: workgroups of the dQ part
    int nq;                   // workgroups per (graph, head): the ceil(T/32) 32-row wave tiles are dealt out evenly over them
};

// Staging of a [KC x D] row-major slab (rows row0.., row stride ld, head column offset already applied) into
// LDS as bf16, split in two halves so that the global loads of chunk c+1 are IN FLIGHT while chunk c is being
// computed: `load` pulls the slab into registers (KC*4/NT pieces of 8 elements per thread; rows >= T and
// columns >= D read as zero), `store` -- one chunk later -- scales by `mul`, rounds to bf16 and writes the
// row-major image `rm` (if RM) and/or the transposed image `tr` (if TR).  A kernel's time here is the serial
// chain of its chunks (3-4 co-resident workgroups per CU never saturate anything), so taking two L2 round
// trips out of every chunk matters more than any instruction count.
template <typename TQ> struct Raw8;
template <> struct Raw8<bf16_t> {
    uint4 r;
    __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void zero() { r = make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ bf16x8 as_bf16() const { return __builtin_bit_cast(bf16x8, r); }
    __device__ __forceinline__ void get(float (&v)[8]) const {
        v[0] = bf16_lo(r.x); v[1] = bf16_hi(r.x); v[2] = bf16_lo(r.y); v[3] = bf16_hi(r.y);
        v[4] = bf16_lo(r.z); v[5] = bf16_hi(r.z); v[6] = bf16_lo(r.w); v[7] = bf16_hi(r.w);
    }
};
template <> struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ bf16x8 as_bf16() const { float v[8]; get(v); return pack8(v); }
    __device__ __forceinline__ void get(float (&v)[8]) const {
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};

template <int D, typename TQ, int NT>
struct Slab {
    static constexpr int ITEMS = KC * 4 / NT;
    Raw8<TQ> it[ITEMS];
    __device__ __forceinline__ void load(const TQ* __restrict__ src, int64_t ld, int row0, int T) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int e = threadIdx.x + k * NT, r = e >> 2, c0 = (e & 3) * 8;
            if (c0 < D && row0 + r < T) it[k].load(src + (int64_t)(row0 + r) * ld + c0);
            else it[k].zero();
        }
    }
    // the same with rows >= T clamped onto row T - 1 instead of zero-filled: no branch around the loads (the compiler's
    // wait-count bookkeeping stays exact across the chunk loop), and every kernel here multiplies rows >= T by an exact zero
    // (masked keys: P = 0; queries beyond T in the dK/dV pass: bias_t = -inf) so any FINITE stand-in serves
    __device__ __forceinline__ void load_clamped(const TQ* __restrict__ src, int64_t ld, int row0, int T) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int e = threadIdx.x + k * NT, r = e >> 2, c0 = (e & 3) * 8;
            if (D % 32 == 0 || c0 < D) it[k].load(src + (int64_t)min(row0 + r, T - 1) * ld + c0);
            else it[k].zero();
        }
    }
    template <bool RM, bool TR, bool UNIT>
    static __device__ __forceinline__ void put(const Raw8<TQ>& x, int e, float mul, bf16_t (*rm)[ROWP], bf16_t (*tr)[COLP]) {
        const int r = e >> 2, c0 = (e & 3) * 8;
        bf16x8 b;
        if (UNIT) {
            b = x.as_bf16();                                  // bf16 in, no scaling: the 16 bytes as they are
        } else {
            float v[8];
            x.get(v);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] *= mul;
            b = pack8(v);
        }
        if (RM) *reinterpret_cast<bf16x8*>(&rm[r][c0]) = b;
        if (TR) {
#pragma unroll
            for (int i = 0; i < 8; ++i) tr[c0 + i][r] = b[i];
        }
    }
    // UNIT: mul == 1 (then bf16 input is copied as it is, no unpack / multiply / repack)
    template <bool RM, bool TR, bool UNIT = false>
    __device__ __forceinline__ void store(float mul, bf16_t (*rm)[ROWP], bf16_t (*tr)[COLP]) const {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) put<RM, TR, UNIT>(it[k], threadIdx.x + k * NT, mul, rm, tr);
    }
    // load + store of one chunk, piece by piece (the single-chunk kernels of graphs with T <= 64: nothing to overlap)
    template <bool RM, bool TR, bool UNIT = false>
    static __device__ __forceinline__ void direct(const TQ* __restrict__ src, int64_t ld, int row0, int T, float mul,
                                                  bf16_t (*rm)[ROWP], bf16_t (*tr)[COLP]) {
        for (int e = threadIdx.x; e < KC * 4; e += NT) {
            const int r = e >> 2, c0 = (e & 3) * 8;
            Raw8<TQ> x;
            if (c0 < D && row0 + r < T) x.load(src + (int64_t)(row0 + r) * ld + c0);
            else x.zero();
            put<RM, TR, UNIT>(x, e, mul, rm, tr);
        }
    }
};

template <typename TQ>
__device__ __forceinline__ void load_frag(const TQ* rowptr, bool valid, float mul, bf16x8& f, float (*keep)[8] = nullptr) {
    float v[8];
    if (valid) {
        load8(rowptr, v);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
    if (keep) {
#pragma unroll
        for (int i = 0; i < 8; ++i) (*keep)[i] = v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= mul;
    f = pack8(v);
}

// Exchange with the partner lane (lane ^ 32).  v_permlane32_swap swaps the upper half of its first operand with the
// lower half of its second in ONE VALU instruction (__shfl_xor lowers to ds_bpermute_b32, an LDS round trip of ~100
// cycles in the serial chain of every tile).  Given the same value in both operands, `lo` ends up holding the lower
// half's value in all 64 lanes and `hi` the upper half's.
__device__ __forceinline__ void half_values(float v, float& lo, float& hi) {
    // inline asm, not __builtin_amdgcn_permlane32_swap: with both operands derived from one value hipcc (ROCm 7.2) drops
    // the builtin's second result and uses the first twice (seen in the .s: v_add_f32 v, v0, v0 after the swap).
    // s_nop 1 = the 2 wait states the hazard rule wants between a VALU write of an operand and the swap.
    uint32_t a = __builtin_bit_cast(uint32_t, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    lo = __builtin_bit_cast(float, a);
    hi = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float xhalf_max(float v) {        // max(v, partner's v)
    float lo, hi;
    half_values(v, lo, hi);
    return fmaxf(lo, hi);
}
__device__ __forceinline__ float xhalf_sum(float v) {        // v + partner's v
    float lo, hi;
    half_values(v, lo, hi);
    return lo + hi;
}

// bf16 pairs as they sit in an MFMA operand word (even element low): packing, the f32 values of a packed pair, and the sum of a
// pair added to an f32 accumulator in ONE instruction (v_dot2c_f32_bf16 against packed ones: products with 1.0 are exact).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    const bf16x2_t v = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float add_pair(uint32_t w, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, 0x3f803f80u), acc, false);
}
// the backward's view of a row's forward statistics: M' = rint(lse log2 e) (any integer serves: bf16 rounding commutes with
// powers of two) and 1 / l' = 2^(M' - lse log2 e), l' = the sum of the row's bf16(2^(x - M'))
__device__ __forceinline__ void row_norm(float lse, float& mq, float& il) {
    const float lse2 = lse * MOBGT_LOG2E;
    mq = __builtin_rintf(lse2);
    il = fast_exp2(mq - lse2);
}

// ---- bias tiles: coalesced from HBM, re-distributed through a wave-private LDS image ------------------------------------------
// The MFMA C operand wants lane (n, hi) to hold 16 contiguous keys of query row n.  Loaded that way (rounds 1-2) a wave's
// load instruction touched 32 different rows -- 64 separate 16-byte pieces in 32 different 128-byte lines: the texture
// addresser walks such an instruction one lane per clock (a quarter of its rate for a coalesced one; the same effect was
// measured on the chain kernel's weight loads), and every line was requested twice, by two tiles 32 keys apart,
// with 64 KB of other waves' lines in between in a 32 KB L1.  At c5 that is ~50 k texture-addresser cycles per CU per launch
// for the bias alone.  Now a wave fetches its 32 rows x 64 keys "super-tile" as WHOLE 128-byte row segments (bf16; 8 lanes
// per row, 8 rows per instruction: 8 full lines per instruction, each line requested once), parks the registers in a
// wave-private LDS image two chunks later and every lane reads its 16 keys per MFMA tile from there (row pitch +16 B: the
// 16 rows of a ds_read_b128 lane group fall on different banks).  Rows are line-aligned because ld_bias % 64 == 0.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// NT: the tiles are read with non-temporal loads -- long batches (T > 64), where a launch reads its bias exactly once: c5 forward
// 51.7 -> 50.2 us, 55.5 -> 53.6 us inside the S-BIG step.  Short batches re-read one small bias in every layer: plain loads.
template <typename TB, bool NT = false>
struct BiasStage {
    static constexpr int SEG = 64 * (int)sizeof(TB);         // bytes of a row segment (64 keys)
    static constexpr int PITCH = SEG + 16;                   // LDS row pitch
    static constexpr int NPIECE = SEG / 16;                  // 16-byte pieces per row segment (8 / 16)
    static constexpr int RPI = 64 / NPIECE;                  // rows per load instruction (8 / 4)
    static constexpr int NI = 32 / RPI;                      // load instructions per super-tile (4 / 8)
    static constexpr int BYTES = 32 * PITCH;                 // LDS image of one wave
    // (native vectors, not HIP's uint4 struct: copying that struct to LDS is a memcpy the optimiser would not split, and the
    // whole ring then lived in scratch memory)
    u32x4 r[NI];
    uint32_t off[NI];          // this lane's byte offset inside the wave's 32-row block, per load instruction
    // rows beyond row_max belong to nobody: their loads are clamped onto row_max
    __device__ __forceinline__ void init(int64_t ld, int row_max, int lane) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            off[j] = (uint32_t)min(RPI * j + lane / NPIECE, row_max) * (uint32_t)(ld * (int64_t)sizeof(TB)) + 16u * (uint32_t)(lane % NPIECE);
    }
    // `rows`: WAVE-UNIFORM pointer to element [row0][0] of this wave's 32 rows (an SGPR base: the loads then take the
    // scalar-base + 32-bit-offset form and cost no address arithmetic); c = chunk of 64 columns
    __device__ __forceinline__ void load(const TB* __restrict__ rows, int c) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>(rows + c * 64);
#pragma unroll
        for (int j = 0; j < NI; ++j) r[j] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + off[j])) : *reinterpret_cast<const u32x4*>(base + off[j]);
    }
    __device__ __forceinline__ void park(unsigned char* img, int lane) const {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            *reinterpret_cast<u32x4*>(img + (RPI * j + lane / NPIECE) * PITCH + 16 * (lane % NPIECE)) = r[j];
    }
    // the 16 elements [32 t + 16 hi, +16) of row n -> accumulator registers
    static __device__ __forceinline__ void to_acc(const unsigned char* img, int n, int hi, int t, f32x16& s) {
        const unsigned char* src = img + n * PITCH + (32 * t + 16 * hi) * (int)sizeof(TB);
        if constexpr (sizeof(TB) == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 w = reinterpret_cast<const u32x4*>(src)[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) { s[8 * i + 2 * k] = bf16_lo(w[k]); s[8 * i + 2 * k + 1] = bf16_hi(w[k]); }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                typedef float f32x4_t __attribute__((ext_vector_type(4)));
                const f32x4_t w = reinterpret_cast<const f32x4_t*>(src)[i];
                s[4 * i] = w[0]; s[4 * i + 1] = w[1]; s[4 * i + 2] = w[2]; s[4 * i + 3] = w[3];
            }
        }
    }
};


// The 32-row tiles of one (graph, head) -- nwt = ceil(T / 32), one per wave -- dealt out evenly over nq workgroups of NW
// waves: workgroup qt owns tiles [qt * nwt / nq, (qt + 1) * nwt / nq), at most NW of them (nq >= ceil(nwt / NW)); a wave beyond
// its workgroup's count has no rows (row0 = T).  Round 2 gave every workgroup NW consecutive tiles: at T = 785 that is
// 4,4,4,4,4,4,1 -- 896 workgroups of which every seventh is nearly empty, on 1024 / 768 / 512 resident slots (forward / dQ /
// dK-dV pass): compute units with four workgroups next to units with three, or a mostly idle last round.  Now the host picks
// nq (choose_nq below: 8 at c5 -> 1024 workgroups of 3 or 4 live waves, whole rounds on every compute unit).
__device__ __forceinline__ int wave_row0(int qt, int nq, int T, int wave_uniform, int NWv) {
    const int nwt = (T + 31) >> 5;
    const int t0 = qt * nwt / nq, t1 = (qt + 1) * nwt / nq;
    (void)NWv;
    return wave_uniform < t1 - t0 ? (t0 + wave_uniform) * 32 : T;
}
__device__ __forceinline__ int wg_tile0(int qt, int nq, int T) { return qt * ((T + 31) >> 5) / nq; }
// =================================================================================== forward
// (four waves per SIMD for the long-batch instantiation -- the d = 24 one with dropout had come out at 130 registers: three)
template <int D, typename TQ, typename TB, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, (NW == 4 && sizeof(TB) == 2) ? 4 : 1) void attn_fwd_kernel(const AttnParams p) {
    constexpr int KS = (D + 15) / 16;
    constexpr int NT = NW * 64;
    constexpr bool PIPE = NW == 4;                     // NW < 4 is launched for T <= 64 only: one chunk, two tiles
    constexpr int NB = PIPE ? 2 : 1;                   // K / V images in LDS (double-buffered when there is a chunk loop)
    __shared__ __attribute__((aligned(16))) bf16_t Ksb[NB][KC][ROWP];
    __shared__ __attribute__((aligned(16))) bf16_t Vtb[NB][32][COLP];

    const int T = p.T, H = p.H;
    const int nQ = p.nq;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = lid % nQ, gh = lid / nQ;
    const int g = gh / H, h = gh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const int q0w = wave_row0(qt, nQ, T, __builtin_amdgcn_readfirstlane(wave), NW);   // this wave's first query row (an SGPR)
    const int my_q = q0w + n;
    const bool q_ok = my_q < T;
    const int qc = q_ok ? my_q : T - 1;

    const TQ* Q = reinterpret_cast<const TQ*>(p.q) + (int64_t)g * T * p.ldq + h * D;
    const TQ* K = reinterpret_cast<const TQ*>(p.k) + (int64_t)g * T * p.ldk + h * D;
    const TQ* V = reinterpret_cast<const TQ*>(p.v) + (int64_t)g * T * p.ldv + h * D;
    __shared__ __attribute__((aligned(16))) unsigned char Bs[NW][BiasStage<TB>::BYTES];
    const bool wave_live = !ATTN_DEADSKIP || q0w < T;                          // (wave-uniform)
    const TB* brows = reinterpret_cast<const TB*>(p.bias) + ((int64_t)gh * T + min(q0w, T - 1)) * p.ld_bias;
    const int brow_max = max(T - 1 - q0w, 0);
    unsigned char* bimg = Bs[wave];

    // The prologue's loads are all ISSUED before anything waits, in the order they are needed: the first chunk's K / V rows
    // (the whole workgroup waits for them at the first barrier), this lane's Q fragment, the first two bias super-tiles, the
    // second chunk's K / V.  (Round 2 loaded Q, waited, loaded 8 KB of bias per wave, then K / V: at a cold start, with every
    // wave of the launch in its prologue at once, the first MFMA sat behind three serial round trips.)
    const int nchunk = (T + KC - 1) / KC;
    Slab<D, TQ, PIPE ? NT : KC * 4> kreg, vreg;        // (one dummy piece when not pipelined)
    if (PIPE) {
        kreg.load_clamped(K, p.ldk, 0, T);
        vreg.load_clamped(V, p.ldv, 0, T);
    }
    Raw8<TQ> qraw[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)          // (a fragment beyond D is read from column 0 and multiplied by zero below)
        qraw[ks].load(Q + (int64_t)qc * p.ldq + (ks * 16 + 8 * hi < D ? ks * 16 + 8 * hi : 0));
    BiasStage<TB, PIPE> ring[2];
    ring[0].init(p.ld_bias, brow_max, lane);
    ring[0].load(brows, 0);
    if (PIPE) {
#pragma unroll
        for (int j = 0; j < BiasStage<TB>::NI; ++j) ring[1].off[j] = ring[0].off[j];
        ring[1].load(brows, min(1, nchunk - 1));
    }

    uint32_t rowh = 0;
    uint64_t seed = 0;
    if (DROP) {
        seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
        rowh = dropout_row_hash(seed, (uint32_t)(gh * T + qc));
    }

    float m = MOBGT_NEG_BIG, l = 0.f;                  // m: running maximum in the LOG2 domain, integer-valued (header: consistent softmax)
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;

    // The chunk loop (T > 64), as the in-kernel stamps of round 3 shaped it (tools/attn_stamp.py: a wave spent 30 % of its
    // life waiting for loads that had been issued only ONE chunk -- ~1.5 us of its own work -- earlier, whatever the nominal
    // prefetch distance: vmcnt retires in order, and the compiler drained the counter at the top of every other chunk):
    //   * bias: a chunk's 32 x 64 super-tile (4 KB per wave, bf16) is requested two chunks before it is parked in LDS;
    //   * K / V: chunk c + 2 is requested while chunk c is computed; chunk c + 1 goes from registers into the OTHER LDS image
    //     at the END of chunk c (its loads have then had a whole chunk + this chunk's tiles to arrive), so a chunk needs ONE
    //     barrier (the next image is complete / this image's readers are done) instead of two;
    //   * no load sits behind a branch (row and chunk indices are clamped instead; re-read data is never used), the loop is
    //     unrolled by two with the odd tail peeled, so the compiler's wait counts are exact.
    if (PIPE) {
        kreg.template store<true, false, true>(1.f, Ksb[0], nullptr);
        vreg.template store<false, true, true>(1.f, nullptr, Vtb[0]);
        kreg.load_clamped(K, p.ldk, min(1, nchunk - 1) * KC, T);
        vreg.load_clamped(V, p.ldv, min(1, nchunk - 1) * KC, T);
    } else {
        Slab<D, TQ, NT>::template direct<true, false, true>(K, p.ldk, 0, T, 1.f, Ksb[0], nullptr);
        Slab<D, TQ, NT>::template direct<false, true, true>(V, p.ldv, 0, T, 1.f, nullptr, Vtb[0]);
    }
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float v[8];
        qraw[ks].get(v);
        const float mul = (q_ok && (ks * 16 + 8 * hi < D)) ? p.scale : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= mul;
        qf[ks] = pack8(v);
    }
    __syncthreads();

#ifdef ATTN_STAMP
    uint32_t st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last, st_real0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_real0)::"memory");      // 100 MHz wall clock
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    auto chunk = [&](const int c, BiasStage<TB, PIPE>& bst, const int b) {
        bf16_t (*Ks)[ROWP] = Ksb[PIPE ? b : 0];
        bf16_t (*Vt)[COLP] = Vtb[PIPE ? b : 0];
        // this wave's bias super-tile: registers -> its LDS image (the image's readers of the previous chunk are this very
        // wave's earlier ds_reads: LDS operations of one wave complete in order), then the slot is refilled two chunks ahead
        // (past the end the last chunk is re-read: a branch around the load made the compiler copy the slot)
        if (wave_live) bst.park(bimg, lane);
        STAMP(0);                                          // waited for this chunk's bias super-tile
        // (also by a wave without rows -- it re-reads row T - 1, an L2 hit: a load behind a branch would cost every wave its
        // exact wait counts, see above)
        if (PIPE) bst.load(brows, min(c + 2, nchunk - 1));
        // (a wave whose 32 query rows all lie beyond T -- three of the four waves of every (graph, head)'s last workgroup at
        // T = 785 -- only helps staging K / V: it owns no output, and its tiles would take a ninth of the chip's vector issue)
        if (wave_live)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key0 = c * KC + t * 32;
            if (key0 >= T) break;
            f32x16 s;
            BiasStage<TB>::to_acc(bimg, n, hi, t, s);
            if (key0 + 32 > T) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (key0 + 16 * hi + i >= T) s[i] = -INFINITY;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Ks[t * 32 + kappa(n)][ks * 16 + 8 * hi]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s, 0, 0, 0);
            }
#ifdef ATTN_STAMP_FINE
            { float d_; asm volatile("v_mov_b32 %0, %1" : "=v"(d_) : "v"(s[15])); asm volatile("" :: "v"(d_)); }
            STAMP(1);                                      // bias tile from LDS + QK^T result available
#endif
            // online softmax over this lane's 16 keys + the partner half's 16; the running maximum is kept as an INTEGER in the
            // log2 domain, so the rescale below is an exact power of two (header: consistent softmax)
            float tmax = s[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
            tmax = xhalf_max(tmax);
            const float m_new = fmaxf(m, __builtin_ceilf(tmax * MOBGT_LOG2E));
            if (__any(m_new > m)) {
                const float alpha = fast_exp2(m - m_new);
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] *= alpha;
                l *= alpha;
                m = m_new;
            }
            // dropout rule v2 (common.h; this lane's 16 keys = one block), applied to the PACKED probabilities: a pair's word w
            // carries two 16-bit uniforms in the order the pair is packed (even key low), so the pair's keep mask is two packed
            // 16-bit instructions -- saturating (thr - 1) - w, arithmetic shift by 15 -- and one AND on the packed pair, instead
            // of a sign extension, two compares and two selects on the f32 values.  1/(1-p) is applied once, to the output row.
            uint32_t hb = 0;
            if (DROP) hb = attn_drop_block(seed, rowh, (uint32_t)((key0 >> 4) + hi));
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 pw;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = 8 * s2 + 2 * k;
                    pw[k] = pack2(fast_exp2(fmaf(s[i], MOBGT_LOG2E, -m)), fast_exp2(fmaf(s[i + 1], MOBGT_LOG2E, -m)));
                    l = add_pair(pw[k], l);                // the row sum of the ROUNDED probabilities, before dropout
                    if (DROP) pw[k] &= attn_drop_keep_mask2(attn_drop_word(hb, attn_drop_mult(4 * s2 + k)), p.thr_s);
                }
                const bf16x8 pb = __builtin_bit_cast(bf16x8, pw);
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Vt[kappa(n)][t * 32 + 16 * hi + 8 * s2]);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb, o, 0, 0, 0);
            }
#ifdef ATTN_STAMP_FINE
            { float d_; asm volatile("v_mov_b32 %0, %1" : "=v"(d_) : "v"(o[15])); asm volatile("" :: "v"(d_)); }
#endif
            STAMP(4 + t);                                  // one 32-key tile: QK^T, softmax, dropout, PV
        }
        if (PIPE) {
            // chunk c + 1: registers -> the other image; chunk c + 2: requested now; one barrier per chunk
            kreg.template store<true, false, true>(1.f, Ksb[b ^ 1], nullptr);
            vreg.template store<false, true, true>(1.f, nullptr, Vtb[b ^ 1]);
            STAMP(2);                                      // waited for the next chunk's K / V rows, stored them
            kreg.load_clamped(K, p.ldk, min(c + 2, nchunk - 1) * KC, T);
            vreg.load_clamped(V, p.ldv, min(c + 2, nchunk - 1) * KC, T);
            __syncthreads();
            STAMP(3);                                      // the chunk's barrier
        }
    };

    if (PIPE) {
        int c = 0;
        for (; c + 1 < nchunk; c += 2) {
            chunk(c, ring[0], 0);
            chunk(c + 1, ring[1], 1);
        }
        if (c < nchunk) chunk(c, ring[0], 0);
    } else {
        chunk(0, ring[0], 0);
    }

    const float ltot = xhalf_sum(l);
    const float inv = (DROP ? p.inv_keep : 1.f) / ltot;
    if (q_ok) {
        TQ* O = reinterpret_cast<TQ*>(p.o) + ((int64_t)g * T + my_q) * p.ldo + h * D + 16 * hi;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (16 * hi + 8 * j < D) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = o[8 * j + i] * inv;
                store8(O + 8 * j, v);
            }
        }
        if (hi == 0) p.lse[(int64_t)gh * T + my_q] = m * MOBGT_LN2 + logf(ltot);
        if constexpr (std::is_same<TQ, bf16_t>::value) {
            // bf16 output: the rounding residual beside it, for the backward's delta = dO . O (header: consistent softmax)
            if (p.o_lo) {
                bf16_t* OL = reinterpret_cast<bf16_t*>(p.o_lo) + ((int64_t)g * T + my_q) * p.ldo + h * D + 16 * hi;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (16 * hi + 8 * j < D) {
                        float v[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float x = o[8 * j + i] * inv;
                            v[i] = x - (float)(bf16_t)x;
                        }
                        store8(OL + 8 * j, v);
                    }
                }
            }
        }
    }
#ifdef ATTN_STAMP
    STAMP(6);                                              // epilogue
    {
        unsigned long long r1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
        st_acc[7] = (uint32_t)(r1 - st_real0);             // wave lifetime in 10 ns units
        st_acc[6] = (uint32_t)(st_real0 & 0xffffffu);      // start time (10 ns units, low 24 bits: exact in a float)
    }
    __syncthreads();
    if (lane == 0 && q0w + 8 <= T)
        for (int k = 0; k < 8; ++k) p.lse[(int64_t)gh * T + q0w + k] = (float)st_acc[k];
#endif
}

// ======================================================================= backward, pass 1: dQ + dBias
// Same decomposition as the forward (one wave = 32 query rows, sweep over keys).
template <int D, typename TQ, typename TB, int NW, bool DROP>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnParams& p, const int bid, const int nwg) {
    constexpr int KS = (D + 15) / 16;
    constexpr int NT = NW * 64;
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KC][ROWP];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KC][ROWP];
    __shared__ __attribute__((aligned(16))) bf16_t Kt[32][COLP];

    const int T = p.T, H = p.H;
    const int nQ = p.nq;
    const int lid = xcd_remap(bid, nwg);
    const int qt = lid % nQ, gh = lid / nQ;
    const int g = gh / H, h = gh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const int q0w = wave_row0(qt, nQ, T, __builtin_amdgcn_readfirstlane(wave), NW);   // this wave's first query row (an SGPR)
    const int my_q = q0w + n;
    const bool q_ok = my_q < T;
    const int qc = q_ok ? my_q : T - 1;

    const TQ* Q = reinterpret_cast<const TQ*>(p.q) + (int64_t)g * T * p.ldq + h * D;
    const TQ* K = reinterpret_cast<const TQ*>(p.k) + (int64_t)g * T * p.ldk + h * D;
    const TQ* V = reinterpret_cast<const TQ*>(p.v) + (int64_t)g * T * p.ldv + h * D;
    const TQ* O = reinterpret_cast<const TQ*>(p.out) + (int64_t)g * T * p.ldo + h * D;
    const TQ* dO = reinterpret_cast<const TQ*>(p.dout) + (int64_t)g * T * p.ldo + h * D;
    __shared__ __attribute__((aligned(16))) unsigned char Bs[NW][BiasStage<TB>::BYTES];
    const TB* brows = reinterpret_cast<const TB*>(p.bias) + ((int64_t)gh * T + min(q0w, T - 1)) * p.ld_bias;
    const int brow_max = max(T - 1 - q0w, 0);
    unsigned char* bimg = Bs[wave];
    const int64_t dboff = ((int64_t)gh * T + qc) * p.ld_bias + 16 * hi;
    float* dbrow = p.dbias && !p.dbias_bf16 ? reinterpret_cast<float*>(p.dbias) + dboff : nullptr;
    // bf16 dBias goes out through LDS: written straight from the MFMA layout, a store instruction carried 64 separate
    // 16-byte pieces (two per row) and the 158 MB of dBias at c5 took 75 of the pass's 128 us (2.1 TB/s; without the
    // writes the pass ran in 53 us).  A wave's chunk of dS is 32 rows x 64 keys x 2 B = one 128-byte line per row: the
    // tile is parked in LDS in the MFMA layout and read back row-contiguously, 8 full lines per store instruction.
    __shared__ __attribute__((aligned(16))) bf16_t dSs[NW][32][KC + 8];
    const bool db16 = p.dbias && p.dbias_bf16;
    bf16_t* db16_base = reinterpret_cast<bf16_t*>(p.dbias) + ((int64_t)gh * T + min(q0w, T - 1)) * p.ld_bias;
    auto flush_dbias = [&](const int c) {
        if (!db16) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (lane >> 3) + 8 * j, col = c * KC + 8 * (lane & 7);
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(&dSs[wave][row][8 * (lane & 7)]);
            // columns [T, ld_bias) exist in the buffer and receive the zeros that masked keys produce; a tile that
            // lies entirely beyond ld_bias was never computed
            if (q0w + row < T && col + 8 <= p.ld_bias)
                *reinterpret_cast<bf16x8*>(db16_base + (int64_t)row * p.ld_bias + col) = v;
        }
    };

    // Consistent softmax (header): this row's probabilities are re-formed as the forward's bf16 values P_b,j = bf16(2^(x_j - M')),
    // their normalisation 1 / l' rides on the dO fragment (dO~ = bf16(dO / l'): the B operand of the dP product), and
    // delta = dO~ . O -- with the ROUNDED dO~ and O in full precision -- is then exactly sum_j P_b,j dP_j / sum_j P_b,j.
    float mq, il;
    row_norm(p.lse_in[(int64_t)gh * T + qc], mq, il);
    bf16x8 qf[KS], dof[KS];
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bool ok = q_ok && (ks * 16 + 8 * hi < D);
        load_frag(Q + (int64_t)qc * p.ldq + ks * 16 + 8 * hi, ok, p.scale, qf[ks]);
        load_frag(dO + (int64_t)qc * p.ldo + ks * 16 + 8 * hi, ok, il, dof[ks]);
        float ov[8];
        bf16x8 unused;
        load_frag(O + (int64_t)qc * p.ldo + ks * 16 + 8 * hi, ok, 1.f, unused, &ov);
        if constexpr (std::is_same<TQ, bf16_t>::value) {
            if (p.out_lo && ok) {
                float lo[8];
                load8(reinterpret_cast<const bf16_t*>(p.out_lo) + ((int64_t)g * T + qc) * p.ldo + h * D + ks * 16 + 8 * hi, lo);
#pragma unroll
                for (int i = 0; i < 8; ++i) ov[i] += lo[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) dpart = fmaf((float)dof[ks][i], ov[i], dpart);
    }
    const float delta = xhalf_sum(dpart);                          // rowsum(dO~ * O)

    uint32_t rowh = 0;
    uint64_t seed = 0;
    if (DROP) {
        seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
        rowh = dropout_row_hash(seed, (uint32_t)(gh * T + qc));
    }

    f32x16 dq;
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[i] = 0.f;

    // bias prefetch ring as in the forward: two 4 KB super-tiles in flight per wave (one tile ahead, the pass ran at the
    // same speed with and without dropout -- it was waiting for its bias tiles, not computing)
    const int nchunk = (T + KC - 1) / KC;
    constexpr bool PIPE = NW == 4;
    BiasStage<TB, PIPE> ring[2];
    ring[0].init(p.ld_bias, brow_max, lane);
    ring[0].load(brows, 0);
    if (PIPE) {
#pragma unroll
        for (int j = 0; j < BiasStage<TB>::NI; ++j) ring[1].off[j] = ring[0].off[j];
        ring[1].load(brows, min(1, nchunk - 1));
    }
    Slab<D, TQ, PIPE ? NT : KC * 4> kreg, vreg;
    if (PIPE) {
        kreg.load_clamped(K, p.ldk, 0, T);
        vreg.load_clamped(V, p.ldv, 0, T);
    }
    auto chunk = [&](const int c, BiasStage<TB, PIPE>& bst) {
        bst.park(bimg, lane);                                                  // (see the forward kernel)
        if (PIPE) bst.load(brows, min(c + 2, nchunk - 1));                    // (never behind a branch: see the forward)
        __syncthreads();
        if (PIPE) {
            kreg.template store<true, true, true>(1.f, Ks, Kt);
            vreg.template store<true, false, true>(1.f, Vs, nullptr);
        } else {
            Slab<D, TQ, NT>::template direct<true, true, true>(K, p.ldk, c * KC, T, 1.f, Ks, Kt);
            Slab<D, TQ, NT>::template direct<true, false, true>(V, p.ldv, c * KC, T, 1.f, Vs, nullptr);
        }
        __syncthreads();
        if (PIPE) {                                                            // (unconditional, clamped: see the forward)
            kreg.load_clamped(K, p.ldk, min(c + 1, nchunk - 1) * KC, T);
            vreg.load_clamped(V, p.ldv, min(c + 1, nchunk - 1) * KC, T);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key0 = c * KC + t * 32;
            if (key0 >= T) break;
            f32x16 s, dp;
            BiasStage<TB>::to_acc(bimg, n, hi, t, s);
            const bool tail = key0 + 32 > T;
            if (tail) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (key0 + 16 * hi + i >= T) s[i] = -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 ak = *reinterpret_cast<const bf16x8*>(&Ks[t * 32 + kappa(n)][ks * 16 + 8 * hi]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qf[ks], s, 0, 0, 0);
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(&Vs[t * 32 + kappa(n)][ks * 16 + 8 * hi]);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, dof[ks], dp, 0, 0, 0);
            }
            float ds[16];
            uint32_t hb = 0;
            if (DROP) hb = attn_drop_block(seed, rowh, (uint32_t)((key0 >> 4) + hi));     // rule v2: 16 keys = one block
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const uint32_t w = DROP ? attn_drop_word(hb, attn_drop_mult(m)) : 0u;
                // the forward's probabilities, bit for bit (up to the exact 2^(M - M')): rounded to bf16 as a pair, read back as f32
                const uint32_t pw = pack2(fast_exp2(fmaf(s[2 * m], MOBGT_LOG2E, -mq)), fast_exp2(fmaf(s[2 * m + 1], MOBGT_LOG2E, -mq)));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = 2 * m + u;
                    const float pr = u ? bf16_hi(pw) : bf16_lo(pw);
                    float dd = DROP ? fmaf(dp[i], p.inv_keep, -delta) : dp[i] - delta;
                    if (DROP) {
                        const bool keep = u ? attn_drop_keep_odd(w, p.thr_s) : attn_drop_keep_even(w, p.thr_s);
                        dd = keep ? dd : -delta;
                    }
                    ds[i] = pr * dd;                       // (1 / l' is inside dp and delta: it rode in on the dO fragment)
                }
            }
            if (dbrow && q_ok) {
                float* dst = dbrow + key0;
                if (!tail) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float4 v = make_float4(ds[4 * j], ds[4 * j + 1], ds[4 * j + 2], ds[4 * j + 3]);
                        if (p.accumulate) {
                            const float4 old = reinterpret_cast<const float4*>(dst)[j];
                            v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                        }
                        reinterpret_cast<float4*>(dst)[j] = v;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (key0 + 16 * hi + i < T) dst[i] = p.accumulate ? dst[i] + ds[i] : ds[i];
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float dv8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) dv8[j] = ds[8 * s2 + j];
                const bf16x8 db = pack8(dv8);
                // bf16 dBias: the very values the dQ / dK contractions use, written once (no read-modify-write; the
                // layers' slices are summed by the consumer).  Parked in a wave-private LDS tile here and written out
                // by `flush_dbias` as whole 128-byte rows per chunk.
                if (db16) *reinterpret_cast<bf16x8*>(&dSs[wave][n][t * 32 + 16 * hi + 8 * s2]) = db;
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Kt[kappa(n)][t * 32 + 16 * hi + 8 * s2]);
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, db, dq, 0, 0, 0);
            }
        }
        flush_dbias(c);
    };
    for (int c = 0; c < nchunk; c += 2) {
        chunk(c, ring[0]);
        if (PIPE && c + 1 < nchunk) chunk(c + 1, ring[1]);
    }

    if (q_ok) {
        TQ* DQ = reinterpret_cast<TQ*>(p.dq) + ((int64_t)g * T + my_q) * p.lddq + h * D + 16 * hi;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (16 * hi + 8 * j < D) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = dq[8 * j + i] * p.scale;
                store8(DQ + 8 * j, v);
            }
        }
    }
}

template <int D, typename TQ, typename TB, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const AttnParams p) {
    attn_bwd_dq_body<D, TQ, TB, NW, DROP>(p, blockIdx.x, gridDim.x);
}

// ======================================================================= backward, pass 2: dK + dV
// One wave = 32 keys (on the lanes), sweep over queries; reads the TRANSPOSED bias so that a lane's
// 16 accumulator registers are again 16 contiguous elements (queries q0+16*hi .. +15 of its key row).
// Row statistics (header: consistent softmax): per query of the chunk M' (log2 domain), 1 / l' and delta = dO_b . O with the bf16 dO
// values this pass stages and O in full precision -- formed HERE, by the thread that carries the row (the dQ pass folds 1 / l' into
// its dO fragment before rounding, so its delta belongs to slightly different dP values; each pass is consistent in itself).
template <int D, typename TQ, typename TB, int NW, bool DROP>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnParams& p, const int bid, const int nwg) {
    constexpr int KS = (D + 15) / 16;
    constexpr int NT = NW * 64;
    __shared__ __attribute__((aligned(16))) bf16_t Qs[KC][ROWP];
    __shared__ __attribute__((aligned(16))) bf16_t dOs[KC][ROWP];
    __shared__ __attribute__((aligned(16))) bf16_t Qt[32][COLP];
    __shared__ __attribute__((aligned(16))) bf16_t dOt[32][COLP];
    __shared__ __attribute__((aligned(16))) float lseS[KC];          // M' of the chunk's queries
    __shared__ __attribute__((aligned(16))) float ilS[KC];           // 1 / l'
    __shared__ __attribute__((aligned(16))) float dlS[KC];           // delta
    // dropout rule v2 (common.h): the w words of the chunk's 2 tiles x this workgroup's 2*NW key blocks x 8 key pairs x
    // 32 query rows, built cooperatively while the chunk is staged (one block hash + 8 mads per entry, 2 entries per thread)
    // (rows padded to 36 words: at 32 every row starts on bank 0 or 32 and a lane group's 16 rows collide 8 ways)
    __shared__ __attribute__((aligned(16))) uint32_t dropW[DROP ? 2 : 1][DROP ? NW * 2 : 1][DROP ? 8 : 1][DROP ? 36 : 4];

    const int T = p.T, H = p.H;
    const int nK = p.nq;
    const int lid = xcd_remap(bid, nwg);
    const int kt = lid % nK, gh = lid / nK;
    const int g = gh / H, h = gh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const int k0w = wave_row0(kt, nK, T, __builtin_amdgcn_readfirstlane(wave), NW);   // this wave's first key row of bias_t (an SGPR)
    const int my_k = k0w + n;
    const bool k_ok = my_k < T;
    const int kc = k_ok ? my_k : T - 1;

    const TQ* Q = reinterpret_cast<const TQ*>(p.q) + (int64_t)g * T * p.ldq + h * D;
    const TQ* K = reinterpret_cast<const TQ*>(p.k) + (int64_t)g * T * p.ldk + h * D;
    const TQ* V = reinterpret_cast<const TQ*>(p.v) + (int64_t)g * T * p.ldv + h * D;
    const TQ* dO = reinterpret_cast<const TQ*>(p.dout) + (int64_t)g * T * p.ldo + h * D;
    __shared__ __attribute__((aligned(16))) unsigned char Bs[NW][BiasStage<TB>::BYTES];
    const TB* brows = reinterpret_cast<const TB*>(p.bias_t) + ((int64_t)gh * T + min(k0w, T - 1)) * p.ld_bias;
    const int brow_max = max(T - 1 - k0w, 0);
    unsigned char* bimg = Bs[wave];

    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bool ok = k_ok && (ks * 16 + 8 * hi < D);
        // the softmax scale rides on this lane's K fragment (S = Q . (scale K)); Q is then staged as it is -- a plain
        // 16-byte copy for bf16 input -- and dK = scale . dS^T Q gets the factor once, at the end
        load_frag(K + (int64_t)kc * p.ldk + ks * 16 + 8 * hi, ok, p.scale, kf[ks]);
        // (with dropout, 1/(1-p) of dP rides on this lane's V fragment)
        load_frag(V + (int64_t)kc * p.ldv + ks * 16 + 8 * hi, ok, DROP ? p.inv_keep : 1.f, vf[ks]);
    }
    uint64_t seed = 0;
    if (DROP) seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
    const int drop_sh = (n & 1) ? 0 : 16, thr_hi = p.thr_s * 65536;

    f32x16 dk, dv;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[i] = 0.f; dv[i] = 0.f; }

    const int nchunk = (T + KC - 1) / KC;
    constexpr bool PIPE = NW == 4;
    // one super-tile (a chunk's 32 keys x 64 queries of bias_t) in registers, requested one chunk ahead (this pass is
    // compute-bound and short of registers: a deeper ring cost 26 VGPRs and bought nothing)
    BiasStage<TB, PIPE> bst;
    bst.init(p.ld_bias, brow_max, lane);
    bst.load(brows, 0);
    Slab<D, TQ, PIPE ? NT : KC * 4> qreg, doreg;
    if (PIPE) {
        qreg.load_clamped(Q, p.ldq, 0, T);
        doreg.load_clamped(dO, p.ldo, 0, T);
    }
    float lse_r = 0.f, dl_r = 0.f;                         // thread it < KC carries query it of the chunk (NT >= KC)
    auto load_rowstats = [&](const int c) {
        // (every thread loads, from a clamped row: no branch around the loads; rows >= T have P = 0, any finite value serves)
        const int q = min(c * KC + (int)(threadIdx.x & (KC - 1)), T - 1);
        lse_r = p.lse_in[(int64_t)gh * T + q];
        const TQ* O = reinterpret_cast<const TQ*>(p.out) + (int64_t)g * T * p.ldo + h * D;
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < D; e += 8) {
            float a[8], b[8];
            load8(dO + (int64_t)q * p.ldo + e, a);
            load8(O + (int64_t)q * p.ldo + e, b);
            if constexpr (std::is_same<TQ, bf16_t>::value) {
                if (p.out_lo) {
                    float lo[8];
                    load8(reinterpret_cast<const bf16_t*>(p.out_lo) + ((int64_t)g * T + q) * p.ldo + h * D + e, lo);
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[j] += lo[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) d = fmaf((float)(bf16_t)a[j], b[j], d);       // the bf16 dO the dP product multiplies
        }
        dl_r = d;
    };
    load_rowstats(0);
    auto chunk = [&](const int c) {
        bst.park(bimg, lane);                                                  // (see the forward kernel)
        if (PIPE) bst.load(brows, min(c + 1, nchunk - 1));                    // (never behind a branch: see the forward)
        __syncthreads();
        if (PIPE) {
            qreg.template store<true, true, true>(1.f, Qs, Qt);
            doreg.template store<true, true, true>(1.f, dOs, dOt);
        } else {
            Slab<D, TQ, NT>::template direct<true, true, true>(Q, p.ldq, c * KC, T, 1.f, Qs, Qt);
            Slab<D, TQ, NT>::template direct<true, true, true>(dO, p.ldo, c * KC, T, 1.f, dOs, dOt);
        }
        if (threadIdx.x < KC) {
            float mq, il;
            row_norm(lse_r, mq, il);
            lseS[threadIdx.x] = mq;
            ilS[threadIdx.x] = il;
            dlS[threadIdx.x] = dl_r;
        }
        if (DROP) {
            for (int e = threadIdx.x; e < 2 * NW * 2 * 32; e += NT) {
                const int row = e & 31, kbl = (e >> 5) % (NW * 2), t = e / (NW * 64);
                const int q = c * KC + t * 32 + row;
                const uint32_t rh = dropout_row_hash(seed, (uint32_t)(gh * T + (q < T ? q : T - 1)));
                const uint32_t hb = attn_drop_block(seed, rh, (uint32_t)(wg_tile0(kt, nK, T) * 2 + kbl));
#pragma unroll
                for (int m = 0; m < 8; ++m) dropW[t][kbl][m][row] = attn_drop_word(hb, attn_drop_mult(m));
            }
        }
        __syncthreads();
        if (PIPE) {                                                            // (unconditional, clamped: see the forward)
            qreg.load_clamped(Q, p.ldq, min(c + 1, nchunk - 1) * KC, T);
            doreg.load_clamped(dO, p.ldo, min(c + 1, nchunk - 1) * KC, T);
        }
        load_rowstats(min(c + 1, nchunk - 1));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q0 = c * KC + t * 32;
            if (q0 >= T) break;
            f32x16 s, dp;
            BiasStage<TB>::to_acc(bimg, n, hi, t, s);
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 aq = *reinterpret_cast<const bf16x8*>(&Qs[t * 32 + kappa(n)][ks * 16 + 8 * hi]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[ks], s, 0, 0, 0);
                const bf16x8 ad = *reinterpret_cast<const bf16x8*>(&dOs[t * 32 + kappa(n)][ks * 16 + 8 * hi]);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, vf[ks], dp, 0, 0, 0);
            }
            // (query rows >= T of the last tile need no masking: their bias_t columns are -inf -- both pack kernels write
            // the padding so -- hence P = 0 and dS = 0 there; keys >= T live in lanes whose columns are never stored)
            // eight query rows at a time (row statistics, dropout words, P and dS of a half are dead before the next
            // half starts: ~50 fewer live registers than all sixteen at once)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float lse8[8], il8[8], dl8[8];
                uint32_t w8[8];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r0 = t * 32 + 16 * hi + 8 * s2 + 4 * j;
                    const float4 a = *reinterpret_cast<const float4*>(&lseS[r0]);
                    const float4 b = *reinterpret_cast<const float4*>(&dlS[r0]);
                    const float4 e = *reinterpret_cast<const float4*>(&ilS[r0]);
                    lse8[4 * j] = a.x; lse8[4 * j + 1] = a.y; lse8[4 * j + 2] = a.z; lse8[4 * j + 3] = a.w;
                    dl8[4 * j] = b.x; dl8[4 * j + 1] = b.y; dl8[4 * j + 2] = b.z; dl8[4 * j + 3] = b.w;
                    il8[4 * j] = e.x; il8[4 * j + 1] = e.y; il8[4 * j + 2] = e.z; il8[4 * j + 3] = e.w;
                    if (DROP) {     // this key's pair word for each query row (dropout rule v2, common.h)
                        const uint4 w = *reinterpret_cast<const uint4*>(
                            &dropW[t][wave * 2 + (n >> 4)][(n & 15) >> 1][16 * hi + 8 * s2 + 4 * j]);
                        w8[4 * j] = w.x; w8[4 * j + 1] = w.y; w8[4 * j + 2] = w.z; w8[4 * j + 3] = w.w;
                    }
                }
                float a8[8], b8[8];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const int i = 8 * s2 + j;
                    // the forward's probabilities of this key for two query rows (header: consistent softmax), rounded as a pair
                    const uint32_t pw = pack2(fast_exp2(fmaf(s[i], MOBGT_LOG2E, -lse8[j])), fast_exp2(fmaf(s[i + 1], MOBGT_LOG2E, -lse8[j + 1])));
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float pr = u ? bf16_hi(pw) : bf16_lo(pw);
                        // dS = (1 / l') P_b (M dP / (1-p) - delta) = (1 / l') (X dP' - P_b delta) with X = M P_b (what dV sums) and
                        // dP' = dO (V / (1-p)); even key: low half of w, moved to the top by the lane's shift; odd key: high half
                        const bool keep = !DROP || (int)(w8[j + u] << drop_sh) >= thr_hi;
                        const float x = keep ? pr : 0.f;           // 1/(1-p) of dV is applied once, at the end
                        a8[j + u] = x * il8[j + u];
                        b8[j + u] = il8[j + u] * fmaf(x, dp[i + u], -pr * dl8[j + u]);
                    }
                }
                const bf16x8 pb = pack8(a8), db = pack8(b8);
                const bf16x8 ado = *reinterpret_cast<const bf16x8*>(&dOt[kappa(n)][t * 32 + 16 * hi + 8 * s2]);
                dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ado, pb, dv, 0, 0, 0);
                const bf16x8 aq = *reinterpret_cast<const bf16x8*>(&Qt[kappa(n)][t * 32 + 16 * hi + 8 * s2]);
                dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, db, dk, 0, 0, 0);
            }
        }
    };
    for (int c = 0; c < nchunk; ++c) chunk(c);

    if (k_ok) {
        TQ* DK = reinterpret_cast<TQ*>(p.dk) + ((int64_t)g * T + my_k) * p.lddk + h * D + 16 * hi;
        TQ* DV = reinterpret_cast<TQ*>(p.dv) + ((int64_t)g * T + my_k) * p.lddv + h * D + 16 * hi;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (16 * hi + 8 * j < D) {
                float a[8], b[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { a[i] = dk[8 * j + i] * p.scale; b[i] = DROP ? dv[8 * j + i] * p.inv_keep : dv[8 * j + i]; }
                store8(DK + 8 * j, a);
                store8(DV + 8 * j, b);
            }
        }
    }
}

// (tried: amdgpu_waves_per_eu(3), i.e. <= 168 VGPRs instead of 184 -- 19 dwords of scratch per lane in the loop, 103 -> 123 us)
template <int D, typename TQ, typename TB, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const AttnParams p) {
    attn_bwd_dkv_body<D, TQ, TB, NW, DROP>(p, blockIdx.x, gridDim.x);
}

// Both backward passes in one launch (small graphs: each pass alone is 128-256 workgroups of 1-2 waves -- a fraction of
// the chip -- and nothing but latency; side by side they take the time of the longer one): the first n_first workgroups
// run the dQ / dBias pass, the rest the dK / dV pass.
template <int D, typename TQ, typename TB, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_bwd_both_kernel(const AttnParams p) {
    if ((int)blockIdx.x < p.n_first) attn_bwd_dq_body<D, TQ, TB, NW, DROP>(p, blockIdx.x, p.n_first);
    else attn_bwd_dkv_body<D, TQ, TB, NW, DROP>(p, blockIdx.x - p.n_first, gridDim.x - p.n_first);
}


// ======================================================================= backward in ONE pass (round 4; T > 64, bf16 I/O + bf16 bias)
// The two passes above form P and dS twice and stream the bias twice (row-major for dQ, transposed for dK / dV).  Here a
// workgroup of EIGHT waves owns up to 256 keys of one (graph, head) -- keys on the lanes, as in the dK / dV pass: S^T, dP^T, P
// and dS exist ONCE per (query, key) --, sweeps the queries in chunks of 64 and finishes all four gradients:
//   * dK^T / dV^T accumulate in registers over the sweep (no sum across workgroups), as before;
//   * the dS^T tiles (bf16: the very values the dK product multiplies) are parked in an LDS image [256 keys][64 queries]; behind
//     ONE more barrier per chunk that image is read back TRANSPOSED (ds_read_b64_tr_b16: 4 keys x 16 queries per 16-lane group)
//       - as 16-byte pieces dBias[query][key .. key + 7] of this layer's row-major slice (what the dQ pass wrote), and
//       - as the A operand of dQ[16 queries x 16 head columns] += dS[16 x 32 keys] K[32 keys x 16] (v_mfma_f32_16x16x32_bf16; K^T of
//         the workgroup's keys is staged once): each wave owns ONE 16 x 16 tile of the chunk's 64 x 32 and sums it over all
//         256 keys, so a chunk leaves the workgroup as 8 KB of f32 atomic adds into a zero-filled accumulator (ceil(T / 256) adds
//         per element: 54 MB per launch at c5, where 128-key workgroups would add 90 MB);
//   * rowsum(dO~ O) is formed by the threads that stage dO (they request the same pieces of O and its bf16 residual with the same
//     two-chunk lead); the f32 dQ accumulator is zero on entry and attn_dq_finish_kernel zeroes it again behind its read.
// Bias traffic: bias_t read once + dBias written once (2 x 158 MB at c5; the two passes move 3 x 158 MB); exp, dropout, dS once.
typedef short mobgt_v4s __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mobgt_v4s lds_tr16(const bf16_t* p) {
    // per 16-lane group: lane 4q + p supplies the address of row q, columns 4p .. 4p + 3 of a 4 x 16 block; lane i receives
    // column i of the 4 rows (cdna_hip_programming.md T10).  EXEC must be all ones at every call.
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((mobgt_v4s __attribute__((address_space(3)))*)(p));
}
// value of lane (i ^ 4): row_half_mirror (i -> 7 - i within 8 lanes) then the quad reversed (two DPP moves, no LDS round trip)
__device__ __forceinline__ uint32_t lane_xor4(uint32_t x) {
    const int y = __builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(0, y, 0x1B, 0xf, 0xf, false);             // quad_perm [3,2,1,0]
}
#ifndef ONE_TR_B16
#define ONE_TR_B16 0          // timing probe only (tools/attn_ab.sh): 1 = the transposed staging copies as eight 2-byte stores (rounds 4-5)
#endif
#ifndef ONE_SKIP
#define ONE_SKIP 0            // timing probes only (tools/attn_ab.sh): 1 = no dBias / dQ section, 2 = no atomics, 3 = no dBias stores, 4 = no DSk parking either
#endif
constexpr int ONE_NW = 8;                 // waves per workgroup
constexpr int DSP = KC + 8;               // pitch (bf16) of the parked dS^T image: 144 B, an odd multiple of 16 B
constexpr int KTP = 32 * ONE_NW + 8;      // pitch (bf16) of the K^T image: 528 B

template <int D, bool DROP>
__global__ __launch_bounds__(ONE_NW * 64) void attn_bwd_one_kernel(const AttnParams p) {
    typedef bf16_t TQ;
    typedef bf16_t TB;
    constexpr int NW = ONE_NW, NT = NW * 64, KS = (D + 15) / 16;
    __shared__ __attribute__((aligned(16))) bf16_t Qs[KC][ROWP];
    __shared__ __attribute__((aligned(16))) bf16_t dOs[KC][ROWP];
    __shared__ __attribute__((aligned(16))) bf16_t Qt[32][COLP];
    __shared__ __attribute__((aligned(16))) bf16_t dOt[32][COLP];
    __shared__ __attribute__((aligned(16))) float lseS[KC];
    __shared__ __attribute__((aligned(16))) float dlS[KC];
    __shared__ __attribute__((aligned(16))) uint32_t dropW[DROP ? 2 : 1][DROP ? NW * 2 : 1][DROP ? 8 : 1][DROP ? 36 : 4];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[NW][BiasStage<TB>::BYTES];
    __shared__ __attribute__((aligned(16))) bf16_t DSk[32 * NW][DSP];          // dS^T of the chunk: [key][query]
    __shared__ __attribute__((aligned(16))) bf16_t KT[32][KTP];               // K^T of the workgroup's keys: [head column][key]

    const int T = p.T, H = p.H;
    const int nK = p.nq;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int kt = lid % nK, gh = lid / nK;
    const int g = gh / H, h = gh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hi = lane >> 5;
    const int k0g = wg_tile0(kt, nK, T) * 32;                                  // first key of the workgroup
    const int k0w = wave_row0(kt, nK, T, __builtin_amdgcn_readfirstlane(wave), NW);
    const int my_k = k0w + n;
    const bool k_ok = my_k < T;
    const int kc = k_ok ? my_k : T - 1;

    const TQ* Q = reinterpret_cast<const TQ*>(p.q) + (int64_t)g * T * p.ldq + h * D;
    const TQ* K = reinterpret_cast<const TQ*>(p.k) + (int64_t)g * T * p.ldk + h * D;
    const TQ* V = reinterpret_cast<const TQ*>(p.v) + (int64_t)g * T * p.ldv + h * D;
    const TQ* dO = reinterpret_cast<const TQ*>(p.dout) + (int64_t)g * T * p.ldo + h * D;
    const TQ* Oo = reinterpret_cast<const TQ*>(p.out) + (int64_t)g * T * p.ldo + h * D;
    const TB* brows = reinterpret_cast<const TB*>(p.bias_t) + ((int64_t)gh * T + min(k0w, T - 1)) * p.ld_bias;
    const int brow_max = max(T - 1 - k0w, 0);
    unsigned char* bimg = Bs[wave];

    // ---- K^T of the workgroup's keys -> LDS (once): piece e = key e / 4, head columns 8 (e % 4) .. + 7
    for (int e = tid; e < 32 * NW * 4; e += NT) {
        const int kk = e >> 2, c0 = (e & 3) * 8;
        Raw8<TQ> x;
        if (c0 < D && k0g + kk < T) x.load(K + (int64_t)(k0g + kk) * p.ldk + c0);
        else x.zero();
        const bf16x8 b = x.as_bf16();
#pragma unroll
        for (int i = 0; i < 8; ++i) KT[c0 + i][kk] = b[i];
    }
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bool ok = k_ok && (ks * 16 + 8 * hi < D);
        load_frag(K + (int64_t)kc * p.ldk + ks * 16 + 8 * hi, ok, p.scale, kf[ks]);
        load_frag(V + (int64_t)kc * p.ldv + ks * 16 + 8 * hi, ok, DROP ? p.inv_keep : 1.f, vf[ks]);
    }
    uint64_t seed = 0;
    if (DROP) seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
    const int drop_sh = (n & 1) ? 0 : 16, thr_hi = p.thr_s * 65536;
    const uint32_t kmask = k_ok ? 0xffffffffu : 0u;             // keys beyond T: their dS must be exact zeros (dBias columns, dQ)

    f32x16 dk, dv;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[i] = 0.f; dv[i] = 0.f; }

    const int nchunk = (T + KC - 1) / KC;
    BiasStage<TB, true> bst;
    bst.init(p.ld_bias, brow_max, lane);
    // staging of a chunk: threads 0 .. 255 carry one 16-byte piece of Q each, threads 256 .. 511 one of dO
    const bool is_q = tid < 256;
    const int se = tid & 255, sr = se >> 2, sc0 = (se & 3) * 8;
    const TQ* ssrc = is_q ? Q : dO;
    const int64_t sld = is_q ? p.ldq : p.ldo;
    bf16_t (*srm)[ROWP] = is_q ? Qs : dOs;
    bf16_t (*strn)[COLP] = is_q ? Qt : dOt;
    Raw8<TQ> sreg, oreg, lreg;
    float lse_s = 0.f;                                          // (dO threads) lse of the row their piece belongs to
    const bf16_t* OLo = p.out_lo ? reinterpret_cast<const bf16_t*>(p.out_lo) + (int64_t)g * T * p.ldo + h * D : nullptr;
    // Consistent softmax (header): a dO row is staged as dO~ = bf16(dO / l') -- the normalisation of the row's probabilities rides
    // on it, so the tile loop works with the forward's un-normalised P_b = bf16(2^(x - M')) and pays nothing per element for
    // 1 / l' -- and delta = dO~ . (O + O_lo) is formed from the very values that were staged.
    // rowsum(dO~ O) is formed HERE: the threads that stage a piece of dO also request the same piece of O (and of its residual)
    // with the same two-chunk lead and reduce the products over the four pieces of a row when the chunk is stored (a launch in
    // front that read dO and O once more and zero-filled the accumulator cost 10.9 us at c5: round 4).
    // (no branch in here -- the loop's wait counts are exact only over straight-line code: a piece beyond the head dimension is
    //  read from column 0 and zeroed when it is stored; without a residual the O piece is read twice and the copy weighted 0)
    const bool col_ok = D % 32 == 0 || sc0 < D;
    const int scl = col_ok ? sc0 : 0;
    const TQ* osrc = is_q ? ssrc : Oo;
    const TQ* lsrc = is_q ? ssrc : (OLo ? reinterpret_cast<const TQ*>(OLo) : Oo);
    const float lw = OLo ? 1.f : 0.f;
    auto stage_load = [&](const int c) {
        const int64_t ro = (int64_t)min(c * KC + sr, T - 1) * sld + scl;
        sreg.load(ssrc + ro);
        oreg.load(osrc + ro);                            // (Q threads: a second read of their piece, unused)
        lreg.load(lsrc + ro);
        lse_s = p.lse_in[(int64_t)gh * T + min(c * KC + sr, T - 1)];
    };
    // ---- the chunk loop, pipelined by HALVES of a chunk (round 5) ----------------------------------------------------------
    // Round 4 ran a chunk as   tiles(c) | barrier | staging of c + 1, dBias / dQ write-out of c | barrier   : the write-out (26 us
    // of dBias stores, 11 of dQ products, 9 of atomics at c5) sat between two barriers with nothing beside it.  A chunk's two
    // 32-query tiles use disjoint halves of every staging image and of the dS image, so a PHASE is now one tile, and phase p
    // carries, beside its tile, the write-out of phase p - 1's half of dS and the staging of the half that phase p + 1 reads:
    //     phase (c, 0):  tile (c, 0) | write-out of (c - 1, 1) | stage rows 32..63 of chunk c,     dropout words of tile (c, 1)
    //     phase (c, 1):  tile (c, 1) | write-out of (c, 0)     | stage rows 0..31 of chunk c + 1,  dropout words of tile (c + 1, 0)
    // one barrier per phase -- as many as before -- and the write-out's stores and atomics are in flight under the next tile.
    // A wave's staging pieces all belong to one half (waves 0-1 / 4-5: rows 0..31 of Q / dO, waves 2-3 / 6-7: rows 32..63), so
    // "stage" is wave-uniform; its registers are requested one chunk ahead, right behind the store that frees them.  The bias
    // image is wave-private: a wave parks chunk c + 1's super-tile behind tile (c, 1), its last read of chunk c.
    const int my_half = (wave >> 1) & 1;
    auto stage_store = [&](const int c) {
        (void)c;
        bf16x8 b = sreg.as_bf16();
        float mq = 0.f, il = 1.f;
        if (!is_q) {                                     // (wave-uniform: waves 4 .. 7)
            float a8[8];
            row_norm(lse_s, mq, il);
            sreg.get(a8);
#pragma unroll
            for (int i = 0; i < 8; ++i) a8[i] *= il;
            b = pack8(a8);
        }
        if (D % 32 != 0) {
            u32x4 w = __builtin_bit_cast(u32x4, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = col_ok ? w[i] : 0u;
            b = __builtin_bit_cast(bf16x8, w);
        }
        *reinterpret_cast<bf16x8*>(&srm[sr][sc0]) = b;
#if ONE_TR_B16
#pragma unroll
        for (int i = 0; i < 8; ++i) strn[sc0 + i][sr] = b[i];
#else
        {
            // the transposed copy, two rows at a time (round 6): the lanes of rows 2i and 2i + 1 of one piece (4 lanes apart) swap half
            // of their piece through two DPP moves per dword, so that the even row's lane holds head columns 0..3 of BOTH rows and
            // the odd row's lane columns 4..7 -- four 4-byte stores [column][2i, 2i + 1] per lane where eight 2-byte stores
            // [column][row] went (two lanes writing the two halves of one dword: 9 LDS writes per piece -> 5).
            const u32x4 w = __builtin_bit_cast(u32x4, b);
            const bool odd = sr & 1;
            const uint32_t s0 = odd ? w[0] : w[2], s1 = odd ? w[1] : w[3];
            const uint32_t k0 = odd ? w[2] : w[0], k1 = odd ? w[3] : w[1];
            const uint32_t r0 = lane_xor4(s0), r1 = lane_xor4(s1);
            const uint32_t lo0 = odd ? r0 : k0, hi0 = odd ? k0 : r0, lo1 = odd ? r1 : k1, hi1 = odd ? k1 : r1;
            const int cb = sc0 + (odd ? 4 : 0), rp = sr & ~1;
            *reinterpret_cast<uint32_t*>(&strn[cb + 0][rp]) = __builtin_amdgcn_perm(hi0, lo0, 0x05040100u);
            *reinterpret_cast<uint32_t*>(&strn[cb + 1][rp]) = __builtin_amdgcn_perm(hi0, lo0, 0x07060302u);
            *reinterpret_cast<uint32_t*>(&strn[cb + 2][rp]) = __builtin_amdgcn_perm(hi1, lo1, 0x05040100u);
            *reinterpret_cast<uint32_t*>(&strn[cb + 3][rp]) = __builtin_amdgcn_perm(hi1, lo1, 0x07060302u);
        }
#endif
        {
            float o8[8], l8[8];
            oreg.get(o8);
            lreg.get(l8);
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d = fmaf((float)b[i], fmaf(lw, l8[i], o8[i]), d);
            d += __shfl_xor(d, 1, 64);                   // the four pieces of a row sit in four neighbouring lanes
            d += __shfl_xor(d, 2, 64);
            if (!is_q && (se & 3) == 0) {
                dlS[sr] = d;
                lseS[sr] = mq;                           // M' of the row (row_norm)
            }
        }
    };
    auto drop_build = [&](const int c, const int t) {   // the w words of tile (c, t): 16 key blocks x 32 query rows = one entry per thread
        const int row = tid & 31, kbl = tid >> 5;
        const int q = c * KC + t * 32 + row;
        const uint32_t rh = dropout_row_hash(seed, (uint32_t)(gh * T + (q < T ? q : T - 1)));
        const uint32_t hb = attn_drop_block(seed, rh, (uint32_t)(wg_tile0(kt, nK, T) * 2 + kbl));
#pragma unroll
        for (int m = 0; m < 8; ++m) dropW[t][kbl][m][row] = attn_drop_word(hb, attn_drop_mult(m));
    };
    auto tile = [&](const int c, const int t) {
        const int q0 = c * KC + t * 32;
        if (q0 >= T) return;               // (wave-uniform.  The tile's half of DSk keeps older values: they only reach dBias rows /
                                           //  dQ rows >= T, which are never stored)
        f32x16 s, dp;
        BiasStage<TB>::to_acc(bimg, n, hi, t, s);
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 aq = *reinterpret_cast<const bf16x8*>(&Qs[t * 32 + kappa(n)][ks * 16 + 8 * hi]);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[ks], s, 0, 0, 0);
            const bf16x8 ad = *reinterpret_cast<const bf16x8*>(&dOs[t * 32 + kappa(n)][ks * 16 + 8 * hi]);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, vf[ks], dp, 0, 0, 0);
        }
        // (query rows >= T of the last tile need no masking: their bias_t columns are -inf, hence P = 0 and dS = 0 there)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float lse8[8], dl8[8];
            uint32_t w8[8];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r0 = t * 32 + 16 * hi + 8 * s2 + 4 * j;
                const float4 a = *reinterpret_cast<const float4*>(&lseS[r0]);
                const float4 b = *reinterpret_cast<const float4*>(&dlS[r0]);
                lse8[4 * j] = a.x; lse8[4 * j + 1] = a.y; lse8[4 * j + 2] = a.z; lse8[4 * j + 3] = a.w;
                dl8[4 * j] = b.x; dl8[4 * j + 1] = b.y; dl8[4 * j + 2] = b.z; dl8[4 * j + 3] = b.w;
                if (DROP) {
                    const uint4 w = *reinterpret_cast<const uint4*>(
                        &dropW[t][wave * 2 + (n >> 4)][(n & 15) >> 1][16 * hi + 8 * s2 + 4 * j]);
                    w8[4 * j] = w.x; w8[4 * j + 1] = w.y; w8[4 * j + 2] = w.z; w8[4 * j + 3] = w.w;
                }
            }
            float a8[8], b8[8];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int i = 8 * s2 + j;
                // the forward's probabilities of this key for two query rows, rounded to bf16 as a pair (header: consistent softmax)
                const uint32_t pw = pack2(fast_exp2(fmaf(s[i], MOBGT_LOG2E, -lse8[j])), fast_exp2(fmaf(s[i + 1], MOBGT_LOG2E, -lse8[j + 1])));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float pr = u ? bf16_hi(pw) : bf16_lo(pw);
                    if (DROP) {
                        const bool keep = (int)(w8[j + u] << drop_sh) >= thr_hi;
                        const float x = keep ? pr : 0.f;
                        a8[j + u] = x;
                        b8[j + u] = fmaf(x, dp[i + u], -pr * dl8[j + u]);
                    } else {
                        a8[j + u] = pr;
                        b8[j + u] = pr * (dp[i + u] - dl8[j + u]);
                    }
                }
            }
            const bf16x8 pb = pack8(a8);
            u32x4 dbw = __builtin_bit_cast(u32x4, pack8(b8));
#pragma unroll
            for (int k = 0; k < 4; ++k) dbw[k] &= kmask;
            const bf16x8 db = __builtin_bit_cast(bf16x8, dbw);
            if (ONE_SKIP != 4) *reinterpret_cast<bf16x8*>(&DSk[32 * wave + n][t * 32 + 16 * hi + 8 * s2]) = db;
            const bf16x8 ado = *reinterpret_cast<const bf16x8*>(&dOt[kappa(n)][t * 32 + 16 * hi + 8 * s2]);
            dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ado, pb, dv, 0, 0, 0);
            const bf16x8 aq = *reinterpret_cast<const bf16x8*>(&Qt[kappa(n)][t * 32 + 16 * hi + 8 * s2]);
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, db, dk, 0, 0, 0);
        }
    };
    // dBias and dQ of the half (c, t) of the dS image
    // (descriptors of this (graph, head)'s dBias slice and of the graph's dQ accumulator: both far below 2 GB)
    const __amdgpu_buffer_rsrc_t db_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<bf16_t*>(p.dbias) + (p.dbias ? (int64_t)gh * T * p.ld_bias : 0), 0, p.dbias ? (int)(T * p.ld_bias * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t dq_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.dq_acc + (int64_t)g * T * (H * D), 0, T * H * D * 4, 0x00020000);
    // do_dq: this wave carries one of the half's four dQ tiles (compile-time at every call site).  Nothing in here branches: a
    // lane (or a whole wave: half (-1, 1) of the first phase, halves beyond T) that has nothing to write carries an offset beyond
    // the buffer and the hardware drops it, so the wait counts of the staging loads issued in FRONT of the write-out are exact
    // and leave these stores and atomics in flight under the next tile.
    auto writeout = [&](const int c, const int t, const bool do_dq) {
        if (ONE_SKIP == 1 || ONE_SKIP == 4) return;
        const int i16 = lane & 15, g4 = lane >> 4, q4 = i16 >> 2, p4 = i16 & 3;
        // ---- dQ: the half's 32 queries x 32 head columns are four 16 x 16 tiles over all 256 keys, on the four waves that do
        //      not stage in this phase: queries 16 (w >> 2) .., head columns 16 (w & 1) ..
        if (do_dq) {
            const int qh = wave >> 2, dh = wave & 1;
            f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NW; ++ks) {
                const int kl = 32 * ks + 8 * g4;
                const mobgt_v4s r0 = lds_tr16(&DSk[kl + q4][t * 32 + qh * 16 + 4 * p4]);
                const mobgt_v4s r1 = lds_tr16(&DSk[kl + 4 + q4][t * 32 + qh * 16 + 4 * p4]);
                typedef short v8s __attribute__((ext_vector_type(8)));
                const v8s av = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
                const bf16x8 a = __builtin_bit_cast(bf16x8, av);
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(&KT[dh * 16 + i16][kl]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            }
            const int dcol = dh * 16 + i16;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int qglob = c * KC + t * 32 + qh * 16 + 4 * g4 + v;
                const uint32_t off = (dcol < D && (unsigned)qglob < (unsigned)T) ? (uint32_t)(qglob * (H * D) + h * D + dcol) * 4u : 0x80000000u;
                if (ONE_SKIP != 2) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[v], dq_rsrc, off, 0, 0);
                if (ONE_SKIP == 2 && acc[v] == 12345.678f) p.dq_acc[0] = acc[v];          // (keeps the products alive)
            }
        }
        // ---- dBias: wave w = 16 queries (w & 1) x 64 keys (w >> 1); a lane's piece j: query i16, keys 32 j + 8 g4 .. + 7
        if (ONE_SKIP != 3) {
            // (only this workgroup's own key columns: idle waves of a workgroup with fewer than 8 key tiles parked zeros at
            //  rows that are the NEXT workgroup's keys.  No dBias wanted: the descriptor is empty)
            const int kend = kt + 1 < nK ? wg_tile0(kt + 1, nK, T) * 32 : (int)p.ld_bias;
            const int qg = wave & 1, kq = wave >> 1;
            const int qglob = c * KC + 32 * t + 16 * qg + i16;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kl = 64 * kq + 32 * j + 8 * g4;
                const mobgt_v4s r0 = lds_tr16(&DSk[kl + q4][32 * t + 16 * qg + 4 * p4]);
                const mobgt_v4s r1 = lds_tr16(&DSk[kl + 4 + q4][32 * t + 16 * qg + 4 * p4]);
                const int kglob = k0g + kl;
                typedef short v8s __attribute__((ext_vector_type(8)));
                const v8s v = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
                const uint32_t off = ((unsigned)qglob < (unsigned)T && kglob < kend) ? (uint32_t)(qglob * (int)p.ld_bias + kglob) * 2u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), db_rsrc, off, 0, DB_AUX);
            }
        }
    };
    // prologue: chunk 0 staged in full; the waves of rows 0 .. 31 request chunk 1, the others chunk 0 once more (they stage
    // their rows in phase (c, 0), also for c = 0: the loop body has no first-iteration case)
    stage_load(0);
    bst.load(brows, 0);
    stage_store(0);
    if (DROP) {
        drop_build(0, 0);
        drop_build(0, 1);
    }
    bst.park(bimg, lane);
    stage_load(my_half ? 0 : min(1, nchunk - 1));
    bst.load(brows, min(1, nchunk - 1));
    __syncthreads();

    // The loop is instantiated per staging half, and nothing in its body is conditional at run time (chunk numbers are clamped;
    // what is staged or parked beyond the last chunk is never read): the compiler's wait counts are then exact, every wait on a
    // staging / bias load leaves the write-out's stores and atomics of the phases behind it in flight.
    auto phases = [&](auto mh_c) {
        constexpr int MH = decltype(mh_c)::value;
        for (int c = 0; c < nchunk; ++c) {
            // ---- phase (c, 0)
            tile(c, 0);
            if (MH == 1) {                                      // rows 32 .. 63 of chunk c: read by the next phase
                stage_store(c);
                stage_load(min(c + 1, nchunk - 1));
            }
            if (DROP) drop_build(c, 1);
            writeout(c - 1, 1, MH == 0);
            __syncthreads();
            // ---- phase (c, 1)
            tile(c, 1);
            if (MH == 0) {                                      // rows 0 .. 31 of chunk c + 1
                stage_store(c + 1);
                stage_load(min(c + 2, nchunk - 1));
            }
            bst.park(bimg, lane);                               // (the image's readers of chunk c are this wave's own earlier reads)
            bst.load(brows, min(c + 2, nchunk - 1));
            if (DROP) drop_build(c + 1, 0);
            writeout(c, 0, MH == 1);
            __syncthreads();
        }
        writeout(nchunk - 1, 1, MH == 0);
    };
    if (my_half) phases(std::integral_constant<int, 1>{});
    else phases(std::integral_constant<int, 0>{});

    if (k_ok) {
        TQ* DK = reinterpret_cast<TQ*>(p.dk) + ((int64_t)g * T + my_k) * p.lddk + h * D + 16 * hi;
        TQ* DV = reinterpret_cast<TQ*>(p.dv) + ((int64_t)g * T + my_k) * p.lddv + h * D + 16 * hi;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (16 * hi + 8 * j < D) {
                float a[8], b[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { a[i] = dk[8 * j + i] * p.scale; b[i] = DROP ? dv[8 * j + i] * p.inv_keep : dv[8 * j + i]; }
                store8(DK + 8 * j, a);
                store8(DV + 8 * j, b);
            }
        }
    }
}

// Behind the one-pass backward: scale + cast of dQ; the accumulator is left zero for the next call.
__global__ void attn_dq_finish_kernel(float* acc, bf16_t* dq, int64_t rows, int C, int64_t lddq, float scale) {
    const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (e >= rows * C) return;
    const int64_t r = e / C;
    const int c = (int)(e % C);
    float v[8];
    load8(acc + e, v);
    *reinterpret_cast<float4*>(acc + e) = make_float4(0.f, 0.f, 0.f, 0.f);          // zero again for the next call
    *reinterpret_cast<float4*>(acc + e + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= scale;
    store8(dq + r * lddq + c, v);
}

// ------------------------------------------------------------------------------------ dispatch
enum Pass { FWD, BWD_DQ, BWD_DKV, BWD_BOTH };

// Workgroups per (graph, head).  Cost model: every round of resident workgroups costs (workgroups per compute unit in that
// round) x (live waves per workgroup = nwt / nq); candidates nq0 = ceil(nwt / NW) ... nq0 + 3, fewer workgroups on near-ties
// (each one stages all of K / V).  c5 (nwt 25, 128 (graph, head)s, 256 CUs): forward cap 4 -> 8 (one round of 1024 instead of
// 896 uneven ones), dQ pass cap 3 -> 8 (768 + 256 instead of 768 + 128 nearly empty ones), dK/dV pass cap 2 -> 8 (two whole rounds).
int choose_nq(int GH, int T, int NW, int cap, int cus) {
    const int nwt = (T + 31) / 32, nq0 = (nwt + NW - 1) / NW;
    if (nwt <= NW || cap < 1 || cus < 1) return nq0;
    int best = nq0;
    double best_cost = 1e30;
    for (int nq = nq0; nq <= nwt && nq <= nq0 + 3; ++nq) {
        const long slots = (long)cus * cap;
        double cost = 0;
        for (long left = (long)GH * nq; left > 0; left -= slots) {
            const long in_round = left < slots ? left : slots;
            cost += (double)((in_round + cus - 1) / cus) * nwt / nq;
        }
        if (cost < best_cost * 0.97) { best_cost = cost; best = nq; }
    }
    return best;
}

// MOBGT_ATTN_TWO_PASS=1: the two deterministic passes of rounds 1-3 also where the one-pass kernel applies (A/B runs, tests)
bool one_pass_off() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MOBGT_ATTN_TWO_PASS");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

template <typename K>
int blocks_per_cu(K kernel, int threads) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kernel), threads, 0) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 1;
    }
    return n;
}

int cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
            (void)hipGetLastError();
            cus = 256;
        }
    }
    return cus;
}

template <Pass PASS, int D, typename TQ, typename TB, int NW, bool DROP>
hipError_t launch_one(const AttnParams& p0, hipStream_t st) {
    const dim3 block(NW * 64);
    const int GH = p0.G * p0.H;
    AttnParams p = p0;
    if constexpr (PASS == BWD_BOTH) {
        if constexpr (NW < 4) {
            p.nq = ((p.T + 31) / 32 + NW - 1) / NW;
            p.n_first = GH * p.nq;
            hipLaunchKernelGGL((attn_bwd_both_kernel<D, TQ, TB, NW, DROP>), dim3(2 * GH * p.nq), block, 0, st, p);
        } else if (std::is_same<TQ, bf16_t>::value && std::is_same<TB, bf16_t>::value && p.dq_acc && p.dbias_bf16 && !one_pass_off()) {
            // ONE pass (attn_bwd_one_kernel) into the caller's zero accumulator, then scale + cast of dQ (which zeroes it again)
            const int C = p.H * D;
            p.nq = ((p.T + 31) / 32 + ONE_NW - 1) / ONE_NW;
            hipLaunchKernelGGL((attn_bwd_one_kernel<D, DROP>), dim3(GH * p.nq), dim3(ONE_NW * 64), 0, st, p);
            const int64_t n8 = (int64_t)p.G * p.T * C / 8;
            // (dq rows: q / k / v gradients share a [G, T, 3C] buffer in the fused layer, hence the row stride lddq; one head's
            //  columns start at h * D inside the C the accumulator holds, and the launcher passes dq already offset to column 0)
            hipLaunchKernelGGL(attn_dq_finish_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, p.dq_acc,
                               reinterpret_cast<bf16_t*>(p.dq), (int64_t)p.G * p.T, C, p.lddq, p.scale);
        } else {
            const int nq0 = ((p.T + 31) / 32 + NW - 1) / NW;
            p.nq = nq0;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<D, TQ, TB, NW, DROP>), dim3(GH * p.nq), block, 0, st, p);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, TQ, TB, NW, DROP>), dim3(GH * p.nq), block, 0, st, p);
        }
        return hipGetLastError();
    }
    if (PASS == FWD) {
        static const int cap = blocks_per_cu(attn_fwd_kernel<D, TQ, TB, NW, DROP>, NW * 64);
        p.nq = NW < 4 ? ((p.T + 31) / 32 + NW - 1) / NW : choose_nq(GH, p.T, NW, cap, cu_count());
        hipLaunchKernelGGL((attn_fwd_kernel<D, TQ, TB, NW, DROP>), dim3(GH * p.nq), block, 0, st, p);
    } else if (PASS == BWD_DQ) {
        p.nq = ((p.T + 31) / 32 + NW - 1) / NW;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<D, TQ, TB, NW, DROP>), dim3(GH * p.nq), block, 0, st, p);
    } else {
        p.nq = ((p.T + 31) / 32 + NW - 1) / NW;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, TQ, TB, NW, DROP>), dim3(GH * p.nq), block, 0, st, p);
    }
    return hipGetLastError();
}

template <Pass PASS, int D, typename TQ, typename TB>
hipError_t launch_nw(const AttnParams& p, bool drop, hipStream_t st) {
    // small graphs: fewer waves per workgroup -> more workgroups to spread over the 256 CUs
    const int nw = p.T <= 32 ? 1 : (p.T <= 64 ? 2 : 4);
    if (drop) {
        if (nw == 1) return launch_one<PASS, D, TQ, TB, 1, true>(p, st);
        if (nw == 2) return launch_one<PASS, D, TQ, TB, 2, true>(p, st);
        return launch_one<PASS, D, TQ, TB, 4, true>(p, st);
    }
    if (nw == 1) return launch_one<PASS, D, TQ, TB, 1, false>(p, st);
    if (nw == 2) return launch_one<PASS, D, TQ, TB, 2, false>(p, st);
    return launch_one<PASS, D, TQ, TB, 4, false>(p, st);
}

template <Pass PASS, typename TQ, typename TB>
int launch_d(const AttnParams& p, int d, bool drop, hipStream_t st) {
    switch (d) {
        case 16: return (int)launch_nw<PASS, 16, TQ, TB>(p, drop, st);
        case 24: return (int)launch_nw<PASS, 24, TQ, TB>(p, drop, st);
        case 32: return (int)launch_nw<PASS, 32, TQ, TB>(p, drop, st);
        default: return MOBGT_EBADDIM;
    }
}

#include "attn_f32_body.h"

template <Pass PASS>
int launch(const AttnParams& p, int d, int io_dtype, int bias_dtype, bool drop, hipStream_t st) {
    // f32 I/O: the full-f32 instantiation (attn_f32_body.h: f32 matrix instruction, no bf16 rounding anywhere); the kernels above
    // are instantiated for bf16 I/O only
    if (io_dtype == MOBGT_F32 && bias_dtype == MOBGT_F32) return launch_f32<PASS == FWD, float>(p, d, drop, st);
    if (io_dtype == MOBGT_F32 && bias_dtype == MOBGT_BF16) return launch_f32<PASS == FWD, bf16_t>(p, d, drop, st);
    if (io_dtype == MOBGT_BF16 && bias_dtype == MOBGT_F32) return launch_d<PASS, bf16_t, float>(p, d, drop, st);
    if (io_dtype == MOBGT_BF16 && bias_dtype == MOBGT_BF16) return launch_d<PASS, bf16_t, bf16_t>(p, d, drop, st);
    return MOBGT_EDTYPE;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_common(int G, int H, int T, int d, int64_t ld_bias, int io_dtype, int64_t lds_min_mult) {
    if (G <= 0 || H <= 0 || T <= 0) return MOBGT_EBADDIM;
    if (d != 16 && d != 24 && d != 32) return MOBGT_EBADDIM;
    // 64-column chunks are read as whole row segments (BiasStage): rows must cover roundup(T, 64) columns and start on
    // 128-byte lines (bf16)
    if (ld_bias % 64 != 0 || ld_bias < T) return MOBGT_EALIGN;
    (void)io_dtype; (void)lds_min_mult;
    return 0;
}

void set_dropout(AttnParams& p, float dropout_p, uint64_t seed, const uint64_t* seed_dev) {
    p.drop_thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    p.inv_keep = p.drop_thr ? 1.f / (1.f - (float)p.drop_thr / 65536.f) : 1.f;
    p.thr_s = (int)p.drop_thr - 32768;
    p.seed = seed;
    p.seed_dev = seed_dev;
}

}  // namespace

extern "C" int mobgt_attn_bias_fwd(const void* q, const void* k, const void* v, const void* bias, void* out, void* out_lo, float* lse,
                                   int G, int H, int T, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                   int64_t ld_bias, float scale, float dropout_p, uint64_t seed,
                                   const uint64_t* seed_dev, int io_dtype, int bias_dtype, void* stream) {
    int rc = check_common(G, H, T, d, ld_bias, io_dtype, 8);
    if (rc) return rc;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || !aligned16(bias) || !aligned16(out_lo)) return MOBGT_EALIGN;
    if ((ldq | ldk | ldv | ldo) % 8 != 0) return MOBGT_EALIGN;
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.bias = bias; p.o = out; p.lse = lse;
    p.o_lo = io_dtype == MOBGT_BF16 ? out_lo : nullptr;
    p.G = G; p.H = H; p.T = T;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.ld_bias = ld_bias;
    p.scale = scale;
    set_dropout(p, dropout_p, seed, seed_dev);
    return launch<FWD>(p, d, io_dtype, bias_dtype, p.drop_thr != 0, (hipStream_t)stream);
}

static int attn_bwd_impl(const void* q, const void* k, const void* v, const void* bias, const void* bias_t,
                                   const void* out, const void* out_lo, const float* lse, const void* dout, void* dq, void* dk, void* dv,
                                   void* dbias, float* delta, int G, int H, int T, int d, int64_t ldq, int64_t ldk,
                                   int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int64_t ld_bias,
                                   float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                   int accumulate_dbias, int dbias_dtype, int io_dtype, int bias_dtype, float* dq_acc, void* stream) {
    int rc = check_common(G, H, T, d, ld_bias, io_dtype, 8);
    if (rc) return rc;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || !aligned16(dout) || !aligned16(bias) ||
        !aligned16(bias_t) || !aligned16(dq) || !aligned16(dk) || !aligned16(dv) || !aligned16(dbias) || !aligned16(out_lo))
        return MOBGT_EALIGN;
    if ((ldq | ldk | ldv | ldo | lddq | lddk | lddv) % 8 != 0) return MOBGT_EALIGN;
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.bias = bias; p.bias_t = bias_t; p.out = out; p.dout = dout; p.lse_in = lse;
    p.out_lo = io_dtype == MOBGT_BF16 ? out_lo : nullptr;
    p.dq = dq; p.dk = dk; p.dv = dv; p.dbias = dbias; p.delta = delta;
    p.dq_acc = (dq_acc && aligned16(dq_acc)) ? dq_acc : nullptr;
    p.G = G; p.H = H; p.T = T;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.ld_bias = ld_bias;
    p.scale = scale;
    p.accumulate = accumulate_dbias;
    if (dbias_dtype != MOBGT_F32 && dbias_dtype != MOBGT_BF16) return MOBGT_EDTYPE;
    p.dbias_bf16 = dbias_dtype == MOBGT_BF16;
    set_dropout(p, dropout_p, seed, seed_dev);
    const bool drop = p.drop_thr != 0;
    // T <= 64: both passes in one launch; larger graphs: ONE pass when the caller brought the dQ accumulator (bf16 I/O, bf16 bias
    // and dBias slice), else the dQ pass, then the dK/dV pass (it reads the dQ pass's `delta`)
    return launch<BWD_BOTH>(p, d, io_dtype, bias_dtype, drop, (hipStream_t)stream);
}

extern "C" int mobgt_attn_bias_bwd(const void* q, const void* k, const void* v, const void* bias, const void* bias_t,
                                   const void* out, const void* out_lo, const float* lse, const void* dout, void* dq, void* dk, void* dv,
                                   void* dbias, float* delta, int G, int H, int T, int d, int64_t ldq, int64_t ldk,
                                   int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int64_t ld_bias,
                                   float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                   int accumulate_dbias, int dbias_dtype, int io_dtype, int bias_dtype, void* stream) {
    return attn_bwd_impl(q, k, v, bias, bias_t, out, out_lo, lse, dout, dq, dk, dv, dbias, delta, G, H, T, d, ldq, ldk, ldv, ldo, lddq, lddk,
                         lddv, ld_bias, scale, dropout_p, seed, seed_dev, accumulate_dbias, dbias_dtype, io_dtype, bias_dtype,
                         nullptr, stream);
}

/* The same backward with an f32 accumulator for dQ: dq_acc [G, T, H * d], 16-byte aligned, ALL ZEROS on entry and all zeros again
 * when the call's last launch has run (the caller keeps one per stream and never touches it).  With it, T > 64, bf16 I/O, a bf16
 * bias and a bf16 dBias slice the gradients are formed in ONE pass over the bias (attn_bwd_one_kernel: S / P / dS once per pair,
 * bias_t read once, dBias written once, rowsum(dO O) formed inside the pass, dQ summed over key blocks by f32 atomics -- so dQ is
 * then NOT bitwise reproducible from run to run; MOBGT_ATTN_TWO_PASS=1 keeps the two deterministic passes). */
extern "C" int mobgt_attn_bias_bwd_fused_z(const void* q, const void* k, const void* v, const void* bias, const void* bias_t,
                                   const void* out, const void* out_lo, const float* lse, const void* dout, void* dq, void* dk, void* dv,
                                   void* dbias, float* delta, int G, int H, int T, int d, int64_t ldq, int64_t ldk,
                                   int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int64_t ld_bias,
                                   float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                   int accumulate_dbias, int dbias_dtype, int io_dtype, int bias_dtype, float* dq_acc, void* stream) {
    return attn_bwd_impl(q, k, v, bias, bias_t, out, out_lo, lse, dout, dq, dk, dv, dbias, delta, G, H, T, d, ldq, ldk, ldv, ldo, lddq, lddk,
                         lddv, ld_bias, scale, dropout_p, seed, seed_dev, accumulate_dbias, dbias_dtype, io_dtype, bias_dtype,
                         dq_acc, stream);
}

extern "C" int mobgt_dropout_keep_host(uint64_t seed, int H, int T, int g, int h, int i, int j, float dropout_p) {
    const uint32_t thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    if (!thr) return 1;
    const uint32_t rowh = dropout_row_hash(seed, (uint32_t)((g * H + h) * T + i));
    return attn_drop_keep(seed, rowh, (uint32_t)j, (int)thr - 32768) ? 1 : 0;
}

// Bulk forms of the host replay (tests replay whole masks into the oracle; one ctypes call per element is too slow at
// T = 785): out[((g*H + h)*T + i)*T + j] = keep of probability (g,h,i,j) ...
extern "C" int mobgt_attn_dropout_mask_host(uint64_t seed, int G, int H, int T, float dropout_p, uint8_t* out) {
    if (G < 0 || H <= 0 || T <= 0 || !out) return MOBGT_EBADDIM;
    const uint32_t thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    for (int64_t row = 0; row < (int64_t)G * H * T; ++row) {
        const uint32_t rowh = dropout_row_hash(seed, (uint32_t)row);
        uint8_t* o = out + row * T;
        for (int j = 0; j < T; ++j) o[j] = (!thr || attn_drop_keep(seed, rowh, (uint32_t)j, (int)thr - 32768)) ? 1 : 0;
    }
    return 0;
}

// ... and of the rule every OTHER dropout site of the step uses (residual branches, token assembly, head, GCN: csrc/layer.hip,
// chain.hip, lngemm.hip, sgemm.hip, smallgcn.hip): out[r*C + c] = dropout_bits16(seed, hash(seed, (row0 + r) ^ salt), c) >= thr.
extern "C" int mobgt_dropout_mask_host(uint64_t seed, uint32_t salt, int64_t row0, int64_t R, int C, float dropout_p, uint8_t* out) {
    if (R < 0 || C <= 0 || !out) return MOBGT_EBADDIM;
    const uint32_t thr = dropout_p > 0.f ? dropout_threshold(dropout_p) : 0u;
    for (int64_t r = 0; r < R; ++r) {
        const uint32_t rowh = dropout_row_hash(seed, (uint32_t)(row0 + r) ^ salt);
        uint8_t* o = out + r * C;
        for (int c = 0; c < C; ++c) o[c] = (!thr || dropout_bits16(seed, rowh, (uint32_t)c) >= thr) ? 1 : 0;
    }
    return 0;
}
