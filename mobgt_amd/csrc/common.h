// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MOBGT_LOG2E 1.4426950408889634f
#define MOBGT_LN2 0.6931471805599453f
#define MOBGT_NEG_BIG (-1.0e30f)

// MFMA row permutation.  v_mfma_f32_32x32x16_bf16 leaves output row m = (i&3) + 8*(i>>2) + 4*hi in
// accumulator register i of lane half hi.  Feeding the A operand's row m from logical row kappa(m)
// makes register i of half hi hold logical row 16*hi + i: 16 CONTIGUOUS keys (or head columns) per
// lane, so bias tiles, dBias tiles and outputs move as 16-byte vectors.
__device__ __forceinline__ int kappa(int m) { return 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3); }

// raw v_exp_f32: inputs here are <= 0 (score minus running max / LSE), so the denormal-range fix-up that
// exp2f() adds (v_ldexp + compare + select per element) buys nothing; -inf -> 0 exactly.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// 8 contiguous elements -> fp32 registers.  p must be 16-byte aligned.
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = bf16_lo(a.x); v[1] = bf16_hi(a.x); v[2] = bf16_lo(a.y); v[3] = bf16_hi(a.y);
    v[4] = bf16_lo(a.z); v[5] = bf16_hi(a.z); v[6] = bf16_lo(a.w); v[7] = bf16_hi(a.w);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = o;
}
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
    return o;
}

// 16 contiguous bias elements -> the accumulator registers of one lane.
template <typename TB> struct BiasRegs;
template <> struct BiasRegs<float> {
    float4 r[4];
    __device__ __forceinline__ void load(const float* p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = reinterpret_cast<const float4*>(p)[i];
    }
    __device__ __forceinline__ void to_acc(f32x16& s) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s[4 * i] = r[i].x; s[4 * i + 1] = r[i].y; s[4 * i + 2] = r[i].z; s[4 * i + 3] = r[i].w; }
    }
};
template <> struct BiasRegs<bf16_t> {
    uint4 r[2];
    __device__ __forceinline__ void load(const bf16_t* p) {
        r[0] = reinterpret_cast<const uint4*>(p)[0];
        r[1] = reinterpret_cast<const uint4*>(p)[1];
    }
    __device__ __forceinline__ void to_acc(f32x16& s) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            s[8 * i + 0] = bf16_lo(r[i].x); s[8 * i + 1] = bf16_hi(r[i].x);
            s[8 * i + 2] = bf16_lo(r[i].y); s[8 * i + 3] = bf16_hi(r[i].y);
            s[8 * i + 4] = bf16_lo(r[i].z); s[8 * i + 5] = bf16_hi(r[i].z);
            s[8 * i + 6] = bf16_lo(r[i].w); s[8 * i + 7] = bf16_hi(r[i].w);
        }
    }
};

// ---- attention dropout: keep(seed, row, key) -- a pure function so fwd and bwd agree --------------
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// row = (g*H + h)*T + i ; returns 16 random bits for key j
__host__ __device__ __forceinline__ uint32_t dropout_row_hash(uint64_t seed, uint32_t row) {
    return mix32(row ^ (uint32_t)seed);
}
// Per-element mixer built from FULL-RATE instructions only: v_mul_lo_u32 (what mix32 needs twice) issues at a
// quarter of the rate of v_mad_u32_u24 on CDNA, and this function runs once per pair of probabilities in the
// attention kernels' inner loops.  The 24-bit multiplies drop the state's top byte, which the addend (x >> 8,
// x >> 11) re-injects.  Checked on 16 M-element masks against mix32: mean keep rate, lag-1/2/8/32 row and column
// autocorrelations (|r| < 1e-3), chi-square of both bytes, row/column-sum variance vs binomial, step-to-step
// correlation -- indistinguishable.
__host__ __device__ __forceinline__ uint32_t mix24(uint32_t x) {
    x = (x & 0xffffffu) * 0x9E3779u + (x >> 8);
    x ^= x >> 13;
    x = (x & 0xffffffu) * 0x85EBCBu + (x >> 11);
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t dropout_bits16(uint64_t seed, uint32_t row_hash, uint32_t key) {
    const uint32_t w = mix24(row_hash + (key >> 1) * 0x9E3779B9u + (uint32_t)(seed >> 32));
    return (key & 1u) ? (w >> 16) : (w & 0xffffu);
}
__host__ __device__ __forceinline__ uint32_t dropout_threshold(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }

// ---- attention dropout (model.py:451), rule v2: ONE strong hash per (query row, block of 16 keys) ---------------------
// The attention kernels hold 16 contiguous keys of one query row per lane (forward, dQ pass) or 16 contiguous query rows
// of one key per lane (dK/dV pass).  Rule v1 paid a full mix24 per pair of keys (forward orientation) or per ELEMENT
// (dK/dV orientation): 7-14 VALU instructions per probability, more than the softmax itself.  v2:
//     hb   = mix24(row_hash + (key >> 4) * golden + seed_hi)                  one per (row, 16-key block)
//     w    = (hb & 0xffffff) * A[(key & 15) >> 1] + (hb >> 8)                 one v_mad_u32_u24 per PAIR of keys
//     keep = int16(key odd ? w >> 16 : w) >= thr - 32768                      signed 16-bit uniform vs threshold
// so a lane of the forward orientation spends 1 hash + 8 mads + 16 compares per 16 probabilities, and the dK/dV pass
// builds the w words of a tile cooperatively (one hash + 8 mads per thread) and reads 16 of them per lane from LDS.
// Checked on 8 M-element masks (numpy replica): keep rate, row / column lag correlations, all 120 within-block position
// pairs (|r| <= 0.0035 ~ 2.5 sigma), block / row / column sum variance vs binomial, step-to-step correlation: clean.
// (w = ... + hb instead of + (hb >> 8) showed a 1.5 % correlation between two block positions and was dropped.)
__host__ __device__ __forceinline__ uint32_t attn_drop_mult(int m) {
    constexpr uint32_t A[8] = {0x9E3779u, 0x85EBCBu, 0xC2B2AFu, 0x27D4EBu, 0x165667u, 0xD3A265u, 0xFD7047u, 0xB55A4Fu};
    return A[m & 7];
}
__host__ __device__ __forceinline__ uint32_t attn_drop_block(uint64_t seed, uint32_t row_hash, uint32_t key_block) {
    return mix24(row_hash + key_block * 0x9E3779B9u + (uint32_t)(seed >> 32));
}
#ifndef ATTN_MAD24_ASM
#define ATTN_MAD24_ASM 0
#endif
__host__ __device__ __forceinline__ uint32_t attn_drop_word(uint32_t hb, uint32_t mult) {
#if defined(__HIP_DEVICE_COMPILE__) && ATTN_MAD24_ASM
    // one full-rate v_mad_u32_u24 per word, spelled out: from the C expression hipcc derives word m + 1 from word m by a
    // strength-reduced chain of v_mad_u64_u32 (a quarter-rate instruction with a 64-bit result) -- seen in the ISA of all three
    // attention kernels, four of the eight words of every 16-key block
    uint32_t w;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(w) : "v"(hb), "s"(mult), "v"(hb >> 8));
    return w;
#else
    return (hb & 0xffffffu) * mult + (hb >> 8);
#endif
}
// thr_s = (int)dropout_threshold(p) - 32768
__host__ __device__ __forceinline__ bool attn_drop_keep_even(uint32_t w, int thr_s) { return (int)(int16_t)(w & 0xffffu) >= thr_s; }
__host__ __device__ __forceinline__ bool attn_drop_keep_odd(uint32_t w, int thr_s) { return (int)w >= thr_s * 65536; }
// Both decisions of a pair word at once, as a mask for the PACKED pair (0xffff over a kept half): per 16-bit half,
// saturating (thr_s - 1) - w is negative exactly when w >= thr_s; the arithmetic shift by 15 spreads the sign.
typedef short mobgt_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t attn_drop_keep_mask2(uint32_t w, int thr_s) {
    const short t1 = (short)(thr_s - 1);
    const mobgt_i16x2 tv = {t1, t1};
    const mobgt_i16x2 d = __builtin_elementwise_sub_sat(tv, __builtin_bit_cast(mobgt_i16x2, w));
    return __builtin_bit_cast(uint32_t, d >> 15);
}
__host__ __device__ __forceinline__ bool attn_drop_keep(uint64_t seed, uint32_t row_hash, uint32_t key, int thr_s) {
    const uint32_t w = attn_drop_word(attn_drop_block(seed, row_hash, key >> 4), attn_drop_mult((int)((key & 15u) >> 1)));
    return (key & 1u) ? attn_drop_keep_odd(w, thr_s) : attn_drop_keep_even(w, thr_s);
}

// Bijective XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give
// each XCD a contiguous run of logical ids (neighbouring q-tiles of one (graph, head) share K/V in L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
