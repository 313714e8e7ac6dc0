// Bias-fused attention with FULL f32 arithmetic: the instantiation for f32 I/O (included by attn.hip; same C entry points).
//
// The bf16 kernels of attn.hip round Q / K / V / P / dS to bf16 for the matrix cores (their results sit 2-7e-3 from an fp32
// reference on O(1) values).  A caller who hands over f32 activations -- a reference checkpoint evaluated in `--precision 32`,
// the parity configuration of the tests -- gets every product from v_mfma_f32_32x32x2_f32 here: f32 operands, f32 accumulate,
// f32 softmax; the result is the reference's up to summation order (graphormer/model.py:436-455 and its autograd;
// tests/test_gpu_layer.py pins the encoder layer of golden G4 at 1e-4).  It is the accuracy configuration, not the benched
// one: the f32 matrix instruction has 1/16 of the bf16 one's rate and nothing is staged through LDS -- at MobGT's sizes
// (T <= 130 outside the tail) the launches are latency-bound either way.
//
// Orientation as in attn.hip: S^T = K Q^T, one wave owns 32 query rows (forward, dQ pass) or 32 keys (dK/dV pass), the A
// operand's rows go through kappa() so that accumulator register i of lane half hi is logical row 16 hi + i.  The f32
// instruction contracts TWO k-values per issue (lane half hi supplies k = hi), which makes LDS unnecessary:
//   * "NT" products (S^T = K Q^T, dP^T = V dO^T): k runs over head columns; instruction j pairs column j with column D/2 + j,
//     so lane (n, hi) needs the half row [hi D/2, (hi + 1) D/2) of its A row and of its own B row: two contiguous loads;
//   * "TN" products (O^T = V^T P^T, dQ^T = K^T dS^T, dK^T = Q^T dS, dV^T = dO^T P): k runs over the tile's 32 rows;
//     instruction j pairs row j with row 16 + j, the B value is the lane's own accumulator register j (P or dS of logical row
//     16 hi + j), the A value is element n of tile row 16 hi + j: a coalesced 128-byte row read per half.
// Deterministic (no atomics): dQ / dBias pass, then dK / dV pass (which reads the first pass's `delta`).
#pragma once

typedef float mobgt_f32x16 __attribute__((ext_vector_type(16)));

template <typename TB>
__device__ __forceinline__ void f32_bias16(const TB* p, mobgt_f32x16& s) {
    float v[16];
    if constexpr (sizeof(TB) == 2) {
        load8(p, *reinterpret_cast<float(*)[8]>(&v[0]));
        load8(p + 8, *reinterpret_cast<float(*)[8]>(&v[8]));
    } else {
        load8(p, *reinterpret_cast<float(*)[8]>(&v[0]));
        load8(p + 8, *reinterpret_cast<float(*)[8]>(&v[8]));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = v[i];
}

// this lane's half row [hi D/2, +D/2) of `row` (row stride ld), times mul; zero when !ok
template <int D>
__device__ __forceinline__ void f32_half_row(const float* base, int64_t ld, int row, int hi, bool ok, float mul, float (&r)[D / 2]) {
    const float* p = base + (int64_t)row * ld + hi * (D / 2);
#pragma unroll
    for (int j = 0; j < D / 2; j += 4) {
        const float4 v = ok ? *reinterpret_cast<const float4*>(p + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        r[j] = v.x * mul; r[j + 1] = v.y * mul; r[j + 2] = v.z * mul; r[j + 3] = v.w * mul;
    }
}

// acc^T[m][n] += sum over head columns of A[kappa(m)][c] * B[n][c]   (a = this lane's half of A row kappa(n), b = of B row n)
template <int D>
__device__ __forceinline__ void f32_nt(const float (&a)[D / 2], const float (&b)[D / 2], mobgt_f32x16& acc) {
#pragma unroll
    for (int j = 0; j < D / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}

// acc^T[dim][n] += sum over the tile's 32 rows r of M[row0 + r][dim] * w[r][n], w = the lane's accumulator-layout values
// (register j = logical row 16 hi + j); rows >= T and head columns >= D contribute zero
template <int D>
__device__ __forceinline__ void f32_tn(const float* base, int64_t ld, int row0, int T, int n, int hi, const float (&w)[16], mobgt_f32x16& acc) {
    // (the A operand's row m is logical head column kappa(m): lane n reads column kappa(n))
    const int col = kappa(n);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = row0 + 16 * hi + j;
        const float a = (r < T && col < D) ? base[(int64_t)r * ld + col] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[j], acc, 0, 0, 0);
    }
}

template <int D, typename TB, bool DROP>
__global__ __launch_bounds__(64) void attn_f32_fwd_kernel(const AttnParams p) {
    const int T = p.T, H = p.H;
    const int nqt = (T + 31) >> 5;
    const int qt = blockIdx.x % nqt, gh = blockIdx.x / nqt, g = gh / H, h = gh % H;
    const int lane = threadIdx.x, n = lane & 31, hi = lane >> 5;
    const int q0 = qt * 32, my_q = q0 + n;
    const bool q_ok = my_q < T;
    const int qc = q_ok ? my_q : T - 1;
    const float* Q = reinterpret_cast<const float*>(p.q) + (int64_t)g * T * p.ldq + h * D;
    const float* K = reinterpret_cast<const float*>(p.k) + (int64_t)g * T * p.ldk + h * D;
    const float* V = reinterpret_cast<const float*>(p.v) + (int64_t)g * T * p.ldv + h * D;
    const TB* brow = reinterpret_cast<const TB*>(p.bias) + ((int64_t)gh * T + qc) * p.ld_bias + 16 * hi;
    float qh[D / 2];
    f32_half_row<D>(Q, p.ldq, qc, hi, q_ok, p.scale, qh);
    uint32_t rowh = 0;
    uint64_t seed = 0;
    if (DROP) {
        seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
        rowh = dropout_row_hash(seed, (uint32_t)(gh * T + qc));
    }
    float m = MOBGT_NEG_BIG, l = 0.f;
    mobgt_f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    for (int key0 = 0; key0 < T; key0 += 32) {
        mobgt_f32x16 s;
        f32_bias16(brow + key0, s);                   // (columns [T, ld) hold -inf; ld >= roundup(T, 64) covers the tile)
        float kh[D / 2];
        const int kr = key0 + kappa(n);
        f32_half_row<D>(K, p.ldk, min(kr, T - 1), hi, kr < T, 1.f, kh);
        f32_nt<D>(kh, qh, s);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (key0 + 16 * hi + i >= T) s[i] = -INFINITY;
        float tmax = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
        tmax = xhalf_max(tmax);
        const float m_new = fmaxf(m, tmax);
        const float alpha = fast_exp2((m - m_new) * MOBGT_LOG2E);
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] *= alpha;
        l *= alpha;
        m = m_new;
        const float ms = m * MOBGT_LOG2E;
        float pr[16];
        uint32_t hb = 0;
        if (DROP) hb = attn_drop_block(seed, rowh, (uint32_t)((key0 >> 4) + hi));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = exp2f(fmaf(s[i], MOBGT_LOG2E, -ms));
            l += e;
            bool keep = true;
            if (DROP) {
                const uint32_t w = attn_drop_word(hb, attn_drop_mult(i >> 1));
                keep = (i & 1) ? attn_drop_keep_odd(w, p.thr_s) : attn_drop_keep_even(w, p.thr_s);
            }
            pr[i] = keep ? e : 0.f;
        }
        f32_tn<D>(V, p.ldv, key0, T, n, hi, pr, o);
    }
    const float ltot = xhalf_sum(l);
    const float inv = (DROP ? p.inv_keep : 1.f) / ltot;
    if (q_ok) {
        float* O = reinterpret_cast<float*>(p.o) + ((int64_t)g * T + my_q) * p.ldo + h * D + 16 * hi;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (16 * hi + 8 * j < D) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = o[8 * j + i] * inv;
                store8(O + 8 * j, v);
            }
        }
        if (hi == 0) p.lse[(int64_t)gh * T + my_q] = m + logf(ltot);
    }
}

// dQ + dBias + delta (queries on the lanes, sweep over keys)
template <int D, typename TB, bool DROP>
__global__ __launch_bounds__(64) void attn_f32_dq_kernel(const AttnParams p) {
    const int T = p.T, H = p.H;
    const int nqt = (T + 31) >> 5;
    const int qt = blockIdx.x % nqt, gh = blockIdx.x / nqt, g = gh / H, h = gh % H;
    const int lane = threadIdx.x, n = lane & 31, hi = lane >> 5;
    const int q0 = qt * 32, my_q = q0 + n;
    const bool q_ok = my_q < T;
    const int qc = q_ok ? my_q : T - 1;
    const float* Q = reinterpret_cast<const float*>(p.q) + (int64_t)g * T * p.ldq + h * D;
    const float* K = reinterpret_cast<const float*>(p.k) + (int64_t)g * T * p.ldk + h * D;
    const float* V = reinterpret_cast<const float*>(p.v) + (int64_t)g * T * p.ldv + h * D;
    const float* O = reinterpret_cast<const float*>(p.out) + (int64_t)g * T * p.ldo + h * D;
    const float* dO = reinterpret_cast<const float*>(p.dout) + (int64_t)g * T * p.ldo + h * D;
    const TB* brow = reinterpret_cast<const TB*>(p.bias) + ((int64_t)gh * T + qc) * p.ld_bias + 16 * hi;
    float qh[D / 2], doh[D / 2], oh[D / 2];
    f32_half_row<D>(Q, p.ldq, qc, hi, q_ok, p.scale, qh);
    f32_half_row<D>(dO, p.ldo, qc, hi, q_ok, 1.f, doh);
    f32_half_row<D>(O, p.ldo, qc, hi, q_ok, 1.f, oh);
    float dpart = 0.f;
#pragma unroll
    for (int j = 0; j < D / 2; ++j) dpart = fmaf(doh[j], oh[j], dpart);
    const float delta = xhalf_sum(dpart);
    if (q_ok && hi == 0) p.delta[(int64_t)gh * T + my_q] = delta;
    const float lse2 = p.lse_in[(int64_t)gh * T + qc] * MOBGT_LOG2E;
    uint32_t rowh = 0;
    uint64_t seed = 0;
    if (DROP) {
        seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
        rowh = dropout_row_hash(seed, (uint32_t)(gh * T + qc));
    }
    const int64_t dboff = ((int64_t)gh * T + qc) * p.ld_bias + 16 * hi;
    mobgt_f32x16 dq;
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[i] = 0.f;
    for (int key0 = 0; key0 < T; key0 += 32) {
        mobgt_f32x16 s, dp;
        f32_bias16(brow + key0, s);
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[i] = 0.f;
        float kh[D / 2], vh[D / 2];
        const int kr = key0 + kappa(n);
        f32_half_row<D>(K, p.ldk, min(kr, T - 1), hi, kr < T, 1.f, kh);
        f32_half_row<D>(V, p.ldv, min(kr, T - 1), hi, kr < T, 1.f, vh);
        f32_nt<D>(kh, qh, s);
        f32_nt<D>(vh, doh, dp);
        float ds[16];
        uint32_t hb = 0;
        if (DROP) hb = attn_drop_block(seed, rowh, (uint32_t)((key0 >> 4) + hi));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool live = key0 + 16 * hi + i < T;
            const float pr = live ? exp2f(fmaf(s[i], MOBGT_LOG2E, -lse2)) : 0.f;
            float dd = DROP ? fmaf(dp[i], p.inv_keep, -delta) : dp[i] - delta;
            if (DROP) {
                const uint32_t w = attn_drop_word(hb, attn_drop_mult(i >> 1));
                const bool keep = (i & 1) ? attn_drop_keep_odd(w, p.thr_s) : attn_drop_keep_even(w, p.thr_s);
                dd = keep ? dd : -delta;
            }
            ds[i] = pr * dd;
        }
        if (p.dbias && q_ok) {
            if (p.dbias_bf16) {
                bf16_t* dst = reinterpret_cast<bf16_t*>(p.dbias) + dboff + key0;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (key0 + 16 * hi + 8 * j + 8 <= p.ld_bias) store8(dst + 8 * j, *reinterpret_cast<const float(*)[8]>(&ds[8 * j]));
            } else {
                float* dst = reinterpret_cast<float*>(p.dbias) + dboff + key0;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (key0 + 16 * hi + i < T) dst[i] = p.accumulate ? dst[i] + ds[i] : ds[i];
            }
        }
        f32_tn<D>(K, p.ldk, key0, T, n, hi, ds, dq);
    }
    if (q_ok) {
        float* DQ = reinterpret_cast<float*>(p.dq) + ((int64_t)g * T + my_q) * p.lddq + h * D + 16 * hi;
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

// dK + dV (keys on the lanes, sweep over queries; reads the transposed bias, lse and the first pass's delta)
template <int D, typename TB, bool DROP>
__global__ __launch_bounds__(64) void attn_f32_dkv_kernel(const AttnParams p) {
    const int T = p.T, H = p.H;
    const int nkt = (T + 31) >> 5;
    const int kt = blockIdx.x % nkt, gh = blockIdx.x / nkt, g = gh / H, h = gh % H;
    const int lane = threadIdx.x, n = lane & 31, hi = lane >> 5;
    const int k0 = kt * 32, my_k = k0 + n;
    const bool k_ok = my_k < T;
    const int kc = k_ok ? my_k : T - 1;
    const float* Q = reinterpret_cast<const float*>(p.q) + (int64_t)g * T * p.ldq + h * D;
    const float* K = reinterpret_cast<const float*>(p.k) + (int64_t)g * T * p.ldk + h * D;
    const float* V = reinterpret_cast<const float*>(p.v) + (int64_t)g * T * p.ldv + h * D;
    const float* dO = reinterpret_cast<const float*>(p.dout) + (int64_t)g * T * p.ldo + h * D;
    const TB* brow = reinterpret_cast<const TB*>(p.bias_t) + ((int64_t)gh * T + kc) * p.ld_bias + 16 * hi;
    float kh[D / 2], vh[D / 2];
    f32_half_row<D>(K, p.ldk, kc, hi, k_ok, p.scale, kh);
    f32_half_row<D>(V, p.ldv, kc, hi, k_ok, DROP ? p.inv_keep : 1.f, vh);
    uint64_t seed = 0;
    if (DROP) seed = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
    mobgt_f32x16 dk, dv;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
    for (int q0 = 0; q0 < T; q0 += 32) {
        mobgt_f32x16 s, dp;
        f32_bias16(brow + q0, s);
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[i] = 0.f;
        float qa[D / 2], da[D / 2];
        const int qr = q0 + kappa(n);
        f32_half_row<D>(Q, p.ldq, min(qr, T - 1), hi, qr < T, 1.f, qa);
        f32_half_row<D>(dO, p.ldo, min(qr, T - 1), hi, qr < T, 1.f, da);
        f32_nt<D>(qa, kh, s);
        f32_nt<D>(da, vh, dp);
        float x[16], ds[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q = q0 + 16 * hi + i;
            const bool live = q < T && k_ok;
            const int64_t qi = (int64_t)gh * T + min(q, T - 1);
            const float pr = live ? exp2f(fmaf(s[i], MOBGT_LOG2E, -p.lse_in[qi] * MOBGT_LOG2E)) : 0.f;
            bool keep = true;
            if (DROP) keep = attn_drop_keep(seed, dropout_row_hash(seed, (uint32_t)qi), (uint32_t)my_k, p.thr_s);
            x[i] = keep ? pr : 0.f;                                  // 1/(1-p) of dV is applied once, at the end
            ds[i] = fmaf(x[i], dp[i], -pr * p.delta[qi]);            // (dp carries 1/(1-p): it rode in on the V half row)
        }
        f32_tn<D>(dO, p.ldo, q0, T, n, hi, x, dv);
        f32_tn<D>(Q, p.ldq, q0, T, n, hi, ds, dk);
    }
    if (k_ok) {
        float* DK = reinterpret_cast<float*>(p.dk) + ((int64_t)g * T + my_k) * p.lddk + h * D + 16 * hi;
        float* DV = reinterpret_cast<float*>(p.dv) + ((int64_t)g * T + my_k) * p.lddv + h * D + 16 * hi;
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

template <int PASS_FWD, int D, typename TB, bool DROP>
hipError_t launch_f32_one(const AttnParams& p, hipStream_t st) {
    const int tiles = (p.T + 31) / 32;
    const dim3 grid((unsigned)(p.G * p.H * tiles)), block(64);
    if (PASS_FWD) {
        hipLaunchKernelGGL((attn_f32_fwd_kernel<D, TB, DROP>), grid, block, 0, st, p);
    } else {
        hipLaunchKernelGGL((attn_f32_dq_kernel<D, TB, DROP>), grid, block, 0, st, p);
        hipLaunchKernelGGL((attn_f32_dkv_kernel<D, TB, DROP>), grid, block, 0, st, p);
    }
    return hipGetLastError();
}

template <int PASS_FWD, typename TB>
int launch_f32(const AttnParams& p, int d, bool drop, hipStream_t st) {
#define MOBGT_F32_CASE(DD)                                                                                  \
    case DD: return (int)(drop ? launch_f32_one<PASS_FWD, DD, TB, true>(p, st) : launch_f32_one<PASS_FWD, DD, TB, false>(p, st));
    switch (d) {
        MOBGT_F32_CASE(16)
        MOBGT_F32_CASE(24)
        MOBGT_F32_CASE(32)
        default: return MOBGT_EBADDIM;
    }
#undef MOBGT_F32_CASE
}
