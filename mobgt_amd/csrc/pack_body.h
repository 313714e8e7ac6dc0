// MFMA-operand-order copies of bf16 weights (csrc/chain.hip reads a wave's B operand as one contiguous KB): the job table and
// the body of mobgt_pack_mfma_b, shared with csrc/smallgcn.hip, whose forward launch carries the pack as passenger workgroups
// (the category GCN keeps 19 compute units busy for 26 us; the pack is 11.7 us of pure data movement for the other 237).
#pragma once
#include "common.h"
#include "mobgt_hip.h"

namespace mobgt_pack {

constexpr int PACK_MAX = 96;                 // 12 layers x (4 forward + 4 transposed) weights in one launch

struct PackJobs {
    const uint16_t* src[PACK_MAX];
    uint16_t* dst[PACK_MAX];
    int N[PACK_MAX], K[PACK_MAX];
    int transposed[PACK_MAX];                      // src is [K][N] row-major: pack its transpose
    int first_block[PACK_MAX + 1];                 // job i owns (virtual) blocks [first_block[i], first_block[i + 1]); 256 items per block
};

// dst[((g S + s) 64 + l) * 8 + e] = src[(16 g + (l & 15)) K + 32 s + 8 (l >> 4) + e]: 16-byte pieces, writes contiguous.
// One virtual block = 256 items (threadIdx.x = the item); U virtual blocks vb, vb + stride, ... per call, every load of the
// U items requested before the first store (a passenger workgroup has four waves to itself: one item in flight per thread
// would be a round trip per 16 bytes).
template <int U>
__device__ __forceinline__ void pack_blocks(const PackJobs& jobs, int njobs, int vb0, int stride, int nvb) {
    uint4 lo[U], hi[U];
    uint4* out[U];
    bool two[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int vb = vb0 + u * stride;
        out[u] = nullptr;
        two[u] = false;
        if (vb >= nvb) continue;
        int job = 0;
        while (job + 1 < njobs && vb >= jobs.first_block[job + 1]) ++job;
        const int K = jobs.K[job], S = K / 32;
        const int64_t item = (int64_t)(vb - jobs.first_block[job]) * 256 + threadIdx.x;
        if (!jobs.transposed[job]) {
            const int64_t piece = item;
            if (piece >= (int64_t)jobs.N[job] * K / 8) continue;
            const int l = (int)(piece & 63);
            const int64_t gs = piece >> 6;
            const int g = (int)(gs / S), s_ = (int)(gs % S);
            lo[u] = *reinterpret_cast<const uint4*>(jobs.src[job] + (int64_t)(16 * g + (l & 15)) * K + 32 * s_ + 8 * (l >> 4));
            out[u] = reinterpret_cast<uint4*>(jobs.dst[job] + piece * 8);
        } else {
            // W'[n][k] = src[k][n]: a thread builds the pieces of TWO adjacent columns n (lanes j, j + 1) from eight 4-byte reads
            // -- a wave reads 64-byte runs of eight source rows -- and writes their 32 contiguous bytes
            const int64_t piece = 2 * item;
            if (piece >= (int64_t)jobs.N[job] * K / 8) continue;
            const int N = jobs.N[job];
            const int l = (int)(piece & 63);                                  // even
            const int64_t gs = piece >> 6;
            const int g = (int)(gs / S), s_ = (int)(gs % S);
            const uint16_t* from = jobs.src[job] + (int64_t)(32 * s_ + 8 * (l >> 4)) * N + 16 * g + (l & 15);
            uint32_t e[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = *reinterpret_cast<const uint32_t*>(from + (int64_t)i * N);
            lo[u].x = (e[0] & 0xffffu) | (e[1] << 16); lo[u].y = (e[2] & 0xffffu) | (e[3] << 16);
            lo[u].z = (e[4] & 0xffffu) | (e[5] << 16); lo[u].w = (e[6] & 0xffffu) | (e[7] << 16);
            hi[u].x = (e[0] >> 16) | (e[1] & 0xffff0000u); hi[u].y = (e[2] >> 16) | (e[3] & 0xffff0000u);
            hi[u].z = (e[4] >> 16) | (e[5] & 0xffff0000u); hi[u].w = (e[6] >> 16) | (e[7] & 0xffff0000u);
            out[u] = reinterpret_cast<uint4*>(jobs.dst[job] + piece * 8);
            two[u] = true;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (out[u]) {
            out[u][0] = lo[u];
            if (two[u]) out[u][1] = hi[u];
        }
    }
}

// host: validate the jobs and fill the table; returns 0 or an error code, *blocks = number of virtual blocks
inline int fill_jobs(PackJobs& jobs, int n, const void* const* src, void* const* dst, const int* N, const int* K, const int* transposed,
                     int* blocks_out) {
    if (n > PACK_MAX) return MOBGT_EBADDIM;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (N[i] <= 0 || K[i] <= 0 || (N[i] & 15) || (K[i] & 31)) return MOBGT_EBADDIM;
        if (((uintptr_t)src[i] | (uintptr_t)dst[i]) & 15) return MOBGT_EALIGN;
        if (transposed && transposed[i] && (N[i] & 1)) return MOBGT_EBADDIM;
        jobs.src[i] = (const uint16_t*)src[i]; jobs.dst[i] = (uint16_t*)dst[i]; jobs.N[i] = N[i]; jobs.K[i] = K[i];
        jobs.transposed[i] = transposed ? transposed[i] : 0;
        jobs.first_block[i] = blocks;
        const int64_t items = (int64_t)N[i] * K[i] / 8 / (jobs.transposed[i] ? 2 : 1);      // a transposed thread builds two pieces
        blocks += (int)((items + 255) / 256);
    }
    jobs.first_block[n] = blocks;
    *blocks_out = blocks;
    return 0;
}

}  // namespace mobgt_pack
