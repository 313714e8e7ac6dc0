// Developer micro-benchmark (GPU box): LDS-array cycles per wave-instruction for the LDS operations of build_bias_bwd.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_rate.hip -o tools/micro/lds_rate && tools/micro/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) unsigned lds_uint;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float tab[16384 + 128];
    for (int i = threadIdx.x; i < 16384 + 128; i += blockDim.x) tab[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // a different pseudo-random row per lane, changing per iteration
    unsigned key = (threadIdx.x * 2654435761u) >> 22;           // 0..1023
    const float v = 1.0f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            if (MODE == 0) __hip_atomic_fetch_add((lds_float*)&tab[h * 1025 + (key & 1023)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_fetch_add((lds_uint*)&tab[h * 1025 + (key & 1023)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) __hip_atomic_fetch_add((lds_float*)&tab[(threadIdx.x & ~63) + h * 1024 / 8 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // lane-linear
            if (MODE == 3) reinterpret_cast<volatile uint16_t*>(tab)[(threadIdx.x & ~63) * 2 + h * 64 + lane] = (uint16_t)it;                // ds_write_b16 lane-linear
            if (MODE == 4) reinterpret_cast<volatile float*>(tab)[(threadIdx.x & ~63) + h * 128 + lane] = v;                                // ds_write_b32
            if (MODE == 6) __hip_atomic_fetch_add((lds_u64*)&reinterpret_cast<unsigned long long*>(tab)[h * 1025 + (key & 1023)], 1ull << 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 7 && lane < 8) __hip_atomic_fetch_add((lds_float*)&tab[h * 1025 + (key & 1023)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 8 && (lane & 7) == 0) __hip_atomic_fetch_add((lds_float*)&tab[h * 1025 + (key & 1023)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 5) __hip_atomic_fetch_add((lds_float*)&tab[(key & 1023) * 8 + h], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // [row][head]
        }
        key = key * 1664525u + 1013904223u; key >>= 3;
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = tab[threadIdx.x];
}

template <int MODE> void run(const char* name, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double winstr = (double)iters * 8 * (threads / 64);
    printf("%-34s waves/CU %d: %.1f us, %.1f s_memtime-cycles (100 MHz) -> %.1f ns per wave-instruction per CU\n", name, threads / 64,
           ms * 1e3, (double)c, ms * 1e6 / winstr);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {256, 512}) {
        if (th == 256) { run<0>("ds_add_f32 [head][row] random", 256); run<1>("ds_add_u32 [head][row] random", 256); run<2>("ds_add_f32 lane-linear", 256);
                         run<3>("ds_write_b16 lane-linear", 256); run<4>("ds_write_b32 lane-linear", 256); run<5>("ds_add_f32 [row][head] random", 256); run<6>("ds_add_u64 [head][row] random", 256); run<7>("ds_add_f32 lanes 0-7 only", 256); run<8>("ds_add_f32 every 8th lane", 256); }
        else { run<0>("ds_add_f32 [head][row] random", 512); run<1>("ds_add_u32 [head][row] random", 512); run<2>("ds_add_f32 lane-linear", 512);
               run<3>("ds_write_b16 lane-linear", 512); run<4>("ds_write_b32 lane-linear", 512); run<5>("ds_add_f32 [row][head] random", 512); run<6>("ds_add_u64 [head][row] random", 512); run<7>("ds_add_f32 lanes 0-7 only", 512); run<8>("ds_add_f32 every 8th lane", 512); }
    }
    return 0;
}
