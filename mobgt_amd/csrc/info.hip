#include "mobgt_hip.h"
extern "C" int mobgt_abi_version(void) { return MOBGT_ABI_VERSION; }
extern "C" const char* mobgt_build_info(void) { return "libmobgt_hip gfx950 (MI355X), wave64, mfma_f32_32x32x16_bf16"; }

// Diagnostic: `workgroups` workgroups of `threads` threads that each hold a compute unit slot for `ticks_100mhz` ticks of the
// 100 MHz wall clock and do nothing else (s_sleep) -- a stand-in for another stream's persistent kernel (RCCL's footprint beside
// the step's kernels under data parallelism) in tests/test_gpu_train.py's co-residency test.  Writes nothing.
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace {
__global__ void occupy_kernel(long long ticks, int lds_bytes) {
    extern __shared__ unsigned char hold[];
    if (lds_bytes > 0 && threadIdx.x == 0) hold[0] = 1;             // (keeps the dynamic LDS request alive)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace
extern "C" int mobgt_debug_occupy(int workgroups, int threads, int lds_bytes, int64_t ticks_100mhz, void* stream) {
    if (workgroups <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || ticks_100mhz < 0)
        return MOBGT_EBADDIM;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(threads), lds_bytes, (hipStream_t)stream, (long long)ticks_100mhz, lds_bytes);
    return (int)hipGetLastError();
}
