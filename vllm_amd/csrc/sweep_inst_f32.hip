// Instantiations of the sweep kernel for one element type (one translation unit per type: they build in parallel).
#include "sweep_kernels.h"

template <int DT, bool COUNT>
static hipError_t launch_one(const SweepParams& p, int grid, hipStream_t st) {
    // the > 64 KB dynamic-LDS attribute belongs to the (function, device) pair: set it at every launch for the
    // device that is current (a host-side field update; no process-wide "done" flag that a second device would miss)
    auto kern = topk_sweep16_kernel<DT, COUNT>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, S16_LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), S16_LDS, st, p);
    return hipGetLastError();
}

hipError_t launch_sweep_f32(bool count, const SweepParams& p, int grid, hipStream_t st) {
    return count ? launch_one<ABC_DT_F32, true>(p, grid, st) : launch_one<ABC_DT_F32, false>(p, grid, st);
}
