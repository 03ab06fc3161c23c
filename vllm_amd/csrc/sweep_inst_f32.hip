// Instantiations of the two sweep kernels for one element type (see topk_sweep.hip: launch_sweep).
#include "sweep_kernels.h"

template <int DT, bool COUNT>
static hipError_t launch_one(bool four, const SweepParams& p, int grid, hipStream_t st) {
    // the > 64 KB dynamic-LDS attribute belongs to the (function, device) pair: set it at every launch for the
    // device that is current (a host-side field update; no process-wide "done" flag that a second device would miss)
    if (four) {
        auto kern4 = topk_sweep4_kernel<DT, COUNT>;
        hipError_t e = hipFuncSetAttribute((const void*)kern4, hipFuncAttributeMaxDynamicSharedMemorySize, S4_LDS);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern4, dim3(grid), dim3(256), S4_LDS, st, p);
        return hipGetLastError();
    }
    auto kern = topk_sweep_kernel<DT, COUNT>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, st, p);
    return hipGetLastError();
}

hipError_t launch_sweep_f32(bool count, bool four, const SweepParams& p, int grid, hipStream_t st) {
    return count ? launch_one<ABC_DT_F32, true>(four, p, grid, st) : launch_one<ABC_DT_F32, false>(four, p, grid, st);
}
