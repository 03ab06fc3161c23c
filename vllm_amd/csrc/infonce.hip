// InfoNCE forward/backward kernels (placeholder until the fused kernels land in this file).
#include "common.h"
#include "topk_sweep.h"
int abc_fail_invalid(const char* msg);
extern "C" size_t abc_infonce_workspace_bytes(int B, int C, int d) { return 0; }
extern "C" int abc_infonce_fwd(const float*, const float*, float, float, int, int, int, float*, float*, int32_t*, void*, size_t, void*) {
    return abc_fail_invalid("abc_infonce_fwd: not built yet");
}
extern "C" int abc_infonce_bwd(const float*, const float*, const float*, float, float, const float*, int, int, int, int, float*, float*, float*, void*, size_t, void*) {
    return abc_fail_invalid("abc_infonce_bwd: not built yet");
}
