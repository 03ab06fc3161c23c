// Micro-benchmark behind DESIGN.md 8.1: does the ORDER of the 128 products of a phase change what the chip sustains?
// All operands in registers (random bf16 fragments loaded once), 4 waves per workgroup, one workgroup per CU, 256 AGPR
// accumulators per wave (64 tiles of 16x16), v_mfma_f32_16x16x32_bf16 tied in place.  Orders:
//   0  corpus tile outer, query tile inner, k-step outermost             (round 2 before the change)
//   1  the same with the query tiles back and forth (one operand changes per product: the product kernel's order)
//   2  k-steps innermost: both k-steps of a phase back to back on the SAME accumulator tile
//   3  order 2 with the query tiles back and forth between the pairs
//   4  quarter-steps of 4 corpus tiles: query tile outer, corpus tile inner, k-steps innermost; 5: corpus tiles back and forth
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=200000 experiments/mfma_order_bench.hip -o experiments/mfma_order_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(4))) int v4i;

__device__ __forceinline__ void mma(v4f& acc, const v4i& a, const v4i& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <int ORDER, int LDSR = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bench(const v4i* frags, float* out, int iters,
                                                                                       unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    __shared__ v4i lds[8192];                       // 128 KB of fragments for the LDS-read variant
    if (LDSR) { for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = frags[i % 2560]; __syncthreads(); }
    v4i sink = {0, 0, 0, 0};
    v4i a[2][16], b[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[k][i] = frags[(k * 16 + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) b[k][i] = frags[(32 + k * 4 + i) * 64 + lane];
    }
    v4f acc[16][4];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (ORDER <= 1) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int mb = 0; mb < 16; ++mb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int nb = (ORDER == 1 && (mb & 1)) ? 3 - j : j;
                        mma(acc[mb][nb], a[k][mb], b[k][nb]);
                    }
        } else if constexpr (ORDER >= 4 && LDSR > 0) {
            // order 5 with LDSR ds_read_b128 per phase spread between the pairs (what the kernel's fragment refills cost)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int mb = 4 * q + ((nb & 1) ? 3 - j : j);
                        mma(acc[mb][nb], a[0][mb], b[0][nb]);
                        mma(acc[mb][nb], a[1][mb], b[1][nb]);
                        const int pr = q * 16 + nb * 4 + j;
                        if (pr * LDSR / 64 != (pr + 1) * LDSR / 64) {
                            v4i r;
                            asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((unsigned)(((pr * 67 + (it & 31) * 64 + lane) & 8191) * 16)));
                            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                            sink ^= r;
                        }
                    }
        } else if constexpr (ORDER >= 4) {
            // quarter-steps of 4 corpus tiles: query tile outer, corpus tile inner (5: back and forth), k-steps innermost
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int mb = 4 * q + ((ORDER >= 5 && (nb & 1)) ? 3 - j : j);
                        const int kf = (ORDER == 6) ? (j & 1) : 0;      // 6: k-steps back and forth (changes the rounding order)
                        mma(acc[mb][nb], a[kf][mb], b[kf][nb]);
                        mma(acc[mb][nb], a[1 - kf][mb], b[1 - kf][nb]);
                    }
        } else {
#pragma unroll
            for (int mb = 0; mb < 16; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nb = (ORDER == 3 && (mb & 1)) ? 3 - j : j;
                    mma(acc[mb][nb], a[0][mb], b[0][nb]);
                    mma(acc[mb][nb], a[1][mb], b[1][nb]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = (float)(sink[0] ^ sink[1] ^ sink[2] ^ sink[3]) * 1e-30f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int zeros = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<unsigned short> h(2560 * 8);
    srand(1);
    for (auto& x : h) {   // unit-norm-like bf16: gaussian-ish magnitude 1/64, random sign and mantissa
        float f = zeros ? 0.f : ((rand() / (float)RAND_MAX) - 0.5f) * 0.06f;
        unsigned u; memcpy(&u, &f, 4);
        x = (unsigned short)(u >> 16);
    }
    v4i* d; float* out; unsigned long long* clk;
    hipMalloc(&d, h.size() * 2); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 16);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int order = 0; order < 9; ++order) {
            hipEventRecord(e0);
            if (order == 0) hipLaunchKernelGGL(bench<0>, dim3(256), dim3(256), 0, 0, d, out, iters, clk);
            if (order == 1) hipLaunchKernelGGL(bench<1>, dim3(256), dim3(256), 0, 0, d, out, iters, clk);
            if (order == 2) hipLaunchKernelGGL(bench<2>, dim3(256), dim3(256), 0, 0, d, out, iters, clk);
            if (order == 3) hipLaunchKernelGGL(bench<3>, dim3(256), dim3(256), 0, 0, d, out, iters, clk);
            if (order == 4) hipLaunchKernelGGL(bench<4>, dim3(256), dim3(256), 0, 0, d, out, iters, clk);
            if (order == 5) hipLaunchKernelGGL(bench<5>, dim3(256), dim3(256), 0, 0, d, out, iters, clk);
            if (order == 6) hipLaunchKernelGGL(bench<6>, dim3(256), dim3(256), 0, 0, d, out, iters, clk);
            if (order == 7) hipLaunchKernelGGL((bench<5, 40>), dim3(256), dim3(256), 0, 0, d, out, iters, clk);   // + 40 LDS reads / phase
            if (order == 8) hipLaunchKernelGGL((bench<5, 20>), dim3(256), dim3(256), 0, 0, d, out, iters, clk);   // + 20
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
            const double flop = 2.0 * 16 * 16 * 32 * 128.0 * iters * 1024.0;
            printf("rep %d order %d: %.2f ms  %.1f TFLOP/s  clock %.3f GHz  clocks/phase %.0f\n", rep, order, ms, flop / ms / 1e9,
                   0.1 * hc[0] / (double)hc[1], (double)hc[0] / iters);
        }
    return 0;
}
