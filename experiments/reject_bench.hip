// Micro-benchmark behind DESIGN.md 4.1 (epilogue, quick reject): what do the 16 block tests of a tile cost a wave that has
// its SIMD to itself, and where do the clocks go?  256 AGPR "accumulators" per lane, no score passes (bound = +inf).
//   0  the product's shape: per block 16 v_accvgpr_read, 8 max, one compare, one branch
//   1  one branch per FOUR blocks (the four compares OR-ed in SGPRs)
//   2  one branch per SIXTEEN blocks
//   3  the reads alone            4  the max trees alone (on VGPRs)          5  compare + branch alone
//   6  128 MFMAs (16x16x32 bf16, 64 accumulator tiles twice) alone; 7  the same with two reads of ANOTHER tile's accumulators
//      and half a max3 behind every MFMA (= all 256 reads of variant 3 hidden behind one phase's products?)
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=200000 experiments/reject_bench.hip -o experiments/reject_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max2(float a, float b) {
    float r;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bench(float* out, int iters, float bound,
                                                                                       unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    float acc[256];
#pragma unroll
    for (int i = 0; i < 256; ++i) {
        const float x = (float)((lane * 131 + i * 17) % 251) * 0.001f;
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[i]) : "v"(x));
    }
    float th[4] = {bound, bound + 1.f, bound + 2.f, bound + 3.f};
    float sink = 0.f;
    float w[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) w[e] = (float)(lane + e);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        uint64_t any16 = 0;
#pragma unroll
        for (int mbg = 0; mbg < 4; ++mbg) {
            uint64_t any4 = 0;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float v[16];
                if constexpr (V != 4 && V != 5) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[e]) : "a"(acc[(mbg * 4 + nb) * 16 + e]));
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = w[e];
                }
                if constexpr (V == 3) {
                    asm volatile("" ::"v"(v[0]), "v"(v[15]));
                    continue;
                }
                float m;
                if constexpr (V == 5) m = v[0];
                else
                    m = max2(max3(max3(v[0], v[1], v[2]), max3(v[3], v[4], v[5]), max3(v[6], v[7], v[8])),
                             max3(max3(v[9], v[10], v[11]), max3(v[12], v[13], v[14]), v[15]));
                if constexpr (V == 4) {
                    asm volatile("" ::"v"(m));
                    continue;
                }
                const uint64_t b = __builtin_amdgcn_ballot_w64(m > th[nb]);
                if constexpr (V == 0 || V == 5) {
                    if (b == 0) continue;
                    asm volatile("");
                    sink += m;                                   // the cold path
                    out[lane] = sink;
                } else {
                    any4 |= b;
                }
            }
            if constexpr (V == 1) {
                if (any4 == 0) continue;
                asm volatile("");
                sink += 1.f;
                out[lane] = sink;
            }
            any16 |= any4;
        }
        if constexpr (V == 2) {
            if (any16 != 0) {
                asm volatile("");
                sink += 1.f;
                out[lane] = sink;
            }
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = sink;
#pragma unroll
    for (int i = 0; i < 256; i += 37) {
        float x;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[i]));
        s += x;
    }
    out[256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(4))) int v4i;
template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bench_mfma(float* out, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    v4f acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = (float)((lane * 131 + i * 17 + j) % 251) * 0.001f;
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[i][j]) : "v"(x));
        }
    v4i a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = v4i{0x3c003c00 + lane * 77 + i, 0x3b803b80 + i, 0x3c103c10, 0x3a003a00 + lane};
        b[i] = v4i{0x3c203c00 + lane * 31 + i, 0x3b903b80 + i, 0x3c303c10, 0x3a103a00 + lane};
    }
    float sink = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
                if constexpr (V == 7) {
                    const int o = (i + 32) & 63;
                    float x0, x1;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(acc[o][2 * r]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(acc[o][2 * r + 1]));
                    if (i & 1) sink = max3(sink, x0, x1); else sink = max2(sink, x0);
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = sink;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        float x;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[i][1]));
        s += x;
    }
    out[256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int V>
static void run_mfma(float* out, unsigned long long* clk, int iters) {
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(bench_mfma<V>, dim3(256), dim3(256), 0, 0, out, iters, clk);
        (void)hipDeviceSynchronize();
    }
    unsigned long long h = 0;
    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("variant %d: %.1f core clocks per 128 MFMAs\n", V, (double)h / iters);
}

template <int V>
static void run(float* out, unsigned long long* clk, int iters) {
    hipLaunchKernelGGL(bench<V>, dim3(256), dim3(256), 0, 0, out, iters, 1e30f, clk);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(bench<V>, dim3(256), dim3(256), 0, 0, out, iters, 1e30f, clk);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("variant %d: %.1f core clocks per tile (16 blocks)\n", V, (double)h / iters);
}

int main() {
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, 4096 * 4);
    hipMalloc(&clk, 8);
    const int iters = 20000;
    run<0>(out, clk, iters);
    run<1>(out, clk, iters);
    run<2>(out, clk, iters);
    run<3>(out, clk, iters);
    run<4>(out, clk, iters);
    run<5>(out, clk, iters);
    run_mfma<6>(out, clk, iters);
    run_mfma<7>(out, clk, iters);
    return 0;
}
