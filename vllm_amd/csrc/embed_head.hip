// Embedding epilogue for gfx950 (MI355X): the step immediately before the search / loss path.
//
//   mean pool   model/abc_util.py:33-41       mean over ALL S positions of (ids != pad) * instr * hidden
//                                             (divides by the sequence length, not the token count)
//   MLP head    model/modeling_abc.py:16-34   out = x + W2 selu(W1 x + b1) + b2          (fp32)
//   normalise   model/modeling_abc.py:151-154 F.normalize(.float(), dim=-1): x / max(||x||_2, 1e-12)
//   + the corpus forms the search kernels contract in: bf16 (RNE) and int8 with per-vector scale
//     (SURVEY.md 8c rule), produced from the fp32 result in the same pass.
//
// mean pool: HBM-bound (B*S_valid*H*elem bytes; rows of padded positions are never read).
// MLP: two exact-fp32 NT GEMMs on the matrix cores (nt_gemm.h) with the bias/SELU and the
// bias/residual/normalise/quantise epilogues fused into the kernels that sum the split-K partials.
#include "common.h"
#include "../../include/abcsearch.h"
#include "nt_gemm.h"

int abc_fail_hip(hipError_t e, const char* where);
int abc_fail_invalid(const char* msg);
int abc_fail_ws(size_t need, size_t have);

// ------------------------------------------------------------------------------------------------
// out[b, h] = (1/S) * sum_s [ids[b,s] != pad] * instr[b,s] * hidden[b,s,h]
// grid (ceil(H / (256*8)), B); each thread owns 8 consecutive h (bf16: one 16-byte load per row).
template <bool BF16>
__global__ __launch_bounds__(256) void mean_pool_kernel(const void* __restrict__ hidden, const int64_t* __restrict__ ids,
                                                       int64_t pad_id, const float* __restrict__ instr, int S, int H,
                                                       int round_bf16, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int h0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (h0 >= H) return;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    const int64_t* idr = ids + (size_t)b * S;
    const float* ir = instr ? instr + (size_t)b * S : nullptr;
    // eight positions at a time: their token ids and weights first, then every row load that is wanted (uniform over the
    // workgroup), then the adds in sequence order -- taken one position at a time each row load waited for its id
    for (int s0 = 0; s0 < S; s0 += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + u;
            const bool on = s < S && idr[s < S ? s : S - 1] != pad_id;
            w[u] = on ? (ir ? ir[s] : 1.0f) : 0.0f;
        }
        if constexpr (BF16) {
            v4i raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (w[u] != 0.0f) raw[u] = *(const v4i*)((const uint16_t*)hidden + ((size_t)b * S + s0 + u) * H + h0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (w[u] == 0.0f) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo = __uint_as_float((uint32_t)raw[u][i] << 16), hi = __uint_as_float((uint32_t)raw[u][i] & 0xFFFF0000u);
                    if (round_bf16 && w[u] != 1.0f) {            // instruction_mask * hidden is a bf16 product in the reference
                        lo = __uint_as_float(f32_to_bf16_bits(lo * w[u]) << 16);
                        hi = __uint_as_float(f32_to_bf16_bits(hi * w[u]) << 16);
                        acc[2 * i] += lo;
                        acc[2 * i + 1] += hi;
                    } else {
                        acc[2 * i] += w[u] * lo;
                        acc[2 * i + 1] += w[u] * hi;
                    }
                }
            }
        } else {
            v4f x0[8], x1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (w[u] != 0.0f) {
                    const float* src = (const float*)hidden + ((size_t)b * S + s0 + u) * H + h0;
                    x0[u] = *(const v4f*)src;
                    x1[u] = *(const v4f*)(src + 4);
                }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (w[u] == 0.0f) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] += w[u] * x0[u][i];
                    acc[4 + i] += w[u] * x1[u][i];
                }
            }
        }
    }
    const float inv = 1.0f / (float)S;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float m = __fdiv_rn(acc[i], (float)S);
        (void)inv;
        if (BF16 && round_bf16) m = __uint_as_float(f32_to_bf16_bits(m) << 16);
        out[(size_t)b * H + h0 + i] = m;
    }
}

static int mean_pool(bool bf16, const void* hidden, const int64_t* ids, int64_t pad_id, const float* instr, int B, int S,
                     int H, int round_bf16, float* out, void* stream) {
    if (!hidden || !ids || !out) return abc_fail_invalid("mean_pool: null pointer");
    if (B < 0 || S <= 0 || H <= 0 || (H % 8) != 0) return abc_fail_invalid("mean_pool: need S > 0 and H a positive multiple of 8");
    if (((uintptr_t)hidden | (uintptr_t)out) & 15) return abc_fail_invalid("mean_pool: hidden and out must be 16-byte aligned");
    if (B == 0) return ABC_OK;
    const dim3 grid((H / 8 + 255) / 256, B);
    if (bf16) hipLaunchKernelGGL(mean_pool_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, hidden, ids, pad_id, instr, S, H, round_bf16, out);
    else hipLaunchKernelGGL(mean_pool_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, hidden, ids, pad_id, instr, S, H, round_bf16, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "mean_pool launch");
}

extern "C" int abc_mean_pool_bf16(const void* hidden, const int64_t* ids, int64_t pad_id, const float* instr, int B, int S,
                                  int H, int round_bf16, float* out, void* stream) {
    return mean_pool(true, hidden, ids, pad_id, instr, B, S, H, round_bf16, out, stream);
}
extern "C" int abc_mean_pool_f32(const void* hidden, const int64_t* ids, int64_t pad_id, const float* instr, int B, int S,
                                 int H, float* out, void* stream) {
    return mean_pool(false, hidden, ids, pad_id, instr, B, S, H, 0, out, stream);
}

// ------------------------------------------------------------------------------------------------
// h1[i][j] = selu(sum_s part[s][i][j] + b1[j])
__global__ __launch_bounds__(256) void head_selu_kernel(const float* __restrict__ part, int splitk, int B, int N,
                                                       const float* __restrict__ bias, float* __restrict__ out) {
    const size_t bn = (size_t)B * N;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= bn) return;
    float a = part[e];
    for (int s = 1; s < splitk; ++s) a += part[s * bn + e];
    if (bias) a += bias[e % N];
    const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
    out[e] = a > 0.f ? scale * a : (expf(a) - 1.0f) * (alpha * scale);
}

// One workgroup per row: y = x + (sum_s part[s] + b2); out = y / max(||y||, 1e-12); optional bf16 / int8 forms.
__global__ __launch_bounds__(256) void head_finish_kernel(const float* __restrict__ part, int splitk, int B, int H,
                                                         const float* __restrict__ bias, const float* __restrict__ x,
                                                         int normalize, float* __restrict__ out_f32,
                                                         uint16_t* __restrict__ out_bf16, int8_t* __restrict__ out_i8,
                                                         float* __restrict__ out_scale, float* __restrict__ out_norm) {
    __shared__ float sh[8];
    const int i = blockIdx.x, tid = threadIdx.x;
    const size_t bh = (size_t)B * H;
    float ss = 0.f;
    for (int j = tid; j < H; j += 256) {
        const size_t e = (size_t)i * H + j;
        float a = part[e];
        for (int s = 1; s < splitk; ++s) a += part[s * bh + e];
        if (bias) a += bias[j];
        const float y = x[e] + a;
        out_f32[e] = y;
        ss += y * y;
    }
    // fixed-order block sum
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if ((tid & 63) == 0) sh[tid >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3]));
    if (out_norm && tid == 0) out_norm[i] = nrm;                   // ||y||: what the backward of the normalise needs
    const float den = normalize ? fmaxf(nrm, 1e-12f) : 1.0f;
    float amax = 0.f;
    for (int j = tid; j < H; j += 256) {
        const size_t e = (size_t)i * H + j;
        const float y = __fdiv_rn(out_f32[e], den);
        out_f32[e] = y;
        if (out_bf16) out_bf16[e] = (uint16_t)f32_to_bf16_bits(y);
        amax = fmaxf(amax, fabsf(y));
    }
    if (out_i8) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
        __syncthreads();
        if ((tid & 63) == 0) sh[4 + (tid >> 6)] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(sh[4], sh[5]), fmaxf(sh[6], sh[7]));
        const float scale = amax > 0.f ? __fdiv_rn(amax, 127.0f) : 1.0f;
        for (int j = tid; j < H; j += 256) {
            const size_t e = (size_t)i * H + j;
            float r = rintf(__fdiv_rn(out_f32[e], scale));
            r = fminf(fmaxf(r, -127.f), 127.f);
            out_i8[e] = (int8_t)r;
        }
        if (tid == 0) out_scale[i] = scale;
    }
}

struct HeadLayout {
    int sk1, sk2;
    size_t off_part, off_h1, total;
};

static int pick_splitk(int M, int N, int K) {
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
    int sk = 256 / (tiles > 0 ? tiles : 1);
    if (sk < 1) sk = 1;
    if (sk > 16) sk = 16;
    const int units = K / 8;
    while (sk > 1 && units / (sk * 8) < 1) sk >>= 1;
    return sk;
}

static HeadLayout head_layout(int B, int H, int Hd) {
    HeadLayout L;
    L.sk1 = pick_splitk(B, Hd, H);
    L.sk2 = pick_splitk(B, H, Hd);
    const size_t p1 = (size_t)L.sk1 * B * Hd * 4, p2 = (size_t)L.sk2 * B * H * 4;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    L.off_part = take(p1 > p2 ? p1 : p2);
    L.off_h1 = take((size_t)B * Hd * 4);
    L.total = o;
    return L;
}

extern "C" size_t abc_embed_head_workspace_bytes(int B, int H, int Hd) {
    if (B <= 0 || H <= 0 || Hd <= 0 || (H % 8) || (Hd % 8)) return 0;
    return head_layout(B, H, Hd).total;
}

extern "C" int abc_embed_head_f32(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                  int B, int H, int Hd, int normalize, float* out_f32, void* out_bf16, int8_t* out_i8,
                                  float* out_scale, float* out_h1, float* out_norm, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || H <= 0 || Hd <= 0 || (H % 8) || (Hd % 8)) return abc_fail_invalid("embed_head: H and Hd must be positive multiples of 8");
    if (!x || !w1 || !w2 || !out_f32) return abc_fail_invalid("embed_head: null pointer");
    if ((out_i8 == nullptr) != (out_scale == nullptr)) return abc_fail_invalid("embed_head: out_i8 and out_scale go together");
    if (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2) & 15) return abc_fail_invalid("embed_head: x, w1, w2 must be 16-byte aligned");
    if (B == 0) return ABC_OK;
    const HeadLayout L = head_layout(B, H, Hd);
    if (!ws || ws_bytes < L.total) return abc_fail_ws(L.total, ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)((char*)ws + L.off_part);
    float* h1 = out_h1 ? out_h1 : (float*)((char*)ws + L.off_h1);      // training keeps selu(x W1^T + b1) for the backward
    auto gemm = nt_gemm_partials_kernel<1>;
    hipError_t e = hipFuncSetAttribute((const void*)gemm, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) return abc_fail_hip(e, "embed_head attr");
    // h1 = selu(x W1^T + b1)            x [B,H], W1 [Hd,H]
    hipLaunchKernelGGL(gemm, dim3(((B + 63) / 64) * ((Hd + 63) / 64), L.sk1), dim3(512), 65536, st, x, w1, B, Hd, H, L.sk1, part);
    hipLaunchKernelGGL(head_selu_kernel, dim3((unsigned)(((size_t)B * Hd + 255) / 256)), dim3(256), 0, st, part, L.sk1, B, Hd, b1, h1);
    // y = x + h1 W2^T + b2 ; normalise  h1 [B,Hd], W2 [H,Hd]
    hipLaunchKernelGGL(gemm, dim3(((B + 63) / 64) * ((H + 63) / 64), L.sk2), dim3(512), 65536, st, h1, w2, B, H, Hd, L.sk2, part);
    hipLaunchKernelGGL(head_finish_kernel, dim3(B), dim3(256), 0, st, part, L.sk2, B, H, b2, x, normalize, out_f32,
                       (uint16_t*)out_bf16, out_i8, out_scale, out_norm);
    e = hipGetLastError();
    return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "embed_head launch");
}
