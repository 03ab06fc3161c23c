// Backward of the embedding epilogue (embed_head.hip) for gfx950 (MI355X), exact fp32 on the matrix cores: what autograd
// runs behind model/modeling_abc.py:148-154 in every training step,
//
//     pooled = get_mean_token_embed(ids, hidden, 0, instruction_mask)       model/abc_util.py:33-41
//     emb    = F.normalize(mlp_head(pooled).float(), dim=-1)                model/modeling_abc.py:16-34,151-154
//
// between the loss kernels (which hand back d emb) and the backbone (which wants d hidden).  With
//     a1 = x W1^T + b1,  h1 = selu(a1),  y = x + h1 W2^T + b2,  n = ||y||,  out = y / max(n, 1e-12):
//     dy  = (g - out (out . g)) / n                (n > 1e-12; g / 1e-12 below it; g itself without the normalise)
//     dW2 = dy^T h1      db2 = colsum(dy)      dh1 = dy W2
//     da1 = dh1 * selu'(a1),   selu'(a1) = scale (h1 > 0),  h1 + alpha scale (else: alpha scale exp(a1))
//     dW1 = da1^T x      db1 = colsum(da1)     dx  = dy + da1 W1
//     d hidden[b, s, :] = ((dx[b, :] / S) * [ids[b,s] != pad]) * instr[b,s]
// Four GEMMs (two "NN": contraction over the feature axis against a weight matrix read along its rows; two "TN":
// contraction over the batch axis) on ONE kernel template, v_mfma_f32_32x32x2_f32 with 64 x 64 tiles as nt_gemm.h, the
// element-wise pieces fused into the kernels that sum the split-K partials.  Fixed-order reductions everywhere.
#include "common.h"
#include "../../include/abcsearch.h"
#include "nt_gemm.h"

int abc_fail_hip(hipError_t e, const char* where);
int abc_fail_invalid(const char* msg);
int abc_fail_ws(size_t need, size_t have);

// part[s][m][n] = sum over the K range of split s of A(m, k) * B(k, n).
//   AK: A is stored [M][K] (k contiguous: 16-byte loads along k, K % 8 == 0), else [K][M] (lanes run along m)
//   BK: B is stored [N][K] (k contiguous), else [K][N] (lanes run along n)
template <bool AK, bool BK>
__global__ __launch_bounds__(512) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm, int M, int N,
                                                      int K, int splitk, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = (float*)smem_raw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_n = (N + 63) / 64;
    const int m0 = (blockIdx.x / tiles_n) * 64, n0 = (blockIdx.x % tiles_n) * 64;
    const int s = blockIdx.y;
    const int units = (K + 7) / 8;
    const int slice = s * 8 + wave, nslices = splitk * 8;
    const int u0 = (int)(((int64_t)slice * units) / nslices), u1 = (int)(((int64_t)(slice + 1) * units) / nslices);
    int ar[2], bc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        ar[b] = m0 + b * 32 + l31; if (ar[b] >= M) ar[b] = M - 1;       // clamped rows / columns are never stored
        bc[b] = n0 + b * 32 + l31; if (bc[b] >= N) bc[b] = N - 1;
    }
    v16f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll 2
    for (int u = u0; u < u1; ++u) {
        const int k0 = 8 * u + 4 * h;                                    // this lane half's four k of the unit
        v4f a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (AK) a[i] = *(const v4f*)(A + (size_t)ar[i] * K + k0);
            else {
#pragma unroll
                for (int t = 0; t < 4; ++t) a[i][t] = k0 + t < K ? A[(size_t)(k0 + t) * M + ar[i]] : 0.f;
            }
            if constexpr (BK) b[i] = *(const v4f*)(Bm + (size_t)bc[i] * K + k0);
            else {
#pragma unroll
                for (int t = 0; t < 4; ++t) b[i][t] = k0 + t < K ? Bm[(size_t)(k0 + t) * N + bc[i]] : 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
    float v[64];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[(i * 2 + j) * 16 + r] = acc[i][j][r];
    wg_tree_reduce<64>(v, lds, wave, lane);
    if (wave == 0) {
        float* out = part + (size_t)s * M * N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < M && col < N) out[(size_t)row * N + col] = v[(i * 2 + j) * 16 + r];
                }
            }
    }
}

// One workgroup per row: dy = (g - out (out . g)) / n  (F.normalize backward), or g when the head did not normalise.
__global__ __launch_bounds__(256) void head_bwd_dy_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                         const float* __restrict__ norm, int H, int normalize,
                                                         float* __restrict__ dy) {
    __shared__ float sh[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float* gr = g + (size_t)i * H;
    float* dr = dy + (size_t)i * H;
    if (!normalize) {
        for (int j = tid; j < H; j += 256) dr[j] = gr[j];
        return;
    }
    const float* orow = out + (size_t)i * H;
    float dot = 0.f;
    for (int j = tid; j < H; j += 256) dot += orow[j] * gr[j];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) dot += __shfl_xor(dot, m, 64);
    if ((tid & 63) == 0) sh[tid >> 6] = dot;
    __syncthreads();
    dot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    const float n = norm[i];
    if (n > 1e-12f) {
        for (int j = tid; j < H; j += 256) dr[j] = __fdiv_rn(gr[j] - orow[j] * dot, n);
    } else {                                           // the clamp is active: out = y / 1e-12, a plain scaling
        for (int j = tid; j < H; j += 256) dr[j] = __fdiv_rn(gr[j], 1e-12f);
    }
}

// out[j] = sum_i in[i][j], rows in order (one thread per column: coalesced across the columns)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, int R, int N, float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    float s = 0.f;
    for (int i = 0; i < R; ++i) s += in[(size_t)i * N + j];
    out[j] = s;
}

// da1 = (sum_s part[s]) * selu'(a1), from h1 = selu(a1)
__global__ __launch_bounds__(256) void head_bwd_da1_kernel(const float* __restrict__ part, int splitk, size_t bn,
                                                          const float* __restrict__ h1, float* __restrict__ da1) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= bn) return;
    float a = part[e];
    for (int s = 1; s < splitk; ++s) a += part[s * bn + e];
    const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
    const float hv = h1[e];
    da1[e] = a * (hv > 0.f ? scale : hv + alpha * scale);
}

// dx = dy + sum_s part[s]
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ part, int splitk, size_t bn,
                                                         const float* __restrict__ dy, float* __restrict__ dx) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= bn) return;
    float a = part[e];
    for (int s = 1; s < splitk; ++s) a += part[s * bn + e];
    dx[e] = dy[e] + a;
}

static int bwd_splitk(int M, int N, int K) {
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
    int sk = 256 / (tiles > 0 ? tiles : 1);
    if (sk < 1) sk = 1;
    if (sk > 16) sk = 16;
    const int units = (K + 7) / 8;
    while (sk > 1 && units / (sk * 8) < 1) sk >>= 1;
    return sk;
}

struct HeadBwdLayout {
    int sk_h1, sk_x;
    size_t off_dy, off_da1, off_part, total;
};

static HeadBwdLayout head_bwd_layout(int B, int H, int Hd) {
    HeadBwdLayout L;
    L.sk_h1 = bwd_splitk(B, Hd, H);
    L.sk_x = bwd_splitk(B, H, Hd);
    const size_t p1 = (size_t)L.sk_h1 * B * Hd * 4, p2 = (size_t)L.sk_x * B * H * 4;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    L.off_dy = take((size_t)B * H * 4);
    L.off_da1 = take((size_t)B * Hd * 4);
    L.off_part = take(p1 > p2 ? p1 : p2);
    L.total = o;
    return L;
}

extern "C" size_t abc_embed_head_bwd_workspace_bytes(int B, int H, int Hd) {
    if (B <= 0 || H <= 0 || Hd <= 0 || (H % 8) || (Hd % 8)) return 0;
    return head_bwd_layout(B, H, Hd).total;
}

template <bool AK, bool BK>
static hipError_t launch_gemm(const float* A, const float* Bm, int M, int N, int K, int sk, float* part, hipStream_t st) {
    auto fn = gemm_f32_kernel<AK, BK>;
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(((M + 63) / 64) * ((N + 63) / 64), sk), dim3(512), 65536, st, A, Bm, M, N, K, sk, part);
    return hipGetLastError();
}

extern "C" int abc_embed_head_bwd(const float* grad_out, const float* out, const float* norm, const float* x, const float* h1,
                                  const float* w1, const float* w2, int B, int H, int Hd, int normalize, float* dx, float* dw1,
                                  float* db1, float* dw2, float* db2, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || H <= 0 || Hd <= 0 || (H % 8) || (Hd % 8)) return abc_fail_invalid("embed_head_bwd: H and Hd must be positive multiples of 8");
    if (!grad_out || !x || !h1 || !w1 || !w2) return abc_fail_invalid("embed_head_bwd: null pointer");
    if (normalize && (!out || !norm)) return abc_fail_invalid("embed_head_bwd: the normalised head needs out and norm of the forward");
    if (((uintptr_t)grad_out | (uintptr_t)x | (uintptr_t)h1 | (uintptr_t)w1 | (uintptr_t)w2) & 15)
        return abc_fail_invalid("embed_head_bwd: operands must be 16-byte aligned");
    if (B == 0) return ABC_OK;
    const HeadBwdLayout L = head_bwd_layout(B, H, Hd);
    if (!ws || ws_bytes < L.total) return abc_fail_ws(L.total, ws_bytes);
    if ((uintptr_t)ws & 15) return abc_fail_invalid("embed_head_bwd: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float* dy = (float*)((char*)ws + L.off_dy);
    float* da1 = (float*)((char*)ws + L.off_da1);
    float* part = (float*)((char*)ws + L.off_part);
    const size_t bh = (size_t)B * H, bhd = (size_t)B * Hd;
    hipError_t e;
    hipLaunchKernelGGL(head_bwd_dy_kernel, dim3(B), dim3(256), 0, st, grad_out, out, norm, H, normalize, dy);
    if (db2) hipLaunchKernelGGL(colsum_kernel, dim3((H + 255) / 256), dim3(256), 0, st, dy, B, H, db2);
    // dW2[H, Hd] = dy^T h1: contraction over the batch, both operands read along their rows
    if (dw2 && (e = launch_gemm<false, false>(dy, h1, H, Hd, B, 1, dw2, st)) != hipSuccess) return abc_fail_hip(e, "embed_head_bwd dW2");
    const bool need_a1 = dx || dw1 || db1;
    if (need_a1) {
        // dh1[B, Hd] = dy[B, H] W2[H, Hd]: W2 read along its rows (lanes along Hd)
        if ((e = launch_gemm<true, false>(dy, w2, B, Hd, H, L.sk_h1, part, st)) != hipSuccess) return abc_fail_hip(e, "embed_head_bwd dh1");
        hipLaunchKernelGGL(head_bwd_da1_kernel, dim3((unsigned)((bhd + 255) / 256)), dim3(256), 0, st, part, L.sk_h1, bhd, h1, da1);
        if (db1) hipLaunchKernelGGL(colsum_kernel, dim3((Hd + 255) / 256), dim3(256), 0, st, da1, B, Hd, db1);
        if (dw1 && (e = launch_gemm<false, false>(da1, x, Hd, H, B, 1, dw1, st)) != hipSuccess) return abc_fail_hip(e, "embed_head_bwd dW1");
        if (dx) {
            // dx[B, H] = dy + da1[B, Hd] W1[Hd, H]
            if ((e = launch_gemm<true, false>(da1, w1, B, H, Hd, L.sk_x, part, st)) != hipSuccess) return abc_fail_hip(e, "embed_head_bwd dx");
            hipLaunchKernelGGL(head_bwd_dx_kernel, dim3((unsigned)((bh + 255) / 256)), dim3(256), 0, st, part, L.sk_x, bh, dy, dx);
        }
    }
    e = hipGetLastError();
    return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "embed_head_bwd launch");
}

// ------------------------------------------------------------------------------------------------
// d hidden[b, s, h] = ((dx[b, h] / S) * [ids[b,s] != pad]) * instr[b,s]      -- every position is written (zeros at pads)
template <bool BF16>
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ ids,
                                                           int64_t pad_id, const float* __restrict__ instr, int S, int H,
                                                           void* __restrict__ dhidden) {
    const int b = blockIdx.y;
    const int h0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (h0 >= H) return;
    float t[8];
    {
        const v4f x0 = *(const v4f*)(dx + (size_t)b * H + h0), x1 = *(const v4f*)(dx + (size_t)b * H + h0 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { t[i] = __fdiv_rn(x0[i], (float)S); t[4 + i] = __fdiv_rn(x1[i], (float)S); }
    }
    const int64_t* idr = ids + (size_t)b * S;
    const float* ir = instr ? instr + (size_t)b * S : nullptr;
    for (int s = 0; s < S; ++s) {
        const float m = idr[s] == pad_id ? 0.f : 1.f;
        const float w = ir ? ir[s] : 1.f;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (t[i] * m) * w;
        const size_t row = ((size_t)b * S + s) * H + h0;
        if constexpr (BF16) {
            v4i o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (int)(f32_to_bf16_bits(v[2 * i]) | (f32_to_bf16_bits(v[2 * i + 1]) << 16));
            *(v4i*)((uint16_t*)dhidden + row) = o;
        } else {
            *(v4f*)((float*)dhidden + row) = v4f{v[0], v[1], v[2], v[3]};
            *(v4f*)((float*)dhidden + row + 4) = v4f{v[4], v[5], v[6], v[7]};
        }
    }
}

static int mean_pool_bwd(bool bf16, const float* dx, const int64_t* ids, int64_t pad_id, const float* instr, int B, int S, int H,
                         void* dhidden, void* stream) {
    if (!dx || !ids || !dhidden) return abc_fail_invalid("mean_pool_bwd: null pointer");
    if (B < 0 || S <= 0 || H <= 0 || (H % 8) != 0) return abc_fail_invalid("mean_pool_bwd: need S > 0 and H a positive multiple of 8");
    if (((uintptr_t)dx | (uintptr_t)dhidden) & 15) return abc_fail_invalid("mean_pool_bwd: dx and dhidden must be 16-byte aligned");
    if (B == 0) return ABC_OK;
    const dim3 grid((H / 8 + 255) / 256, B);
    if (bf16) hipLaunchKernelGGL(mean_pool_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dx, ids, pad_id, instr, S, H, dhidden);
    else hipLaunchKernelGGL(mean_pool_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dx, ids, pad_id, instr, S, H, dhidden);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "mean_pool_bwd launch");
}

extern "C" int abc_mean_pool_bwd_bf16(const float* dx, const int64_t* ids, int64_t pad_id, const float* instr, int B, int S, int H,
                                      void* dhidden, void* stream) {
    return mean_pool_bwd(true, dx, ids, pad_id, instr, B, S, H, dhidden, stream);
}
extern "C" int abc_mean_pool_bwd_f32(const float* dx, const int64_t* ids, int64_t pad_id, const float* instr, int B, int S, int H,
                                     float* dhidden, void* stream) {
    return mean_pool_bwd(false, dx, ids, pad_id, instr, B, S, H, dhidden, stream);
}
