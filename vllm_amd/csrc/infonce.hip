// Temperature-scaled, label-smoothed InfoNCE forward / backward for gfx950 (MI355X), exact fp32.
//
// The problem is small (B=128 x C=1024 x D=4096 per rank: ~1 GFLOP) and latency bound, so the design
// goal is few launches that each fill the chip, not MFMA tuning:
//   fwd : logits_nt (fp32 MFMA 32x32x2, 64x64 tiles, split-K over workgroups and over the 8 waves of a
//         workgroup) -> row_stats (sum the K partials, scale by 1/temp, online max/argmax/logsumexp,
//         label-smoothed row loss; keeps z for the backward) -> reduce_rows (mean loss)
//   bwd : grad_logits (G = softmax - target, scaled) -> gemm_nn twice (dq = G p, dp_local = G^T q)
//         -> reduce_rows (d_temp)
// All reductions use fixed orders: results are bitwise reproducible run to run.
#include "common.h"
#include "../../include/abcsearch.h"
#include "nt_gemm.h"

int abc_fail_hip(hipError_t e, const char* where);
int abc_fail_invalid(const char* msg);
int abc_fail_ws(size_t need, size_t have);

#define NCE_SPLITK_MAX 16

struct NceLayout {
    int splitk;
    size_t off_z, off_part, off_gs, off_rowloss, off_rowdt, total;
};

static NceLayout nce_layout(int B, int C, int d) {
    NceLayout L;
    const int tiles = ((B + 63) / 64) * ((C + 63) / 64);
    int sk = 256 / (tiles > 0 ? tiles : 1);
    if (sk < 1) sk = 1;
    if (sk > NCE_SPLITK_MAX) sk = NCE_SPLITK_MAX;
    const int units = d / 8;                       // K is consumed in units of 8 floats per wave step
    while (sk > 1 && units / (sk * 8) < 1) sk >>= 1;
    L.splitk = sk;
    size_t o = 0;
    const size_t bc = (size_t)B * C;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    L.off_z = take(bc * 4);
    L.off_part = take(bc * 4 * sk);
    L.off_gs = take(bc * 4);
    L.off_rowloss = take((size_t)B * 4);
    L.off_rowdt = take((size_t)B * 4);
    L.total = o;
    return L;
}

extern "C" size_t abc_infonce_workspace_bytes(int B, int C, int d) {
    if (B <= 0 || C < B || d <= 0 || (d % 8) != 0) return 0;
    return nce_layout(B, C, d).total;
}

// ------------------------------------------------------------------------------------------------
// fixed-order block reductions (256 threads)
__device__ __forceinline__ float block_sum(float x, float* sh, int tid) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) x += __shfl_xor(x, m, 64);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = x;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// One workgroup per query row: z = inv_temp * sum_s part[s]; max/argmax/logsumexp; row loss.
__global__ __launch_bounds__(256) void nce_row_stats_kernel(const float* __restrict__ part, int splitk, int B, int C,
                                                           const float* __restrict__ temp, float eps, float* __restrict__ z,
                                                           float* __restrict__ lse, int* __restrict__ argmax,
                                                           float* __restrict__ row_loss) {
    __shared__ float sh[8];
    __shared__ int shi[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const size_t bc = (size_t)B * C;
    const float t = temp[0];
    float mx = -INFINITY, zsum = 0.f, zii = 0.f;
    int amx = 0x7FFFFFFF;
    for (int j = tid; j < C; j += 256) {
        float acc = part[(size_t)i * C + j];
        for (int s = 1; s < splitk; ++s) acc += part[s * bc + (size_t)i * C + j];
        const float zz = __fdiv_rn(acc, t);          // score / temp, as Temperature.forward
        z[(size_t)i * C + j] = zz;
        if (zz > mx) { mx = zz; amx = j; }          // j increases per thread: first maximum kept
        zsum += zz;
        if (j == i) zii = zz;
    }
    // max + first argmax across the block
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const float om = __shfl_xor(mx, m, 64);
        const int oa = __shfl_xor(amx, m, 64);
        if (om > mx || (om == mx && oa < amx)) { mx = om; amx = oa; }
    }
    if ((tid & 63) == 0) { sh[4 + (tid >> 6)] = mx; shi[tid >> 6] = amx; }
    __syncthreads();
    float bm = sh[4]; int ba = shi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) if (sh[4 + w] > bm || (sh[4 + w] == bm && shi[w] < ba)) { bm = sh[4 + w]; ba = shi[w]; }
    float se = 0.f;
    for (int j = tid; j < C; j += 256) se += expf(z[(size_t)i * C + j] - bm);
    se = block_sum(se, sh, tid);
    zsum = block_sum(zsum, sh, tid);
    zii = block_sum(zii, sh, tid);
    if (tid == 0) {
        const float l = bm + logf(se);
        lse[i] = l;
        argmax[i] = ba;
        // (1-eps)*(-logp_ii) + (eps/C)*sum_j(-logp_ij), logp = z - lse
        row_loss[i] = (1.f - eps) * (l - zii) + eps * (l - zsum / (float)C);
    }
}

// out[0] = scale * sum_i x[i], fixed order (one workgroup).
__global__ __launch_bounds__(256) void nce_reduce_rows_kernel(const float* __restrict__ x, int n, float scale,
                                                             const float* __restrict__ neg_div, float* __restrict__ out) {
    __shared__ float sh[8];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += x[i];
    a = block_sum(a, sh, threadIdx.x);
    if (threadIdx.x == 0) out[0] = neg_div ? -__fdiv_rn(a, neg_div[0]) : a * scale;
}

// One workgroup per row: GS = (softmax(z) - target) * (go/B) / temp ; row_dt = sum_j G*z (G not divided by temp)
__global__ __launch_bounds__(256) void nce_grad_logits_kernel(const float* __restrict__ z, const float* __restrict__ lse,
                                                             const float* __restrict__ grad_out, int B, int C,
                                                             const float* __restrict__ temp, float eps, float* __restrict__ gs,
                                                             float* __restrict__ row_dt) {
    __shared__ float sh[8];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float go = grad_out[0] / (float)B;
    const float t = temp[0];
    const float l = lse[i];
    const float smooth = eps / (float)C;
    float dt = 0.f;
    for (int j = tid; j < C; j += 256) {
        const float zz = z[(size_t)i * C + j];
        const float g = (expf(zz - l) - (smooth + (j == i ? 1.f - eps : 0.f))) * go;
        gs[(size_t)i * C + j] = __fdiv_rn(g, t);
        dt += g * zz;
    }
    dt = block_sum(dt, sh, tid);
    if (tid == 0) row_dt[i] = dt;
}

// ------------------------------------------------------------------------------------------------
// out[M,N] = A' @ Bm,  Bm [K,N] row-major;  A' = A [M,K] row-major (TRANS_A = false) or A^T with A [K,lda>=M]
// (TRANS_A = true).  One workgroup = one 32 x (32*NB) output tile, K split over its 8 waves.
template <int NB, bool TRANS_A>
__global__ __launch_bounds__(512) void nce_gemm_nn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm,
                                                         int M, int N, int Kd, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = (float*)smem_raw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int TN = 32 * NB;
    const int tiles_n = (N + TN - 1) / TN;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * TN;
    const int units = (Kd + 7) / 8;
    const int u0 = (int)(((int64_t)wave * units) / 8), u1 = (int)(((int64_t)(wave + 1) * units) / 8);
    int am = m0 + l31; if (am >= M) am = M - 1;
    int bn = n0 + NB * l31; if (bn + NB > N) bn = N - NB;        // N % NB == 0 is checked on the host

    v16f acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    for (int u = u0; u < u1; ++u) {
        const int kb = 8 * u + 4 * h;
        float a[4];
        float b[4][NB];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = kb + t;
            const bool ok = k < Kd;
            const int kc = ok ? k : Kd - 1;
            a[t] = TRANS_A ? A[(size_t)kc * lda + am] : A[(size_t)am * lda + kc];
            if (!ok) a[t] = 0.f;                 // zero A kills the product; B may hold anything finite
            const float* bp = Bm + (size_t)kc * N + bn;
            if constexpr (NB == 4) {
                const v4f x = *(const v4f*)bp;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[t][j] = x[j];
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j) b[t][j] = bp[j];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t][j], acc[j], 0, 0, 0);
    }
    float v[16 * NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[j * 16 + r] = acc[j][r];
    wg_tree_reduce<16 * NB>(v, lds, wave, lane);
    if (wave == 0) {
        const int nb0 = n0 + NB * l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M && nb0 + NB <= N && bn == nb0) {
#pragma unroll
                for (int j = 0; j < NB; ++j) out[(size_t)row * N + nb0 + j] = v[j * 16 + r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
static int nce_check(int B, int C, int d) {
    if (B <= 0 || C < B || d <= 0) return abc_fail_invalid("infonce: need 0 < B <= C and d > 0");
    if (d % 8 != 0) return abc_fail_invalid("infonce: d must be a multiple of 8");
    return ABC_OK;
}

extern "C" int abc_infonce_fwd(const float* q, const float* p, const float* temp, float eps, int B, int C, int d,
                               float* out_loss, float* out_lse, int32_t* out_argmax, void* ws, size_t ws_bytes,
                               void* stream) {
    int rc = nce_check(B, C, d);
    if (rc) return rc;
    if (!q || !p || !temp || !out_loss || !out_lse || !out_argmax) return abc_fail_invalid("infonce_fwd: null pointer");
    if (((uintptr_t)q | (uintptr_t)p) & 15) return abc_fail_invalid("infonce_fwd: q and p must be 16-byte aligned");
    const NceLayout L = nce_layout(B, C, d);
    if (!ws || ws_bytes < L.total) return abc_fail_ws(L.total, ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    float* z = (float*)(w + L.off_z);
    float* part = (float*)(w + L.off_part);
    float* row_loss = (float*)(w + L.off_rowloss);
    const int tiles = ((B + 63) / 64) * ((C + 63) / 64);
    hipError_t e = hipFuncSetAttribute((const void*)nt_gemm_partials_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) return abc_fail_hip(e, "infonce_fwd attr");
    hipLaunchKernelGGL(nt_gemm_partials_kernel<0>, dim3(tiles, L.splitk), dim3(512), 65536, st, q, p, B, C, d, L.splitk, part);
    hipLaunchKernelGGL(nce_row_stats_kernel, dim3(B), dim3(256), 0, st, part, L.splitk, B, C, temp, eps, z, out_lse,
                       out_argmax, row_loss);
    hipLaunchKernelGGL(nce_reduce_rows_kernel, dim3(1), dim3(256), 0, st, row_loss, B, 1.0f / (float)B, (const float*)nullptr, out_loss);
    e = hipGetLastError();
    return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "infonce_fwd launch");
}

template <bool TRANS_A>
static hipError_t launch_gemm_nn(const float* A, int lda, const float* Bm, int M, int N, int Kd, float* out, hipStream_t st) {
    const int tiles4 = ((M + 31) / 32) * ((N + 127) / 128);
    const bool nb4 = (N % 4 == 0) && tiles4 >= 192;
    if (nb4) {
        auto k = nce_gemm_nn_kernel<4, TRANS_A>;
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(tiles4), dim3(512), 65536, st, A, lda, Bm, M, N, Kd, out);
    } else if (N % 2 == 0) {
        auto k = nce_gemm_nn_kernel<2, TRANS_A>;
        const int tiles = ((M + 31) / 32) * ((N + 63) / 64);
        hipLaunchKernelGGL(k, dim3(tiles), dim3(512), 32768, st, A, lda, Bm, M, N, Kd, out);
    } else {
        auto k = nce_gemm_nn_kernel<1, TRANS_A>;
        const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
        hipLaunchKernelGGL(k, dim3(tiles), dim3(512), 16384, st, A, lda, Bm, M, N, Kd, out);
    }
    return hipGetLastError();
}

extern "C" int abc_infonce_bwd(const float* q, const float* p, const float* lse, const float* temp, float eps,
                               const float* grad_out, int B, int C, int d, int local_rows, float* dq, float* dp_local,
                               float* d_temp, void* ws, size_t ws_bytes, void* stream) {
    int rc = nce_check(B, C, d);
    if (rc) return rc;
    if (!q || !p || !lse || !temp || !grad_out) return abc_fail_invalid("infonce_bwd: null pointer");
    if (local_rows < 0 || local_rows > C) return abc_fail_invalid("infonce_bwd: local_rows out of range");
    if (((uintptr_t)q | (uintptr_t)p) & 15) return abc_fail_invalid("infonce_bwd: q and p must be 16-byte aligned");
    const NceLayout L = nce_layout(B, C, d);
    if (!ws || ws_bytes < L.total) return abc_fail_ws(L.total, ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    const float* z = (const float*)(w + L.off_z);        // logits kept by abc_infonce_fwd in the same workspace
    float* gs = (float*)(w + L.off_gs);
    float* row_dt = (float*)(w + L.off_rowdt);
    hipLaunchKernelGGL(nce_grad_logits_kernel, dim3(B), dim3(256), 0, st, z, lse, grad_out, B, C, temp, eps, gs, row_dt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return abc_fail_hip(e, "infonce_bwd grad_logits");
    if (dq) {
        e = launch_gemm_nn<false>(gs, C, p, B, d, C, dq, st);                  // dq[B,d] = GS[B,C] @ p[C,d]
        if (e != hipSuccess) return abc_fail_hip(e, "infonce_bwd dq");
    }
    if (dp_local && local_rows > 0) {
        e = launch_gemm_nn<true>(gs, C, q, local_rows, d, B, dp_local, st);    // dp[L,d] = GS[:, :L]^T @ q[B,d]
        if (e != hipSuccess) return abc_fail_hip(e, "infonce_bwd dp");
    }
    if (d_temp) {
        hipLaunchKernelGGL(nce_reduce_rows_kernel, dim3(1), dim3(256), 0, st, row_dt, B, 0.0f, temp, d_temp);
        e = hipGetLastError();
        if (e != hipSuccess) return abc_fail_hip(e, "infonce_bwd dtemp");
    }
    return ABC_OK;
}
