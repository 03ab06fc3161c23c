// Temperature-scaled, label-smoothed InfoNCE forward / backward for gfx950 (MI355X), exact fp32.
//
// The problem is small (B=128 x C=1024 x D=4096 per rank: ~1 GFLOP forward, ~1.2 backward, 19 MB) and latency
// bound, so the design is ONE kernel per direction, each filling the chip, with the row statistics and the final
// scalars produced inside the same launch by the workgroup that arrives last (agent-scope release / acquire on
// small counters; no grid barrier, nothing assumed about placement):
//
//   fwd  nce_fwd_kernel: one workgroup per 32 x 32 logits tile, the whole K range split over its 8 waves (fp32 MFMA
//        32x32x2, fixed-order tree reduction through LDS).  Wave 0 writes z = s / temp and, per row, the tile's
//        (max, first argmax, sum exp(z - max), sum z, z_ii).  The last tile of a 32-row block merges the block's
//        tile statistics into lse / argmax / label-smoothed row loss; the last row block sums the row losses.
//   bwd  nce_bwd_kernel: one launch for both gradients.  G = (softmax(z) - target) * go / (B temp) is never stored:
//        each workgroup rebuilds the entries it needs from z and lse while it loads them as the MFMA operand
//        (dq = G p: 32 x 32 tiles of [B, d], contraction over C; dp_local = G^T[:L] q: tiles of [L, d], contraction
//        over B).  The dq workgroups of the first column tile also sum g * z per row; the last of them writes d_temp.
// All reductions use fixed orders: results are bitwise reproducible run to run.
#include "common.h"
#include "../../include/abcsearch.h"
#include "nt_gemm.h"
#include <atomic>

int abc_fail_hip(hipError_t e, const char* where);
int abc_fail_invalid(const char* msg);
int abc_fail_ws(size_t need, size_t have);

#define NCE_T 32                 // tile edge

#define NCE_STAT 8               // floats per (row, column tile) statistics record (5 used; 32-byte records)
#define NCE_MAX_COUNTERS 4096    // tile counters + row-block counters + 2 (whole forward, d_temp)
#define NCE_FT 64                // forward tile edge
#define NCE_SPLITK_MAX 16

struct NceLayout {
    int ldz, n_rt, n_ct;           // backward: 32-row blocks; forward: n_frt x n_fct tiles of 64 x 64, K split splitk ways
    int n_frt, n_fct, splitk, n_cnt;
    size_t off_z, off_part, off_stats, off_rowloss, off_rowdt, off_cnt, total;
};

static NceLayout nce_layout(int B, int C, int d) {
    NceLayout L;
    L.ldz = (C + 7) & ~7;                          // 16-byte aligned rows of z whatever C is
    L.n_rt = (B + NCE_T - 1) / NCE_T;
    L.n_ct = (C + NCE_T - 1) / NCE_T;
    L.n_frt = (B + NCE_FT - 1) / NCE_FT;
    L.n_fct = (C + NCE_FT - 1) / NCE_FT;
    const int tiles = L.n_frt * L.n_fct;
    int sk = 256 / tiles;                          // fill the chip: tiles x splitk ~ 256 workgroups
    if (sk < 1) sk = 1;
    if (sk > NCE_SPLITK_MAX) sk = NCE_SPLITK_MAX;
    const int units = d / 8;                       // K is consumed in units of 8 floats per wave step
    while (sk > 1 && units / (sk * 8) < 1) sk >>= 1;
    L.splitk = sk;
    L.n_cnt = tiles + L.n_frt + 2;                 // per tile, per 64-row block, whole forward, d_temp
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    L.off_z = take((size_t)B * L.ldz * 4);
    L.off_part = take((size_t)sk * B * L.ldz * 4);
    L.off_stats = take((size_t)B * L.n_fct * NCE_STAT * 4);
    L.off_rowloss = take((size_t)B * 8);           // row losses, then the rows' accuracy flags
    L.off_rowdt = take((size_t)B * 4);
    L.off_cnt = take((size_t)L.n_cnt * 4);
    L.total = o;
    return L;
}

static bool nce_shape_ok(int B, int C, int d) {
    if (!(B > 0 && C >= B && d > 0 && (d % 8) == 0)) return false;
    return ((int64_t)((B + NCE_FT - 1) / NCE_FT)) * ((C + NCE_FT - 1) / NCE_FT) + (B + NCE_FT - 1) / NCE_FT + 2 <= NCE_MAX_COUNTERS;
}

extern "C" size_t abc_infonce_workspace_bytes(int B, int C, int d) {
    if (!nce_shape_ok(B, C, d)) return 0;
    return nce_layout(B, C, d).total;
}

// Hand-offs inside the forward launch use write-through (sc1) 16-byte stores on the producer and sc1 loads on the consumer
// instead of an agent release / acquire pair around plain accesses: a release writes back the XCD's whole L2 (about 6.5 us
// once a workgroup has dirtied a 16 KB slab, three times per forward), the sc1 form costs about what the stores cost.
// Protocol (MI355X_MICROARCH.md, "Valid forms", counter row): every byte handed off is stored sc1; every storing wave waits
// vmcnt(0); workgroup barrier; ONE lane adds to the counter (agent scope); the workgroup whose add returns total-1 is the
// consumer: that lane's wave loads after the add has returned, the other waves after a barrier; every load of the bytes is sc1.
typedef __attribute__((ext_vector_type(4))) unsigned int v4u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t nce_rsrc(const void* base, uint32_t bytes) {
    const uint64_t b = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
#define NCE_SC1 16
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, const v4f& x) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, x), r, (int)byte_off, 0, NCE_SC1);
}
__device__ __forceinline__ v4f ld_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, NCE_SC1));
}

struct NceFwdArgs {
    const float *q, *p, *temp;
    float eps;
    int B, C, d, ldz, n_frt, n_fct, splitk;
    float *z, *part, *stats, *row_loss;
    unsigned* cnt;                 // [tiles] split arrivals, [n_frt] tile arrivals per 64-row block, [1] row blocks, [1] d_temp (backward)
    float *out_loss, *out_lse, *out_acc;
    int* out_argmax;
};

// Arrive on a counter after this wave's stores; returns true in every lane of the wave that arrived LAST of `total`
// (and has then acquired the other arrivers' stores).  Called by one whole wave; only that wave's stores are published.
__device__ __forceinline__ bool nce_arrive_last(unsigned* counter, unsigned total, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's stores have left
    unsigned ticket = 0;
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // always: hipcc may drop the fence's own wait
        ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    const bool last = ticket == total - 1;
    if (last) {
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the invalidate has completed before any load below
    }
    return last;
}

// Counter arrival of a whole workgroup in the sc1 form (see above): the caller's hand-off stores were all sc1.  Every wave
// drains its stores, barrier, one lane draws the ticket and tells the others through LDS.  True in the last arriver, whose
// loads of the handed-off bytes must all be sc1 loads.
__device__ __forceinline__ bool nce_arrive_last_wg(unsigned* counter, unsigned total, unsigned* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = ticket == total - 1;
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
        *flag = last ? 1u : 0u;                               // (the ticket has returned: the add's value was used)
    }
    __syncthreads();
    return *flag != 0u;
}

// One workgroup per (64 x 64 logits tile, K split): partial tile on fp32 MFMA 32x32x2 (2 x 2 accumulators per wave, the
// split's K range cut over the 8 waves, fixed-order tree reduction through LDS) -> slab in the workspace.  The split that
// arrives last sums the slabs (fixed order), writes z = s / temp and the tile's row statistics; the tile that arrives
// last in its 64-row block merges the statistics; the last row block writes the mean loss.
__global__ __launch_bounds__(512) void nce_fwd_kernel(const NceFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = (float*)smem_raw;                                    // 64 KB: the tree reduction; reused for the flag
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, frt = tile / a.n_fct, fct = tile % a.n_fct;
    const int m0 = frt * NCE_FT, n0 = fct * NCE_FT;
    const int sp = blockIdx.y;
    const int units = a.d / 8;
    const int slice = sp * 8 + wave, nslices = a.splitk * 8;
    const int u0 = (int)(((int64_t)slice * units) / nslices), u1 = (int)(((int64_t)(slice + 1) * units) / nslices);
    const float* qa[2];
    const float* pb[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        int r = m0 + b * 32 + l31; if (r >= a.B) r = a.B - 1;
        int c = n0 + b * 32 + l31; if (c >= a.C) c = a.C - 1;
        qa[b] = a.q + (size_t)r * a.d + 4 * h;
        pb[b] = a.p + (size_t)c * a.d + 4 * h;
    }
    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // a wave's K slice is short (8 units of 8 floats at cfg4): all of its loads go out before the first MFMA, or every
    // pair of units pays a full memory round trip (the operands are read once: nothing hides it but depth)
    int u = u0;
    for (; u + 8 <= u1; u += 8) {
        v4f x[8][2], y[8][2];
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                x[w][i] = *(const v4f*)(qa[i] + 8 * (u + w));
                y[w][i] = *(const v4f*)(pb[i] + 8 * (u + w));
            }
#pragma unroll
        for (int w = 0; w < 8; ++w)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[w][i][t], y[w][j][t], acc[i][j], 0, 0, 0);
    }
    for (; u < u1; ++u) {
        v4f x[2], y[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            x[i] = *(const v4f*)(qa[i] + 8 * u);
            y[i] = *(const v4f*)(pb[i] + 8 * u);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[i][t], y[j][t], acc[i][j], 0, 0, 0);
    }
    float v[64];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[(i * 2 + j) * 16 + r] = acc[i][j][r];
    wg_tree_reduce<64>(v, lds, wave, lane);
    const size_t slab = (size_t)a.B * a.ldz;
    const __amdgpu_buffer_rsrc_t rpart = nce_rsrc(a.part, (uint32_t)((size_t)a.splitk * slab * 4));
    const __amdgpu_buffer_rsrc_t rstat = nce_rsrc(a.stats, (uint32_t)((size_t)a.B * a.n_fct * NCE_STAT * 4));
    // wave 0 holds the tile (column on the lane): through LDS as [row][col + pad] so that every thread stores two 16-byte
    // pieces of one row (thread = (row, 8-column chunk), the layout the reduction below reads)
    float* tl = lds + 64;                                             // the first words carry the arrival flag
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tl[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 68 + j * 32 + l31] = v[(i * 2 + j) * 16 + r];
    }
    __syncthreads();
    const int row = m0 + (threadIdx.x >> 3), col0 = n0 + 8 * (threadIdx.x & 7);
    const bool row_ok = row < a.B;
    const bool chunk_ok = row_ok && col0 < a.C;                       // ldz is a multiple of 8: a chunk is inside the padded row or outside
    const uint32_t eoff = (uint32_t)(((size_t)row * a.ldz + col0) * 4);
    if (chunk_ok) {
        const float* src = tl + (threadIdx.x >> 3) * 68 + 8 * (threadIdx.x & 7);
        st_sc1(rpart, (uint32_t)(sp * slab * 4) + eoff, *(const v4f*)src);
        st_sc1(rpart, (uint32_t)(sp * slab * 4) + eoff + 16, *(const v4f*)(src + 4));
    }
    unsigned* flag = (unsigned*)lds;
    if (!nce_arrive_last_wg(a.cnt + tile, (unsigned)a.splitk, flag)) return;

    // ---- last split of this tile: thread (row = tid / 8, chunk = tid % 8) owns 8 consecutive columns of one row
    const float t = a.temp[0];
    float zz[8];
    {
        float sum[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) sum[k] = 0.f;
        if (chunk_ok) {
            // every slab load goes out before the first add (a loop over a runtime split count would wait for each pair:
            // a memory round trip per split); slabs beyond the count re-read the last one and are not added.  Fixed order.
            v4f lo[NCE_SPLITK_MAX], hi[NCE_SPLITK_MAX];
#pragma unroll
            for (int s2 = 0; s2 < NCE_SPLITK_MAX; ++s2) {
                const int sc = s2 < a.splitk ? s2 : a.splitk - 1;
                lo[s2] = ld_sc1(rpart, (uint32_t)(sc * slab * 4) + eoff);
                hi[s2] = ld_sc1(rpart, (uint32_t)(sc * slab * 4) + eoff + 16);
            }
#pragma unroll
            for (int s2 = 0; s2 < NCE_SPLITK_MAX; ++s2) {
                const bool on = s2 < a.splitk;
#pragma unroll
                for (int k = 0; k < 4; ++k) { sum[k] += on ? lo[s2][k] : 0.f; sum[4 + k] += on ? hi[s2][k] : 0.f; }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) zz[k] = __fdiv_rn(sum[k], t);   // score / temp, as Temperature.forward
    }
    float mx = -INFINITY, zs = 0.f, zii = 0.f;
    int am = 0x7FFFFFFF;
    if (row_ok) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (col0 + k < a.C) {
                if (zz[k] > mx) { mx = zz[k]; am = col0 + k; }      // first maximum kept
                zs += zz[k];
                if (col0 + k == row) zii = zz[k];
            }
        if (col0 < a.C) {
            float* dst = a.z + (size_t)row * a.ldz + col0;          // padded columns get finite zeros / garbage nobody reads
            *(v4f*)dst = v4f{zz[0], zz[1], zz[2], zz[3]};
            *(v4f*)(dst + 4) = v4f{zz[4], zz[5], zz[6], zz[7]};
        }
    }
    // the row's 8 chunks sit in 8 neighbouring lanes: butterfly over lane bits 0-2 (fixed order)
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        const float om = __shfl_xor(mx, m, 64);
        const int oa = __shfl_xor(am, m, 64);
        if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
        zs += __shfl_xor(zs, m, 64);
        zii += __shfl_xor(zii, m, 64);
    }
    float se = 0.f;
    if (row_ok) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (col0 + k < a.C) se += expf(zz[k] - mx);
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) se += __shfl_xor(se, m, 64);
    if (row_ok && (threadIdx.x & 7) == 0) {
        const uint32_t so = (uint32_t)(((size_t)row * a.n_fct + fct) * NCE_STAT * 4);
        st_sc1(rstat, so, v4f{mx, __int_as_float(am), se, zs});
        st_sc1(rstat, so + 16, v4f{zii, 0.f, 0.f, 0.f});
    }
    if (!nce_arrive_last_wg(a.cnt + a.n_frt * a.n_fct + frt, (unsigned)a.n_fct, flag)) return;
    if (wave != 0) return;
    // ---- last tile of this 64-row block: one lane per row merges the block's statistics
    if (m0 + lane < a.B) {
        const int grow = m0 + lane;
        const uint32_t so = (uint32_t)((size_t)grow * a.n_fct * NCE_STAT * 4);
        float M = -INFINITY, sse = 0.f, szs = 0.f, szii = 0.f;
        int amx = 0x7FFFFFFF;
        for (int t0 = 0; t0 < a.n_fct; t0 += 8) {                // 16 loads in flight; tiles in column order
            v4f r0[8], r1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int tn = t0 + j < a.n_fct ? t0 + j : a.n_fct - 1;
                r0[j] = ld_sc1(rstat, so + tn * NCE_STAT * 4);
                r1[j] = ld_sc1(rstat, so + tn * NCE_STAT * 4 + 16);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (t0 + j >= a.n_fct) continue;
                // online merge of (max, sum exp): the running sum is rescaled when a tile raises the maximum
                if (r0[j][0] > M) { sse *= expf(M - r0[j][0]); M = r0[j][0]; amx = __float_as_int(r0[j][1]); }   // first maximum kept
                sse += r0[j][2] * expf(r0[j][0] - M);
                szs += r0[j][3];
                szii += r1[j][0];
            }
        }
        const float l = M + logf(sse);
        a.out_lse[grow] = l;
        a.out_argmax[grow] = amx;
        // (1-eps)*(-logp_ii) + (eps/C)*sum_j(-logp_ij), logp = z - lse
        __hip_atomic_store(a.row_loss + grow, (1.f - a.eps) * (l - szii) + a.eps * (l - szs / (float)a.C), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);                                  // sc1 store: handed to the last row block
        __hip_atomic_store(a.row_loss + a.B + grow, amx == grow ? 1.f : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    {   // sc1-form arrival of this one wave (its hand-off stores above are sc1; the loads below are sc1 loads)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned ticket = 0;
        unsigned* counter = a.cnt + a.n_frt * a.n_fct + a.n_frt;
        if (lane == 0) {
            ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == (unsigned)a.n_frt - 1) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (__builtin_amdgcn_readfirstlane(ticket) != (unsigned)a.n_frt - 1) return;
    }
    // ---- last row block: mean loss, fixed order
    float s = 0.f, hits = 0.f;
    for (int i = lane; i < a.B; i += 64) {
        s += __hip_atomic_load(a.row_loss + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hits += __hip_atomic_load(a.row_loss + a.B + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { s += __shfl_xor(s, m, 64); hits += __shfl_xor(hits, m, 64); }
    if (lane == 0) {
        a.out_loss[0] = s * (1.0f / (float)a.B);
        // accuracy = (argmax == arange(bs)).sum() / bs (model/abc_util.py:52-54): an exact count, one fp32 division
        if (a.out_acc) a.out_acc[0] = __fdiv_rn(hits, (float)a.B);
    }
}

// ------------------------------------------------------------------------------------------------
struct NceBwdArgs {
    const float *q, *p, *lse, *temp, *grad_out, *z;
    float eps;
    int B, C, d, ldz, local_rows;
    int n_rt, n_dt, n_lt, n_dtp;   // row tiles of dq, its 64-wide d tiles, row tiles of dp_local, its 32-wide d tiles
    int n_dq;                      // workgroups of the dq part (0 when dq is not wanted and d_temp neither)
    float *dq, *dp, *d_temp, *row_dt;
    unsigned* cnt;                 // the d_temp arrival counter
};

__device__ __forceinline__ float nce_gs(float zz, float l, bool diag, float smooth, float eps, float go, float t, float& g_out) {
    const float g = (expf(zz - l) - (smooth + (diag ? 1.f - eps : 0.f))) * go;
    g_out = g;
    return __fdiv_rn(g, t);
}

__global__ __launch_bounds__(512) void nce_bwd_kernel(const NceBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4 * 32 * 64];
    __shared__ float dts[8][NCE_T];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const float go = a.grad_out[0] / (float)a.B;
    const float t = a.temp[0];
    const float smooth = a.eps / (float)a.C;
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if ((int)blockIdx.x < a.n_dq) {
        // ---- dq[m0 .. +32, n0 .. +64] = sum_c GS[i][c] * p[c][n].  A operand (rows i) rebuilt from z; B operand: a lane
        // loads TWO neighbouring columns of p per c (8-byte loads, 256 contiguous bytes per lane half) and feeds one to
        // each of two accumulators: column n0 + 2*l31 + t belongs to accumulator t.
        typedef __attribute__((ext_vector_type(2))) float v2f;
        const int rt = blockIdx.x / a.n_dt, dtile = blockIdx.x % a.n_dt;
        const int m0 = rt * NCE_T, n0 = dtile * 64;
        const int units = a.ldz / 8;
        const int u0 = (int)(((int64_t)wave * units) / 8), u1 = (int)(((int64_t)(wave + 1) * units) / 8);
        const bool row_ok = m0 + l31 < a.B;
        const int row = row_ok ? m0 + l31 : a.B - 1;
        int n = n0 + 2 * l31; if (n + 2 > a.d) n = a.d - 2;           // d is even (d % 8 == 0)
        const float l = a.lse[row];
        const float* zr = a.z + (size_t)row * a.ldz + 4 * h;
        const float* pc = a.p + n;
        float dt = 0.f;
        v16f acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
        auto step = [&](int u, const v4f& zz, const v2f (&b)[4]) {
            const int c0 = 8 * u + 4 * h;
            float gs[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float g;
                gs[k] = nce_gs(zz[k], l, c0 + k == row, smooth, a.eps, go, t, g);
                if (!(row_ok && c0 + k < a.C)) gs[k] = 0.f;             // padded columns of z hold anything
                else dt += g * zz[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[k], b[k][0], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[k], b[k][1], acc1, 0, 0, 0);
            }
        };
        auto load = [&](int u, v4f& zz, v2f (&b)[4]) {
            const int c0 = 8 * u + 4 * h;
            zz = *(const v4f*)(zr + 8 * u);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cc = c0 + k < a.C ? c0 + k : a.C - 1;
                b[k] = *(const v2f*)(pc + (size_t)cc * a.d);
            }
        };
        int u = u0;
        for (; u + 4 <= u1; u += 4) {                                    // 20 loads in flight per wave
            v4f zz[4];
            v2f b[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) load(u + j, zz[j], b[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) step(u + j, zz[j], b[j]);
        }
        for (; u < u1; ++u) {
            v4f zz;
            v2f b[4];
            load(u, zz, b);
            step(u, zz, b);
        }
        float v[32];
#pragma unroll
        for (int i = 0; i < 16; ++i) { v[i] = acc[i]; v[16 + i] = acc1[i]; }
        if (dtile == 0 && a.d_temp) {                       // row sums of g * z: lanes l and l+32, then the 8 waves, fixed order
            dt += __shfl_xor(dt, 32, 64);
            if (h == 0) dts[wave][l31] = dt;
        }
        wg_tree_reduce<32>(v, red, wave, lane);             // (its barriers also publish dts)
        if (wave != 0) return;
        if (a.dq) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (r < a.B && n0 + 2 * l31 + 2 <= a.d) *(v2f*)(a.dq + (size_t)r * a.d + n0 + 2 * l31) = v2f{v[i], v[16 + i]};
            }
        }
        if (dtile != 0 || !a.d_temp) return;
        if (lane < NCE_T && m0 + lane < a.B) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += dts[w][lane];
            a.row_dt[m0 + lane] = s;
        }
        if (!nce_arrive_last(a.cnt, (unsigned)a.n_rt, lane)) return;
        float s = 0.f;
        for (int i = lane; i < a.B; i += 64) s += a.row_dt[i];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) a.d_temp[0] = -__fdiv_rn(s, t);     // -sum(g*s)/temp^2 with z = s/temp
        return;
    }
    // ---- dp[j0.., n0..] = sum_i GS[i][j] * q[i][n]; A operand (rows j) = column j of GS, B operand q[i][n0 + l31]
    const int id = blockIdx.x - a.n_dq;
    const int lt = id / a.n_dtp, dtile = id % a.n_dtp;
    const int j0 = lt * NCE_T, n0 = dtile * NCE_T;
    const int units = (a.B + 7) / 8;
    const int u0 = (int)(((int64_t)wave * units) / 8), u1 = (int)(((int64_t)(wave + 1) * units) / 8);
    const bool col_ok = j0 + l31 < a.local_rows;
    const int j = col_ok ? j0 + l31 : a.local_rows - 1;
    int n = n0 + l31; if (n >= a.d) n = a.d - 1;
    for (int u = u0; u < u1; ++u) {
        const int i0 = 8 * u + 4 * h;
        float gs[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = i0 + k < a.B;
            const int i = ok ? i0 + k : a.B - 1;
            float g;
            gs[k] = nce_gs(a.z[(size_t)i * a.ldz + j], a.lse[i], i == j, smooth, a.eps, go, t, g);
            if (!(ok && col_ok)) gs[k] = 0.f;
            b[k] = a.q[(size_t)i * a.d + n];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[k], b[k], acc, 0, 0, 0);
    }
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = acc[i];
    wg_tree_reduce<16>(v, red, wave, lane);
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = j0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (r < a.local_rows && n0 + l31 < a.d) a.dp[(size_t)r * a.d + n0 + l31] = v[i];
    }
}

// ------------------------------------------------------------------------------------------------
static int nce_check(int B, int C, int d) {
    if (B <= 0 || C < B || d <= 0) return abc_fail_invalid("infonce: need 0 < B <= C and d > 0");
    if (d % 8 != 0) return abc_fail_invalid("infonce: d must be a multiple of 8");
    if (!nce_shape_ok(B, C, d)) return abc_fail_invalid("infonce: B too large");
    return ABC_OK;
}

static void nce_fwd_launch(const float* q, const float* p, const float* temp, float eps, int B, int C, int d,
                           float* out_loss, float* out_lse, int32_t* out_argmax, float* out_acc, char* w, const NceLayout& L,
                           hipStream_t st) {
    NceFwdArgs a;
    a.q = q; a.p = p; a.temp = temp; a.eps = eps;
    a.B = B; a.C = C; a.d = d; a.ldz = L.ldz; a.n_frt = L.n_frt; a.n_fct = L.n_fct; a.splitk = L.splitk;
    a.z = (float*)(w + L.off_z);
    a.part = (float*)(w + L.off_part);
    a.stats = (float*)(w + L.off_stats);
    a.row_loss = (float*)(w + L.off_rowloss);
    a.cnt = (unsigned*)(w + L.off_cnt);
    a.out_loss = out_loss; a.out_lse = out_lse; a.out_argmax = out_argmax; a.out_acc = out_acc;
    hipLaunchKernelGGL(nce_fwd_kernel, dim3(L.n_frt * L.n_fct, L.splitk), dim3(512), 65536, st, a);
}

extern "C" int abc_infonce_fwd(const float* q, const float* p, const float* temp, float eps, int B, int C, int d,
                               float* out_loss, float* out_lse, int32_t* out_argmax, float* out_acc, void* ws,
                               size_t ws_bytes, void* stream) {
    int rc = nce_check(B, C, d);
    if (rc) return rc;
    if (!q || !p || !temp || !out_loss || !out_lse || !out_argmax) return abc_fail_invalid("infonce_fwd: null pointer");
    if (((uintptr_t)q | (uintptr_t)p) & 15) return abc_fail_invalid("infonce_fwd: q and p must be 16-byte aligned");
    const NceLayout L = nce_layout(B, C, d);
    if (!ws || ws_bytes < L.total) return abc_fail_ws(L.total, ws_bytes);
    if ((uintptr_t)ws & 15) return abc_fail_invalid("infonce_fwd: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    // the arrival counters of this call and of the backward that follows start at zero (a caller's workspace holds
    // anything the first time; afterwards the last arriver leaves every counter at zero itself)
    hipError_t e = hipMemsetAsync(w + L.off_cnt, 0, (size_t)L.n_cnt * 4, st);
    if (e != hipSuccess) return abc_fail_hip(e, "infonce_fwd memset");
    nce_fwd_launch(q, p, temp, eps, B, C, d, out_loss, out_lse, out_argmax, out_acc, w, L, st);
    e = hipGetLastError();
    return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "infonce_fwd launch");
}

// ------------------------------------------------------------------------------------------------
// The forward for callers that must not allocate (HIP graph capture, tight evaluation loops, harnesses that forbid
// hipMalloc): same kernel, same layout, but the scratch (logits, split-K slabs, row statistics, arrival counters)
// lives in a file-scope device array instead of a caller-provided workspace.  Each GPU the library runs on holds its
// own copy of the array at its own address until the process ends; the host-side name `scratch` is only a handle for
// hipGetSymbolAddress, never a device address.
#define NCE_SCRATCH_BYTES (16u << 20)
#define NCE_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[NCE_SCRATCH_BYTES];
static std::atomic<void*> g_scratch_ptr[NCE_MAX_DEVICES];
static std::atomic<hipStream_t> g_scratch_stream[NCE_MAX_DEVICES];      // the stream of the last call on each device
static std::atomic<bool> g_scratch_used[NCE_MAX_DEVICES];
static std::atomic<hipEvent_t> g_scratch_event[NCE_MAX_DEVICES];       // recorded behind the last eager call

// The scratch copy of the CURRENT device (each GPU has its own).  `stream`, when given, must belong to that device: the
// kernel runs where the stream lives, and another GPU's scratch address is not valid there.
static int nce_scratch(void** out, int* dev_out, hipStream_t stream, bool check_stream) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return abc_fail_hip(e, "infonce_fwd_noalloc hipGetDevice");
    if (dev < 0 || dev >= NCE_MAX_DEVICES) return abc_fail_invalid("infonce_fwd_noalloc: device index out of range");
    if (check_stream && stream) {
        int sdev = -1;
        e = hipStreamGetDevice(stream, &sdev);
        if (e != hipSuccess) return abc_fail_hip(e, "infonce_fwd_noalloc hipStreamGetDevice");
        if (sdev != dev) return abc_fail_invalid("infonce_fwd_noalloc: the stream belongs to another device than the current one");
    }
    if (dev_out) *dev_out = dev;
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {                                   // first call on this device, with this device current
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return abc_fail_hip(e, "infonce_fwd_noalloc hipGetSymbolAddress(scratch)");
        hipEvent_t ev = nullptr;                 // the "last call has finished" marker of the in-use guard, once per device
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) g_scratch_event[dev].store(ev, std::memory_order_release);
        else (void)hipGetLastError();
        g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return ABC_OK;
}

// Also the warm-up call: resolves the current device's scratch address (hipGetSymbolAddress loads the code object on
// first use, which allocates), so that a later abc_infonce_fwd_noalloc on this device may run under stream capture.
extern "C" size_t abc_infonce_noalloc_max_bytes(void) {
    void* sp = nullptr;
    (void)nce_scratch(&sp, nullptr, nullptr, false);
    return NCE_SCRATCH_BYTES;
}

extern "C" int abc_infonce_fwd_noalloc(const float* q, const float* p, const float* temp, float eps, int B, int C, int d,
                                       float* out_loss, float* out_lse, int32_t* out_argmax, float* out_acc, void* stream) {
    int rc = nce_check(B, C, d);
    if (rc) return rc;
    if (!q || !p || !temp || !out_loss || !out_lse || !out_argmax) return abc_fail_invalid("infonce_fwd_noalloc: null pointer");
    if (((uintptr_t)q | (uintptr_t)p) & 15) return abc_fail_invalid("infonce_fwd_noalloc: q and p must be 16-byte aligned");
    const NceLayout L = nce_layout(B, C, d);
    if (L.total > NCE_SCRATCH_BYTES) return abc_fail_ws(L.total, NCE_SCRATCH_BYTES);   // "workspace too small: need N bytes, have 16 MiB"
    void* sp = nullptr;
    int dev = 0;
    hipStream_t st = (hipStream_t)stream;
    rc = nce_scratch(&sp, &dev, st, true);
    if (rc) return rc;
    // Calls on one GPU share its array.  Calls on ONE stream run in order; a call on another stream while the previous
    // call has not finished would race with it on the array: refused instead of silently corrupting.  "Finished" is
    // an event recorded behind every call (created once per device at set-up; no allocation per call).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cap) != hipSuccess) { cap = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
    // (a capturing stream launches nothing now; what its graph overlaps with at replay is the replaying caller's to order)
    const bool eager = cap == hipStreamCaptureStatusNone;
    hipEvent_t ev = g_scratch_event[dev].load(std::memory_order_acquire);
    if (eager && ev && g_scratch_used[dev].load(std::memory_order_acquire) &&
        g_scratch_stream[dev].load(std::memory_order_acquire) != st) {
        const hipError_t qe = hipEventQuery(ev);
        if (qe == hipErrorNotReady) {
            (void)hipGetLastError();
            return abc_fail_invalid("infonce_fwd_noalloc: the previous call on this GPU, on another stream, has not finished "
                                    "(calls share one scratch array per GPU: use one stream, or synchronise between streams)");
        }
        if (qe != hipSuccess) (void)hipGetLastError();
    }
    char* w = (char*)sp;
    // the array keeps the last call's contents: the counters must start at zero (a completed call leaves them at zero
    // itself; this covers a call that was abandoned half-way).  Everything else is written before it is read.
    hipError_t e = hipMemsetAsync(w + L.off_cnt, 0, (size_t)L.n_cnt * 4, st);
    if (e != hipSuccess) return abc_fail_hip(e, "infonce_fwd_noalloc memset");
    nce_fwd_launch(q, p, temp, eps, B, C, d, out_loss, out_lse, out_argmax, out_acc, w, L, st);
    e = hipGetLastError();
    if (e != hipSuccess) return abc_fail_hip(e, "infonce_fwd_noalloc launch");
    if (eager && ev) {
        if (hipEventRecord(ev, st) == hipSuccess) {
            g_scratch_stream[dev].store(st, std::memory_order_release);
            g_scratch_used[dev].store(true, std::memory_order_release);
        } else (void)hipGetLastError();
    }
    return ABC_OK;
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
    NceBwdArgs a;
    a.q = q; a.p = p; a.lse = lse; a.temp = temp; a.grad_out = grad_out;
    a.z = (const float*)(w + L.off_z);                    // logits kept by abc_infonce_fwd in the same workspace
    a.eps = eps; a.B = B; a.C = C; a.d = d; a.ldz = L.ldz;
    a.local_rows = dp_local ? local_rows : 0;
    a.n_rt = L.n_rt;
    a.n_dt = (d + 63) / 64;
    a.n_dtp = (d + NCE_T - 1) / NCE_T;
    a.n_lt = (a.local_rows + NCE_T - 1) / NCE_T;
    // d_temp alone still needs the row sums: the first column tile of every row block then runs without storing dq
    a.n_dq = dq ? a.n_rt * a.n_dt : (d_temp ? a.n_rt * a.n_dt : 0);
    a.dq = dq; a.dp = dp_local; a.d_temp = d_temp;
    a.row_dt = (float*)(w + L.off_rowdt);
    a.cnt = (unsigned*)(w + L.off_cnt) + L.n_cnt - 1;
    const int grid = a.n_dq + a.n_lt * a.n_dtp;
    if (grid == 0) return ABC_OK;
    hipLaunchKernelGGL(nce_bwd_kernel, dim3(grid), dim3(512), 0, st, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ABC_OK : abc_fail_hip(e, "infonce_bwd launch");
}
