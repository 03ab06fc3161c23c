// Host side of the fused score-GEMM + top-k sweep: launch plan, workspace layout, the list merges and the
// positive-score kernel.  The sweep kernel itself lives in sweep_kernels.h / sweep16.inc.
#include "sweep_kernels.h"

// ------------------------------------------------------------------------------------------------
// Select the top k of each query from its n_lists (= n_splits) candidate lists as the sweep left them: list l holds
// list_cnt[l][q] keys in no order.  One wave per query: entries are filtered with the query's final bound (the minimum of
// the bounds its lists published: at least k entries reach it, so nothing below it can be in the top k), the survivors
// are packed through a per-wave LDS stage and folded 64 at a time (sort + bitonic merge) into the running top k.
template <int EK>
__global__ __launch_bounds__(256) void topk_merge_packed_kernel(const uint64_t* __restrict__ cand, const uint32_t* __restrict__ list_cnt,
                                                                const float* __restrict__ pub, int n_lists, int cap,
                                                                int64_t nq, int64_t nq_pad, int k, int64_t idx_offset,
                                                                float* __restrict__ out_val, int64_t* __restrict__ out_idx,
                                                                uint64_t* __restrict__ out_keys,
                                                                const uint32_t* __restrict__ rank_part,
                                                                int64_t* __restrict__ out_rank) {
    __shared__ uint64_t stage_all[4][128];
    const int lane = threadIdx.x & 63;
    const int64_t qr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qr >= nq) return;
    uint64_t* const stage = stage_all[threadIdx.x >> 6];
    float mn = INFINITY;
    for (int l = lane; l < n_lists; l += 64) mn = fminf(mn, pub[qr * n_lists + l]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
    // key of the smallest score that can still be in the top k (keys order like scores; any index): pass = key >= kmin
    // (a bound of +-0 admits both zeros: fmax/fmin do not keep the sign the keys order by)
    const uint64_t kmin = (mn == -INFINITY) ? 0ull : ((uint64_t)f32_orderable(mn == 0.f ? -0.f : mn) << 32);
    uint64_t best[EK];
#pragma unroll
    for (int e = 0; e < EK; ++e) best[e] = 0ull;
    auto fold = [&](uint64_t chunk) {                    // 64 keys, any order
        uint64_t c1[1] = {chunk};
        wave_sort_desc<1>(c1, lane);
        // best (desc) against the reversed chunk (zeros first for EK = 2): the element-wise max is bitonic and holds the top
        const uint64_t rev = shfl_u64(c1[0], 63 - lane);
        best[EK - 1] = best[EK - 1] > rev ? best[EK - 1] : rev;
        wave_bitonic_merge_desc<EK>(best, lane);
    };
    int ns = 0;
    for (int l = 0; l < n_lists; ++l) {
        const uint64_t* src = cand + ((int64_t)l * nq_pad + qr) * cap;
        int n = (int)list_cnt[(int64_t)l * nq_pad + qr];
        n = n < cap ? n : cap;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const uint64_t key = (i < n) ? src[i] : 0ull;
            const bool keep = key != 0ull && key >= kmin;
            const uint64_t bal = __ballot(keep);
            if (keep) stage[ns + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = key;
            ns += __builtin_popcountll(bal);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (ns >= 64) {
                const uint64_t chunk = stage[lane];
                const uint64_t up = (64 + lane < ns) ? stage[64 + lane] : 0ull;
                __builtin_amdgcn_wave_barrier();
                stage[lane] = up;
                ns -= 64;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                fold(chunk);
            }
        }
    }
    if (ns > 0) fold(lane < ns ? stage[lane] : 0ull);
#pragma unroll
    for (int e = 0; e < EK; ++e) {
        const int i = e * 64 + lane;
        if (i < k) {
            const uint64_t kk = best[e];
            if (out_keys) {
                // packed form for the cross-GPU exchange: the index field becomes 0xFFFFFFFF - (index + idx_offset), i.e. the
                // key minus the offset (no borrow: the host checked idx_offset + nc <= 0xFFFFFFFF); 0 stays "no entry"
                out_keys[qr * k + i] = kk ? kk - (uint64_t)idx_offset : 0ull;
            } else {
                out_val[qr * k + i] = kk ? key_val(kk) : -INFINITY;
                out_idx[qr * k + i] = kk ? (int64_t)key_idx(kk) + idx_offset : -1;
            }
        }
    }
    if (out_rank != nullptr && lane == 0) {
        int64_t tot = 0;
        for (int l = 0; l < n_lists; ++l) tot += rank_part[(int64_t)l * nq_pad + qr];
        out_rank[qr] = tot;
    }
}

// Merge `parts` external (val, idx) lists per query -- the cross-shard / cross-GPU merge.
template <int EK>
__global__ __launch_bounds__(256) void topk_merge_lists_kernel(const float* __restrict__ vals, const int64_t* __restrict__ idx,
                                                               int parts, int64_t nq, int k,
                                                               float* __restrict__ out_val, int64_t* __restrict__ out_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t qr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qr >= nq) return;
    uint64_t best[EK];
#pragma unroll
    for (int e = 0; e < EK; ++e) best[e] = 0ull;
    for (int l = 0; l < parts; ++l) {
        const int64_t base = ((int64_t)l * nq + qr) * k;
        uint64_t nw[EK];
#pragma unroll
        for (int e = 0; e < EK; ++e) {
            const int i = e * 64 + lane;
            uint64_t kk = 0ull;
            if (i < k) {
                const int64_t ix = idx[base + i];
                if (ix >= 0) kk = make_key(vals[base + i], (uint32_t)ix);
            }
            nw[e] = kk;
        }
        // inputs are sorted by (value desc, index asc) == key desc, holes (key 0) last
#pragma unroll
        for (int e = 0; e < EK; ++e) {
            const uint64_t rev = shfl_u64(nw[EK - 1 - e], 63 - lane);
            best[e] = best[e] > rev ? best[e] : rev;
        }
        wave_bitonic_merge_desc<EK>(best, lane);
    }
#pragma unroll
    for (int e = 0; e < EK; ++e) {
        const int i = e * 64 + lane;
        if (i < k) {
            const uint64_t kk = best[e];
            out_val[qr * k + i] = kk ? key_val(kk) : -INFINITY;
            out_idx[qr * k + i] = kk ? (int64_t)key_idx(kk) : -1;
        }
    }
}

// Merge `parts` packed key lists per query (each sorted descending, 0 = no entry) -- the merge after ONE all-gather of
// 8 bytes per entry.
template <int EK>
__global__ __launch_bounds__(256) void topk_merge_keys_kernel(const uint64_t* __restrict__ keys, int parts, int64_t nq, int k,
                                                              float* __restrict__ out_val, int64_t* __restrict__ out_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t qr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qr >= nq) return;
    uint64_t best[EK];
#pragma unroll
    for (int e = 0; e < EK; ++e) best[e] = 0ull;
    for (int l = 0; l < parts; ++l) {
        const uint64_t* src = keys + ((int64_t)l * nq + qr) * k;
        uint64_t nw[EK];
#pragma unroll
        for (int e = 0; e < EK; ++e) {
            const int i = e * 64 + lane;
            nw[e] = (i < k) ? src[i] : 0ull;
        }
#pragma unroll
        for (int e = 0; e < EK; ++e) {
            const uint64_t rev = shfl_u64(nw[EK - 1 - e], 63 - lane);
            best[e] = best[e] > rev ? best[e] : rev;
        }
        wave_bitonic_merge_desc<EK>(best, lane);
    }
#pragma unroll
    for (int e = 0; e < EK; ++e) {
        const int i = e * 64 + lane;
        if (i < k) {
            const uint64_t kk = best[e];
            out_val[qr * k + i] = kk ? key_val(kk) : -INFINITY;
            out_idx[qr * k + i] = kk ? (int64_t)key_idx(kk) : -1;
        }
    }
}

int merge_keys_run(const uint64_t* keys, int parts, int64_t nq, int k, float* out_val, int64_t* out_idx, hipStream_t st,
                   hipError_t* herr) {
    const int mgrid = (int)((nq + 3) / 4);
    if (mgrid == 0) return ABC_OK;
    if (k <= 64) hipLaunchKernelGGL(topk_merge_keys_kernel<1>, dim3(mgrid), dim3(256), 0, st, keys, parts, nq, k, out_val, out_idx);
    else hipLaunchKernelGGL(topk_merge_keys_kernel<2>, dim3(mgrid), dim3(256), 0, st, keys, parts, nq, k, out_val, out_idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *herr = e; return ABC_E_HIP; }
    return ABC_OK;
}

// ------------------------------------------------------------------------------------------------
// Positive scores with the arithmetic of the sweep: out[i] = (q[i] . c[i]) as the element (i, i) of a 16 x 16 score
// tile contracted by the SAME mma16, fragments and k order as the sweep kernel (the 16-byte chunk 4s + g4 of a row goes
// to lane group g4 of k-step s, k-steps of 64 bytes in row order), so that the value is bit-identical to the score the
// filter sees for column i of query i.  The hard-negative rule compares every score with 0.95 * scores[idx] taken out
// of the same GEMV (pretrain/create_negative_index.py:47-49): a positive score from a different accumulation order
// can flip candidates within an ulp of the bound.
// One wave per 16 row pairs; fragments straight from global memory (the rows are read once: no LDS stage).
template <int DT>
__global__ __launch_bounds__(256) void rowdot_mma_kernel(const char* __restrict__ q, const char* __restrict__ c,
                                                         const float* __restrict__ qs, const float* __restrict__ cs,
                                                         int64_t n, int rowbytes, float* __restrict__ out) {
    typedef typename Acc16<DT>::type acc_t;
    const int lane = threadIdx.x & 63, l15 = lane & 15, g4 = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (row0 >= n) return;
    const int64_t row = row0 + l15 < n ? row0 + l15 : n - 1;       // clamped rows compute garbage nobody stores
    const char* ap = c + row * rowbytes + 16 * g4;                  // corpus rows on the MFMA row axis, as in the sweep
    const char* bp = q + row * rowbytes + 16 * g4;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = 0;
    for (int off = 0; off < rowbytes; off += 128) {                 // rowbytes % 128 == 0 (checked by the caller)
        v4i a[2], b[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            a[s] = *(const v4i*)(ap + off + 64 * s);
            b[s] = *(const v4i*)(bp + off + 64 * s);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            mma16<DT, 0>(acc, a[s], b[s]);
            if constexpr (DT == ABC_DT_F32) {
                mma16<DT, 1>(acc, a[s], b[s]);
                mma16<DT, 2>(acc, a[s], b[s]);
                mma16<DT, 3>(acc, a[s], b[s]);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // the last MFMA's result (hipcc pads nothing for asm)
    // element (i, i): column l15 = i lives in lane group g4 = i >> 2, register i & 3
    float v = 0.f;
    int vi = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if constexpr (DT == ABC_DT_I8) {
            int x;
            const int src = acc[r];
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(src));
            if (r == (l15 & 3)) vi = x;
        } else {
            float x;
            const float src = acc[r];
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(src));
            if (r == (l15 & 3)) v = x;
        }
    }
    if (g4 == (l15 >> 2) && row0 + l15 < n) {
        if constexpr (DT == ABC_DT_I8) out[row] = (float)vi * (qs[row] * cs[row]);
        else out[row] = v;
    }
}

int rowdot_run(int dt, const void* q, const float* qs, const void* c, const float* cs, int64_t n, int d, float* out,
               hipStream_t st, hipError_t* herr) {
    const int rowbytes = d * (dt == ABC_DT_BF16 ? 2 : (dt == ABC_DT_I8 ? 1 : 4));
    const unsigned grid = (unsigned)((n + 63) / 64);
    if (dt == ABC_DT_BF16) hipLaunchKernelGGL(rowdot_mma_kernel<ABC_DT_BF16>, dim3(grid), dim3(256), 0, st, (const char*)q, (const char*)c, qs, cs, n, rowbytes, out);
    else if (dt == ABC_DT_F32) hipLaunchKernelGGL(rowdot_mma_kernel<ABC_DT_F32>, dim3(grid), dim3(256), 0, st, (const char*)q, (const char*)c, qs, cs, n, rowbytes, out);
    else hipLaunchKernelGGL(rowdot_mma_kernel<ABC_DT_I8>, dim3(grid), dim3(256), 0, st, (const char*)q, (const char*)c, qs, cs, n, rowbytes, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *herr = e; return ABC_E_HIP; }
    return ABC_OK;
}

// ------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------
static int elem_bytes(int dt) { return dt == ABC_DT_BF16 ? 2 : (dt == ABC_DT_I8 ? 1 : 4); }

// Tuning / diagnostics knobs.  The environment (ABC_DEBUG_FLAGS, ABC_PLAN_QB) is read ONCE, when the
// library is loaded; afterwards only abc_debug_set() changes them (tests and timing scripts).  Flags that make
// results wrong (1: no epilogue, 2: nothing appended) exist only in -DABC_DIAG builds: a production library masks
// them out, so a stray environment variable cannot corrupt a search.
#ifdef ABC_DIAG
#define ABC_FLAG_MASK (~0)
#else
#define ABC_FLAG_MASK (~3)
#endif
struct SweepTuning {
    std::atomic<int> flags{0};         // SweepParams::debug_flags
    std::atomic<int> plan_qb{0};       // > 0: query blocks per launch (timing experiments)
    SweepTuning() {
        const char* e = getenv("ABC_DEBUG_FLAGS");
        if (e && *e) flags = atoi(e) & ABC_FLAG_MASK;
        e = getenv("ABC_PLAN_QB");
        if (e && atoi(e) >= 8) plan_qb = atoi(e);
    }
};
static SweepTuning g_tuning;            // constructed at load time

int sweep_debug_set(const char* name, long value) {
    if (!name) return ABC_E_INVALID;
    if (!strcmp(name, "flags")) g_tuning.flags = (int)value & ABC_FLAG_MASK;
    else if (!strcmp(name, "plan_qb")) g_tuning.plan_qb = value >= 8 ? (int)value : 0;
    else return ABC_E_INVALID;
    return ABC_OK;
}
long sweep_debug_get(const char* name) {
    if (!name) return -2;
    if (!strcmp(name, "flags")) return g_tuning.flags;
    if (!strcmp(name, "plan_qb")) return g_tuning.plan_qb;
    return -2;
}

void sweep_plan(int64_t nq, int64_t nc, int k, SweepPlan* pl) {
    pl->n_tiles = (int)((nc + TILE - 1) / TILE);
    const int64_t qblocks_total = (nq + TILE - 1) / TILE;
    // at most 32 query blocks per launch (8 splits = 8 lists per query), so that every list publishes its 16th best or
    // better (pub_m = ceil(k / n_lists) <= 16 for k <= 128: the tracked bounds come from the 4 x 4 best scores the four
    // lanes of a query remember).  Blocks are spread evenly over the launches (98 blocks -> 25+25+24+24, not
    // 32+32+32+2: a 2-block launch fills 40 of the 256 CUs)
    {
        const int64_t max_qb = 32;
        const int64_t launches = (qblocks_total + max_qb - 1) / max_qb;
        int64_t per = launches > 0 ? (qblocks_total + launches - 1) / launches : 1;
        // multiples of 8 keep the XCD-aware block mapping, which only a long sweep profits from
        if (per > 8 && pl->n_tiles >= 512) per = (per + 7) & ~(int64_t)7;
        pl->qblocks_per_launch = (int)(per < 1 ? 1 : per);
    }
    {   // timing experiments only: plan_qb = query blocks per launch (splits follow as 256 / that)
        const int e = g_tuning.plan_qb;
        if (e >= 8) pl->qblocks_per_launch = (int)(qblocks_total < e ? qblocks_total : e);
    }
    int s = 256 / pl->qblocks_per_launch;
    if (s < 1) s = 1;
    if (s > 256) s = 256;              // one query block (a per-sample call, create_negative_index.py:45-52): every CU takes a split
    if (s > pl->n_tiles) s = pl->n_tiles;
    if (s < 1) s = 1;
    pl->n_splits = s;
    pl->nq_pad = (int64_t)pl->qblocks_per_launch * TILE;
}

size_t sweep_workspace_bytes(int64_t nq, int64_t nc, int k) {
    SweepPlan pl;
    sweep_plan(nq, nc, k, &pl);
    const size_t lists = (size_t)pl.n_splits * pl.nq_pad;
    return lists * CAND_CAP * sizeof(uint64_t) + 2 * lists * sizeof(uint32_t) + lists * sizeof(float) + 256;
}


// The sweep kernel is instantiated in sweep_inst_{bf16,i8,f32}.hip (one translation unit per element type, so that
// they build in parallel); each exports one launcher.
static hipError_t launch_sweep(int dt, bool count, const SweepParams& p, int grid, hipStream_t st) {
    if (dt == ABC_DT_BF16) return launch_sweep_bf16(count, p, grid, st);
    if (dt == ABC_DT_I8) return launch_sweep_i8(count, p, grid, st);
    return launch_sweep_f32(count, p, grid, st);
}

int sweep_run(int dt, const void* q, const float* q_scale, int64_t nq, const void* c, const float* c_scale, int64_t nc,
              int d, int k, const float* upper_bound, int64_t idx_offset, float* out_val, int64_t* out_idx, uint64_t* out_keys,
              const float* rank_ref, int64_t rank_self_offset, int64_t* out_rank, void* workspace, hipStream_t st,
              hipError_t* herr) {
    SweepPlan pl;
    sweep_plan(nq, nc, k, &pl);
    const int rowbytes = d * elem_bytes(dt);
    const size_t lists = (size_t)pl.n_splits * pl.nq_pad;
    uint64_t* cand = (uint64_t*)workspace;
    uint32_t* rank_part = (uint32_t*)(cand + lists * CAND_CAP);
    float* pub = (float*)(rank_part + lists);
    uint32_t* list_cnt = (uint32_t*)(pub + lists);
    unsigned long long* dbg = (unsigned long long*)(((uintptr_t)(list_cnt + lists) + 15) & ~(uintptr_t)15);
    const bool count = rank_ref != nullptr && out_rank != nullptr;

    for (int64_t q0 = 0; q0 < nq; q0 += pl.nq_pad) {
        const int64_t nq_l = (nq - q0 < pl.nq_pad) ? nq - q0 : pl.nq_pad;
        SweepParams p;
        p.q = (const char*)q + q0 * rowbytes;
        p.c = (const char*)c;
        p.nq = nq_l;
        p.nc = nc;
        p.rowbytes = rowbytes;
        p.k = k;
        p.n_qblocks = (int)((nq_l + TILE - 1) / TILE);
        p.n_splits = pl.n_splits;
        p.n_tiles = pl.n_tiles;
        p.nq_pad = pl.nq_pad;
        p.ub = upper_bound ? upper_bound + q0 : nullptr;
        p.q_scale = q_scale ? q_scale + q0 : nullptr;
        p.c_scale = c_scale;
        p.rank_ref = count ? rank_ref + q0 : nullptr;
        p.rank_excl = (rank_self_offset == INT64_MIN) ? INT64_MIN / 2 : rank_self_offset + q0;
        p.debug_flags = g_tuning.flags;
        p.n_lists = pl.n_splits;
        p.pub_m = (k + p.n_lists - 1) / p.n_lists;
        p.pub = pub;
        {   // published bounds start at -inf (0xFF800000) for every launch
            hipError_t em = hipMemsetD32Async((hipDeviceptr_t)pub, 0xFF800000u, lists, st);
            if (em != hipSuccess) { *herr = em; return ABC_E_HIP; }
        }
        p.dbg = dbg;
        if (p.debug_flags & 240) (void)hipMemsetAsync(dbg, 0, 176, st);
        p.cand = cand;
        p.rank_part = rank_part;
        p.list_cnt = list_cnt;
        const int grid = p.n_qblocks * p.n_splits;
        hipError_t e = launch_sweep(dt, count, p, grid, st);
        if (e != hipSuccess) { *herr = e; return ABC_E_HIP; }
        const int mgrid = (int)((nq_l + 3) / 4);
        if (k <= 64)
            hipLaunchKernelGGL(topk_merge_packed_kernel<1>, dim3(mgrid), dim3(256), 0, st, cand, list_cnt, pub, p.n_lists, CAND_CAP, nq_l,
                               pl.nq_pad, k, idx_offset, out_val ? out_val + q0 * k : nullptr, out_idx ? out_idx + q0 * k : nullptr,
                               out_keys ? out_keys + q0 * k : nullptr, rank_part,
                               count ? out_rank + q0 : nullptr);
        else
            hipLaunchKernelGGL(topk_merge_packed_kernel<2>, dim3(mgrid), dim3(256), 0, st, cand, list_cnt, pub, p.n_lists, CAND_CAP, nq_l,
                               pl.nq_pad, k, idx_offset, out_val ? out_val + q0 * k : nullptr, out_idx ? out_idx + q0 * k : nullptr,
                               out_keys ? out_keys + q0 * k : nullptr, rank_part,
                               count ? out_rank + q0 : nullptr);
        e = hipGetLastError();
        if (e != hipSuccess) { *herr = e; return ABC_E_HIP; }
    }
    {
        if (g_tuning.flags & 240) {                     // debug only: synchronous read-back of the last launch's counters
            unsigned long long hc[22] = {0};
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(hc, dbg, sizeof(hc), hipMemcpyDeviceToHost);
            fprintf(stderr, "[abc debug] appends=%llu block descents=%llu compactions=%llu compacted_entries=%llu (last launch)\n", hc[0], hc[19], hc[1], hc[2]);
            fprintf(stderr, "[abc debug] wave-clocks: filter=%llu flush=%llu compact=%llu refresh=%llu final=%llu total=%llu clock=%.3f GHz\n",
                    hc[3], hc[4], hc[5], hc[6], hc[7], hc[8], 0.1 * (double)hc[8] / (double)(hc[9] ? hc[9] : 1));
            fprintf(stderr, "[abc debug] of which tiles 1-4: filter=%llu flush=%llu compact=%llu refresh=%llu\n",
                    hc[14], hc[15], hc[16], hc[17]);
            if (g_tuning.flags & 128) fprintf(stderr, "[abc debug] quarter-steps: Q0=%llu Q1=%llu Q2=%llu Q3=%llu wave-clocks\n", hc[12], hc[13], hc[20], hc[21]);
            else if (hc[12] || hc[13]) fprintf(stderr, "[abc debug] descents: group tests=%llu append loops=%llu wave-clocks (stamps included)\n", hc[12], hc[13]);
            if (hc[10] || hc[11]) fprintf(stderr, "[abc debug] phase loop: vmcnt(0) wait=%llu barrier=%llu wave-clocks\n", hc[10], hc[11]);
        }
    }
    return ABC_OK;
}

int merge_lists_run(const float* vals, const int64_t* idx, int parts, int64_t nq, int k, float* out_val,
                    int64_t* out_idx, hipStream_t st, hipError_t* herr) {
    const int mgrid = (int)((nq + 3) / 4);
    if (mgrid == 0) return ABC_OK;
    if (k <= 64) hipLaunchKernelGGL(topk_merge_lists_kernel<1>, dim3(mgrid), dim3(256), 0, st, vals, idx, parts, nq, k, out_val, out_idx);
    else hipLaunchKernelGGL(topk_merge_lists_kernel<2>, dim3(mgrid), dim3(256), 0, st, vals, idx, parts, nq, k, out_val, out_idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *herr = e; return ABC_E_HIP; }
    return ABC_OK;
}
