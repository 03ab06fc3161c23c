#pragma once
// Fused score-GEMM + running top-k sweep for gfx950 (MI355X).
//
// One workgroup (8 waves, 512 threads) owns a block of 256 queries and a contiguous range of
// 256-row corpus tiles (a "split").  For each tile it contracts the 256x256 score tile on the
// matrix cores (corpus rows on the MFMA row axis, queries on the lane/column axis) and then filters
// the accumulators, still in registers, against each query's running threshold.  Survivors are
// appended to that query's candidate list in the workspace; a full list is compacted (wave-wide
// bitonic sort, keep the best k, raise the threshold).  The score matrix never exists in memory.
//
// Staging: the contraction runs in PHASES of 64 bytes per row (32 bf16).  Both operand tiles of a
// phase (256 rows x 64 B each) go global -> LDS with buffer_load ... lds (no VGPR round trip) into a
// ring of 4 slots; the loads of phase p+3 are issued while phase p is read, waits are counted
// (s_waitcnt vmcnt(N), never 0 in steady state).  Each phase is a read segment and an MFMA segment
// separated by raw s_barriers, and the two wave groups that share the SIMDs run one segment apart,
// so one of them always feeds the matrix pipe while the other reads and issues loads.  The
// 16-byte chunk index is XOR-swizzled with (row>>2)&3 on the SOURCE address and on the fragment read
// so that ds_read_b128 is bank-conflict free (LDS-DMA destinations are lane-linear).
#include "common.h"
#include "topk_sweep.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <atomic>

#define TILE 256
#define ROWB 64                            // bytes of one row staged per phase (32 bf16 / 64 i8 / 16 f32)
#define STAGE_BYTES (TILE * ROWB)          // one operand, one phase: 16 KiB
#define SLOT_BYTES (2 * STAGE_BYTES)       // both operands: 32 KiB
#define N_SLOTS 4                          // ring of phases in LDS: 128 KiB
#define RING_BYTES (N_SLOTS * SLOT_BYTES)
#define Q_CAP 128                          // per-wave append queue in LDS: Q_CAP keys (8 B) + query slots (1 B)
#define Q_BYTES (Q_CAP * 9)
#define CS_BYTES 512                       // int8: the 128 corpus scales of a wave's tile half
#define LDS_BYTES (RING_BYTES + 8 * Q_BYTES + 8 * CS_BYTES)
#define LIST_TRIG (CAND_CAP - 128 - Q_CAP) // compact a list that holds more than this at the end of a tile

// ------------------------------------------------------------------------------------------------
template <int DT> struct Acc;
template <> struct Acc<ABC_DT_BF16> { typedef v16f type; };
template <> struct Acc<ABC_DT_F32> { typedef v16f type; };
template <> struct Acc<ABC_DT_I8> { typedef v16i type; };

template <int DT>
__device__ __forceinline__ void mma_step(typename Acc<DT>::type& acc, const v4i& a, const v4i& b) {
    if constexpr (DT == ABC_DT_BF16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8s, a), __builtin_bit_cast(v8s, b), acc, 0, 0, 0);
    } else if constexpr (DT == ABC_DT_I8) {
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
    } else {
        // f32: each lane holds 4 consecutive k of its row; one 32x32x2 MFMA per element.  The k order
        // (chunk 2s+h, element t) is the same permutation for A and B.
        const v4f af = __builtin_bit_cast(v4f, a), bf = __builtin_bit_cast(v4f, b);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], bf[t], acc, 0, 0, 0);
    }
}

// Issue the LDS-DMA loads of one phase (both operands, 2 + 2 wave-instructions of 16 rows x 64 B).
__device__ __forceinline__ void stage_issue(__amdgpu_buffer_rsrc_t rc, __amdgpu_buffer_rsrc_t rq, char* smem, int slot,
                                            int wave, const int (&voff)[2], int soff) {
    char* dst = smem + slot * SLOT_BYTES + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, LDS_PTR(dst + i * 1024), 16, voff[i], soff, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, LDS_PTR(dst + STAGE_BYTES + i * 1024), 16, voff[i], soff, 0, 0);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base, uint32_t bytes) {
    // wave-uniform by construction (kernel arguments and blockIdx only); readfirstlane makes it provable
    const uint64_t b = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const uint32_t n = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

// ------------------------------------------------------------------------------------------------
// Compaction of one query's candidate list: sort (desc), keep the best k, return the new threshold and
// the value at rank pub_m-1 (the bound this list publishes).  Called by the whole wave (uniform);
// `buf` is wave-uniform.  E = keys per lane, chosen by the caller from the list length (n <= 64*E).
template <int E>
__device__ __forceinline__ void compact_list_e(uint64_t* buf, int n, int k, int pub_m, int lane, float& new_thr,
                                               int& new_cnt, float& pub_val, bool finalize) {
    uint64_t key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        key[e] = (i < n) ? buf[i] : 0ull;
    }
    wave_sort_desc<E>(key, lane);
    const int keep = n < k ? n : k;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        if (i < keep) buf[i] = key[e];
        else if (finalize && i < k) buf[i] = 0ull;
    }
    if (finalize && E == 1 && k > 64 && lane + 64 < k) buf[lane + 64] = 0ull;    // pad the part of [keep,k) beyond this E
    // value at rank k-1 becomes the threshold once the list is full
    uint64_t kth = 0;
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e == ((k - 1) >> 6)) kth = key[e];
    kth = shfl_u64(kth, (k - 1) & 63);
    new_cnt = keep;
    new_thr = (n >= k) ? key_val(kth) : -INFINITY;
    // every list of the query publishes its value at rank pub_m-1 (pub_m <= 64); with L lists and
    // L*pub_m >= k the minimum over the lists is a lower bound of the query's global k-th best
    const uint64_t mth = shfl_u64(key[0], (pub_m - 1) & 63);
    pub_val = (n >= pub_m) ? key_val(mth) : -INFINITY;
}

__device__ __forceinline__ void compact_list(uint64_t* buf, int n, int k, int pub_m, int lane, float& new_thr,
                                             int& new_cnt, float& pub_val, bool finalize) {
    if (n <= 64) compact_list_e<1>(buf, n, k, pub_m, lane, new_thr, new_cnt, pub_val, finalize);
    else if (n <= 128) compact_list_e<2>(buf, n, k, pub_m, lane, new_thr, new_cnt, pub_val, finalize);
    else if (n <= 256) compact_list_e<4>(buf, n, k, pub_m, lane, new_thr, new_cnt, pub_val, finalize);
    else compact_list_e<CAND_CAP / 64>(buf, n, k, pub_m, lane, new_thr, new_cnt, pub_val, finalize);
}

// ------------------------------------------------------------------------------------------------
template <int DT, bool COUNT>
__global__ __launch_bounds__(512, 2) void topk_sweep_kernel(const SweepParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Acc<DT>::type acc_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;        // corpus half of the tile (rows wm*128 ..)
    const int wn = wave & 3;         // query quarter of the tile (cols wn*64 ..)
    const int h = lane >> 5;
    const int l31 = lane & 31;

    // ---- workgroup -> (query block, split); XCD-aware: blocks with equal id%8 share an L2 ----------
    int qb, split;
    {
        const int id = blockIdx.x;
        if ((p.n_qblocks & 7) == 0) {
            // g XCDs share a group of per*g query blocks and take n_splits/g splits each (g = 1: an XCD sweeps
            // the whole corpus for 'per' query blocks)
            const int gsel = (p.debug_flags >> 12) & 3;
            // g = 2: 8 query blocks x 4 splits per XCD, the pair with the fewest distinct operand bytes per phase
            // beyond L2 (+3 % over g = 4 on the mining shape, measured; int8 with 16 blocks per launch: equal)
            int g = gsel == 0 ? 2 : gsel == 1 ? 1 : gsel == 2 ? 4 : 8;
            if (p.n_splits % g) g = 1;
            const int per = (p.n_qblocks >> 3) * g;
            const int x = id & 7, j = id >> 3;
            qb = (x / g) * per + (j % per);
            split = (x % g) * (p.n_splits / g) + j / per;
        } else {
            qb = id % p.n_qblocks;
            split = id / p.n_qblocks;
        }
    }
    const int64_t qrow0 = (int64_t)qb * TILE;
    const int t_begin = (int)(((int64_t)split * p.n_tiles) / p.n_splits);
    const int t_end = (int)(((int64_t)(split + 1) * p.n_tiles) / p.n_splits);
    const int nk = p.nk;
    const int rowbytes = p.rowbytes;

    // ---- staging addresses -----------------------------------------------------------------------
    int voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + i * 16 + (lane >> 2);
        const int swz = (lane >> 4) & 3;                  // == (row >> 2) & 3
        voff[i] = row * rowbytes + (((lane & 3) ^ swz) << 4);
    }
    const int voff_p[2] = {wave * 2048 + lane * 16, wave * 2048 + 1024 + lane * 16};
    const int64_t q_rows_left = p.nq - qrow0;
    const uint32_t q_bytes = (uint32_t)((q_rows_left < TILE ? q_rows_left : TILE) * rowbytes);
#ifdef ABC_PHASE_STAMPS                                            // diagnostic build: where the staging traffic is served from
    const bool same_q = (p.debug_flags & 32768) != 0;              // every workgroup stages query block 0
    const bool same_c = (p.debug_flags & 65536) != 0;              // every workgroup stages corpus tile 0
#else
    constexpr bool same_q = false, same_c = false;
#endif
#ifdef ABC_PHASE_STAMPS
    const bool packed_t = (p.debug_flags & 131072) != 0;           // TIMING ONLY: phase-major operand addressing (wrong data)
#else
    constexpr bool packed_t = false;
#endif
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(p.q + (same_q ? 0 : qrow0) * rowbytes, q_bytes);

    // fragment read offsets: row (l31) of the wave's sub-tile, chunk (2s+h) ^ ((l31>>2)&3)
    const int swz_r = (l31 >> 2) & 3;
    int xs[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) xs[s] = l31 * ROWB + (((2 * s + h) ^ swz_r) << 4);
    const int a_base = wm * 128 * ROWB;                   // corpus operand, rows wm*128 + mb*32 + l31
    const int b_base = STAGE_BYTES + wn * 64 * ROWB;      // query operand, rows wn*64 + nb*32 + l31

    // ---- per-query running state (lanes l and l+32 hold the same query and the same values) --------
    const int ss = split * 2 + wm;                        // candidate-list set of this wave
    float thr[2], thr_sh[2], theff[2], ub[2], qsc[2], rref[2];   // local k-th best, shared bound (pred), max of both
    int cnt[2], base[2];                                 // entries in the list / entries right after its last compaction
    uint32_t rank[2], excl[2];
    int64_t qrow[2];
    // the 4 best scores this lane has passed on per query (descending): lanes l and l+32 together know 8 real
    // entries of the list, whose pub_m-th best (pub_m <= 8) is published as the list's bound WITHOUT a compaction
    float top4[2][4];
    const bool track = p.pub_m <= 8 && p.n_lists > 1;
    // short sweeps (eval-sized corpora, small shards) take their first local thresholds from a radix select over the
    // first tile's scores instead of sorting every list after it (cfg2: 0.48 -> 0.30 ms); on long sweeps the sorted
    // lists of the compaction are worth more than it costs (mining bench: radix select -0.5 %)
    const bool radix_first_split = DT != ABC_DT_I8 && track && (t_end - t_begin) < 64;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        qrow[nb] = qrow0 + wn * 64 + nb * 32 + l31;
        const bool valid = qrow[nb] < p.nq;
        thr[nb] = valid ? -INFINITY : INFINITY;           // invalid query rows never pass
        thr_sh[nb] = -INFINITY;
        theff[nb] = thr[nb];
        ub[nb] = (valid && p.ub) ? p.ub[qrow[nb]] : INFINITY;
        qsc[nb] = (DT == ABC_DT_I8 && valid) ? p.q_scale[qrow[nb]] : 1.0f;
        rref[nb] = (COUNT && valid) ? p.rank_ref[qrow[nb]] : INFINITY;
        cnt[nb] = 0;
        base[nb] = 0;
        rank[nb] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) top4[nb][i] = -INFINITY;
        {
            const int64_t e = qrow[nb] + p.rank_excl;          // the query's own (positive) column, if any
            excl[nb] = (COUNT && e >= 0 && e < p.nc) ? (uint32_t)e : 0xFFFFFFFFu;
        }
    }

    acc_t acc[4][2];
    unsigned dbg_app = 0, dbg_comp = 0, dbg_compn = 0;     // debug counters (ABC_DEBUG_FLAGS & 16)
    const bool stamps = (p.debug_flags & 16) != 0;         // ... and cycle stamps of the epilogue's parts
    uint64_t dbg_t[6] = {0, 0, 0, 0, 0, 0}, dbg_e[6] = {0, 0, 0, 0, 0, 0}, dbg_t0 = 0;   // all tiles / tiles 1-4 only
    bool dbg_early = false;
    const uint64_t dbg_start = stamps ? __builtin_amdgcn_s_memtime() : 0;
    const uint64_t dbg_rstart = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
#ifdef ABC_PHASE_STAMPS                                            // diagnostic build only: where a phase's cycles go
    uint64_t ps_t[4] = {0, 0, 0, 0}, ps_t0 = __builtin_amdgcn_s_memtime();
#define PSTAMP(I)                                                  \
    {                                                              \
        __builtin_amdgcn_sched_barrier(0);                         \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();        \
        ps_t[I] += now_ - ps_t0;                                   \
        ps_t0 = now_;                                              \
        __builtin_amdgcn_sched_barrier(0);                         \
    }
#else
#define PSTAMP(I)
#endif
#define STAMP(I)                                                   \
    if (stamps) {                                                  \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();        \
        dbg_t[I] += now_ - dbg_t0;                                 \
        if (dbg_early) dbg_e[I] += now_ - dbg_t0;                  \
        dbg_t0 = now_;                                             \
    }
    // candidate lists of this wave's 64 queries (slot s = nb*32 + l31; qrow0 + 256 <= nq_pad always) and its
    // append queue in LDS
    uint64_t* const lists0 = p.cand + ((int64_t)ss * p.nq_pad + qrow0 + wn * 64) * CAND_CAP;
    uint64_t* const qkey = (uint64_t*)(smem + RING_BYTES + wave * Q_BYTES);
    uint8_t* const qslt = (uint8_t*)(qkey + Q_CAP);
    float* const cs_lds = (float*)(smem + RING_BYTES + 8 * Q_BYTES + wave * CS_BYTES);
    int qn = 0;                                            // entries in the queue (wave-uniform)
    int drained_upto = -1;                                 // see wait_landed
    int next_refresh = 1;                                  // tracked mode: next tile (1-based) that refreshes the bounds
    auto lanes_below = [&](uint64_t m) {                   // number of set bits of m below this lane
        return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    // Flush the queue: 64 entries per step, one (scattered) 64-lane store.  Entry e of query slot s goes to
    // position cnt(s) + (entries of s among the lanes below); lanes l and l+32 own slots l31 and l31+32.
    auto flush_queue = [&]() {
        for (int e0 = 0; e0 < qn; e0 += 64) {
            const int e = e0 + lane;
            const bool valid = e < qn;
            const uint64_t key = valid ? qkey[e] : 0ull;
            const uint32_t sl = valid ? (uint32_t)qslt[e] : 0u;
            uint64_t same = __ballot(valid), mine5 = same, b5 = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const uint64_t bi = __ballot(valid && ((sl >> i) & 1u));
                same &= ((sl >> i) & 1u) ? bi : ~bi;
                if (i < 5) mine5 &= ((l31 >> i) & 1) ? bi : ~bi;
                else b5 = bi;
            }
            const int c0 = __shfl(cnt[0], (int)(sl & 31u), 64), c1 = __shfl(cnt[1], (int)(sl & 31u), 64);
            const int at = ((sl >> 5) ? c1 : c0) + lanes_below(same);
            if (valid) lists0[(int64_t)sl * CAND_CAP + at] = key;
            cnt[0] += __builtin_popcountll(mine5 & ~b5);
            cnt[1] += __builtin_popcountll(mine5 & b5);
        }
        qn = 0;
    };

    // ---- software pipeline over (tile, kstep) -----------------------------------------------------
    auto corpus_rsrc = [&](int tile) {
        const int64_t r0 = same_c ? 0 : (int64_t)tile * TILE;
        const int64_t left = p.nc - r0;
        return make_rsrc(p.c + r0 * rowbytes, (uint32_t)((left < TILE ? left : TILE) * rowbytes));
    };

    if (t_begin < t_end) {
        const int nph = nk;                                // phases per tile (rowbytes / 64)
        const int p_total = (t_end - t_begin) * nph;
        // issue side of the ring runs AHEAD phases in front of the compute side
        int is_tile = t_begin, is_k = 0, is_cnt = 0;
        int cs_mark = 0;                                   // int8: phases issued when the tile's scales were staged
        __amdgpu_buffer_rsrc_t rc_is = corpus_rsrc(t_begin);
        auto issue_next = [&]() {
            if (is_tile < t_end) {
                stage_issue(rc_is, rq, smem, is_cnt & (N_SLOTS - 1), wave, voff, is_k * ROWB);
                ++is_cnt;
                if (++is_k == nph) {
                    is_k = 0;
                    ++is_tile;
                    if (is_tile < t_end) rc_is = corpus_rsrc(is_tile);
                }
            }
        };
        // wait until phase `phx` has landed: at most the LDS-DMA of the phases issued after it may still be
        // outstanding (4 per phase and wave).  vmcnt counts stores too, in issue order, so an epilogue's
        // append stores (younger than the loads in flight) would turn the next counted waits into waits
        // for a store round trip (-10 % measured).  Instead an epilogue drains the ring once before its
        // first store (the loads have had a whole segment to land) and records the last phase covered:
        // waits for phases up to `drained_upto` are then no-ops.
        auto wait_landed = [&](int phx) {
            if (phx <= drained_upto) return;
            const int later = is_cnt - 1 - phx;
            if (later >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
#define READ_FRAGS(FA, FB, SB, S)                                                                            \
    {                                                                                                        \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) FB[nb] = *(const v4i*)((SB) + b_base + nb * 32 * ROWB + xs[S]); \
        _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) FA[mb] = *(const v4i*)((SB) + a_base + mb * 32 * ROWB + xs[S]); \
    }
#define MMA_FRAGS(FA, FB)                                                                                    \
    {                                                                                                        \
        _Pragma("unroll") for (int mb = 0; mb < 4; ++mb)                                                     \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) mma_step<DT>(acc[mb][nb], FA[mb], FB[nb]);      \
    }
        // Two wave groups (wm = 0 / 1; waves w and w+4 share a SIMD) run the SAME program one barrier
        // interval apart: while one group is in its read segment R(p) (fragment reads, LDS-DMA issue,
        // counted wait) its SIMD partners are in their MFMA segment M(p) or M(p-1), so the matrix pipe
        // of every SIMD always has a wave feeding it.  Hazards (q = any phase):
        //   - slot q%4 is refilled with phase q+4 in R(q+1): both groups finished R(q) at least one
        //     barrier earlier;
        //   - every wave waits for its DMA share of phase q+1 at the end of ITS R(q): that is at
        //     least one barrier before the first R(q+1) of either group.
        __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the per-query loads above (see the epilogue's note)
#pragma unroll
        for (int i = 0; i < N_SLOTS - 1; ++i) issue_next();
        wait_landed(0);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one interval behind
        __builtin_amdgcn_sched_barrier(0);

#ifdef ABC_PHASE_STAMPS
        // diagnostic build, ABC_DEBUG_FLAGS=16384 (with 1): no staging at all, the LDS-read + MFMA ceiling of this loop
        const bool no_dma = (p.debug_flags & 16384) != 0;
#else
        constexpr bool no_dma = false;
#endif
        int ph = 0;
        for (int tile = t_begin; tile < t_end; ++tile) {
            // One real vmcnt(0) per tile, through the builtin that hipcc's waitcnt pass accounts for: without
            // it the pass guards the fragment registers (shared with the epilogue's load destinations) with a
            // vmcnt(0) inside the phase loop, i.e. a drain of the ring in EVERY phase.
            __builtin_amdgcn_s_waitcnt(0x0F70);
            drained_upto = is_cnt - 1;
            if constexpr (DT == ABC_DT_I8) {
                // the 128 corpus scales of this wave's half tile ride the LDS-DMA path too (2 pieces of 256 B,
                // older than every phase load counted below; rows past nc read 0 through the descriptor's range
                // check).  Loading them in the epilogue cost 15 % of the int8 sweep in exposed latency.
                const int64_t r0 = (int64_t)tile * TILE + wm * 128;
                const int64_t left = p.nc - r0;
                const __amdgpu_buffer_rsrc_t rs = make_rsrc((const char*)(p.c_scale + (left > 0 ? r0 : 0)),
                                                            (uint32_t)((left <= 0 ? 0 : (left < 128 ? left : 128)) * 4));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR((char*)cs_lds), 4, lane * 4, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR((char*)cs_lds + 256), 4, lane * 4, 256, 0, 0);
                cs_mark = is_cnt;
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0;

            // Steady state (the issue side stays inside this tile and exactly 3 phases ahead): straight-line
            // issue + vmcnt(8).  The last 4 phases of a tile, where the issue side crosses into the next tile
            // or runs dry, take the general path with its bookkeeping.
#define PHASE_BODY(FAST)                                                                                     \
    {                                                                                                        \
        PSTAMP(3);                                                                                           \
        const char* sb = smem + (ph & (N_SLOTS - 1)) * SLOT_BYTES;                                           \
        v4i a0[4], b0[2], a1[4], b1[2];                                                                      \
        READ_FRAGS(a0, b0, sb, 0);                                                                           \
        READ_FRAGS(a1, b1, sb, 1);                                                                           \
        if (!no_dma) {                                                                                       \
            if (FAST) {                                                                                      \
                if (packed_t) stage_issue(rc_is, rq, smem, (ph + 3) & (N_SLOTS - 1), wave, voff_p, (kk + 3) * STAGE_BYTES); \
                else                                                                                         \
                stage_issue(rc_is, rq, smem, (ph + 3) & (N_SLOTS - 1), wave, voff, (kk + 3) * ROWB);         \
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                             \
            } else {                                                                                         \
                issue_next();                               /* phase ph+3 -> slot (ph-1)%4 */                \
                if (ph + 1 < p_total) wait_landed(ph + 1);                                                   \
            }                                                                                                \
        }                                                                                                    \
        PSTAMP(0);                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        PSTAMP(1);                                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        MMA_FRAGS(a0, b0);                                                                                   \
        MMA_FRAGS(a1, b1);                                                                                   \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        PSTAMP(2);                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
            int kk = 0;
            for (; kk < nph - 4; ++kk, ++ph) PHASE_BODY(true);
            is_cnt += kk;                                   // the fast phases issued one phase each
            is_k += kk;
            for (; kk < nph; ++kk, ++ph) PHASE_BODY(false);
            // ---- epilogue segment E: both groups run it in the SAME interval (group 0 skips the interval in
            // which group 1 finishes its last M), then group 1 skips one to restore the stagger
            if (wm == 0) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (stamps) { dbg_t0 = __builtin_amdgcn_s_memtime(); dbg_early = tile - t_begin < 4; }
            // filter the 128x64 accumulator block of this wave against the thresholds; survivors go to the
            // wave's append queue in LDS (scattered one-lane global stores from here cost 11 % of the sweep)
            if (!(p.debug_flags & 1)) {                     // ABC_DEBUG_FLAGS=1: contraction only (timing experiments)
            const int64_t crow_tile = (int64_t)tile * TILE + wm * 128;
            const bool partial = crow_tile + 128 > p.nc;
            const int ti = tile - t_begin + 1;
            // refresh points.  Tracked mode (pub_m <= 8): publishing a bound costs no compaction, but every
            // refresh is 2 x n_lists agent-scope (L2-bypassing) loads per lane, so the points are spaced
            // geometrically: tiles 1..4, then each about 5/4 (ABC_DEBUG_FLAGS bits 8-9: 9/8, 3/2, 2) of the last.
            // Legacy mode: lists are compacted to publish, at tiles 1,2,3,4,8,16,32,...
            bool refresh;
            if (track) {
                refresh = ti == next_refresh;
                if (refresh) {
                    const int rmode = (p.debug_flags >> 8) & 3;
                    const int step = rmode == 0 ? ti >> 2 : rmode == 1 ? ti >> 3 : rmode == 2 ? ti >> 1 : ti;
                    next_refresh = ti + (step > 1 ? step : 1);
                }
            } else refresh = (ti & (ti - 1)) == 0 || ti == 3;
            refresh = refresh && p.n_lists > 1;
            if constexpr (DT == ABC_DT_I8) {
                // the tile's scales (2 LDS-DMA pieces issued before this tile's phases) must have landed: they are
                // older than the 12 youngest operations once 3 phases (4 pieces each) have been issued after them
                if (is_cnt - cs_mark >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // short rows, or the last tile of the split
            }
            if (partial) {
                // the last tile of the corpus: rows past nc (zero-filled by the staging) must never pass or count.
                // Mask them ONCE here (-inf scores; for int8 a NaN scale below) instead of testing every element.
                if constexpr (DT != ABC_DT_I8) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const bool oob = (uint32_t)(crow_tile + mb * 32 + 4 * h) + (r & 3) + 8 * (r >> 2) >= (uint32_t)p.nc;
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb) acc[mb][nb][r] = oob ? -INFINITY : acc[mb][nb][r];
                        }
                }
            }
            if constexpr (DT != ABC_DT_I8) {
                if (radix_first_split && ti == 1) {
                    // First tile, empty lists: every score would pass, and giving the lists a threshold used to take a
                    // sort of all 64 of them.  The wave still holds each query's 128 scores in registers (64 in lane l,
                    // 64 in lane l+32): a 16-bit radix select on the ordered bit pattern finds a value that at least k
                    // of them reach, for all 64 queries at once.  It becomes the local threshold BEFORE the filter, so
                    // this tile appends about k entries per list and nothing is compacted.
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        uint32_t prefix = 0;
                        for (int bit = 31; bit >= 16; --bit) {
                            const uint32_t cand = prefix | (1u << bit);
                            const float fc = orderable_f32(cand);
                            int c = 0;
#pragma unroll
                            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float x = acc[mb][nb][r];
                                    c += (x >= fc && !(x > ub[nb])) ? 1 : 0;
                                }
                            c += __shfl_xor(c, 32, 64);
                            if (c >= p.k) prefix = cand;
                        }
                        if (prefix != 0 && qrow[nb] < p.nq) {
                            // entries >= value(prefix) must pass `v > thr`: take the float just below it
                            const float fv = orderable_f32(prefix);
                            uint32_t u = __float_as_uint(fv);
                            if (fv == 0.f) u = 0x80000001u;
                            else if (u != 0xFF800000u && u != 0x7F800000u) u = (u >> 31) ? u + 1 : u - 1;
                            thr[nb] = __uint_as_float(u);
                            theff[nb] = fmaxf(thr[nb], thr_sh[nb]);
                        }
                    }
                }
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const uint32_t cbase = (uint32_t)(crow_tile + mb * 32 + 4 * h);
                float csc[16];
                if constexpr (DT == ABC_DT_I8) {
                    // rows mb*32 + 8j + 4h + {0..3}: four consecutive floats of the staged scales per j
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const v4f c4 = *(const v4f*)(cs_lds + mb * 32 + 8 * j + 4 * h);
#pragma unroll
                        for (int t = 0; t < 4; ++t) csc[4 * j + t] = c4[t];
                    }
                    if (partial) {                           // NaN scale -> NaN score: fails every comparison
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (cbase + (r & 3) + 8 * (r >> 2) >= (uint32_t)p.nc) csc[r] = __builtin_nanf("");
                    }
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if constexpr (DT == ABC_DT_I8) v[r] = (float)acc[mb][nb][r] * (qsc[nb] * csc[r]);
                        else v[r] = acc[mb][nb][r];
                    }
                    if constexpr (COUNT) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint32_t ci = cbase + (r & 3) + 8 * (r >> 2);
                            rank[nb] += (v[r] > rref[nb] && ci != excl[nb]) ? 1u : 0u;
                        }
                    }
                    // quick reject, two levels: the 32x32 block, then its four groups of 4 rows
                    float gm[4];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        gm[g4] = fmaxf(fmaxf(v[4 * g4], v[4 * g4 + 1]), fmaxf(v[4 * g4 + 2], v[4 * g4 + 3]));
                    const float m = fmaxf(fmaxf(gm[0], gm[1]), fmaxf(gm[2], gm[3]));
                    if (__ballot(m > theff[nb]) == 0ull || (p.debug_flags & 2)) continue;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        if (__ballot(gm[g4] > theff[nb]) == 0ull) continue;
#pragma unroll
                        for (int r = 4 * g4; r < 4 * g4 + 4; ++r) {
                            const uint32_t ci = cbase + (r & 3) + 8 * (r >> 2);
                            const bool pass = (v[r] > theff[nb]) && !(v[r] > ub[nb]);
                            const uint64_t bal = __ballot(pass);
                            if (bal != 0ull) {
                                {   // keep the lane's 4 best passed scores of this query
                                    float x = pass ? v[r] : -INFINITY, hi;
                                    hi = fmaxf(top4[nb][0], x); x = fminf(top4[nb][0], x); top4[nb][0] = hi;
                                    hi = fmaxf(top4[nb][1], x); x = fminf(top4[nb][1], x); top4[nb][1] = hi;
                                    hi = fmaxf(top4[nb][2], x); x = fminf(top4[nb][2], x); top4[nb][2] = hi;
                                    top4[nb][3] = fmaxf(top4[nb][3], x);
                                }
                                const int npass = __builtin_popcountll(bal);
                                if (qn + npass <= Q_CAP) {
                                    const int pos = qn + lanes_below(bal);
                                    if (pass) {
                                        qkey[pos] = make_key(v[r], ci);
                                        qslt[pos] = (uint8_t)(nb * 32 + l31);
                                    }
                                    qn += npass;
                                } else {
                                    // queue full (first tiles, adversarial data): append straight to the lists;
                                    // lanes l and l+32 append to the same list, lower half first
                                    const uint32_t sel = h ? (uint32_t)bal : (uint32_t)(bal >> 32);
                                    const int mine = pass ? 1 : 0;
                                    const int other = (int)((sel >> l31) & 1u);
                                    const int pos = cnt[nb] + (h ? other : 0);
                                    if (drained_upto != is_cnt - 1) {  // first store of this epilogue: drain the ring
                                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                        drained_upto = is_cnt - 1;
                                    }
                                    if (pass) lists0[(int64_t)(nb * 32 + l31) * CAND_CAP + pos] = make_key(v[r], ci);
                                    cnt[nb] += mine + other;
                                }
                                dbg_app += pass ? 1 : 0;
                            }
                        }
                    }
                }
            }
            STAMP(0);
            // ---- once per tile: flush a queue that is more than half full (always at a refresh point)
            if (qn > Q_CAP / 2 || (refresh && (!track || ti <= 4) && qn > 0)) {   // (a flush without need is harmless)
                if (drained_upto != is_cnt - 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    drained_upto = is_cnt - 1;
                }
                flush_queue();
            }
            STAMP(1);
            // ---- compact every list that could overflow before the next check (until then at most 128 direct
            // appends and one flush of at most Q_CAP entries; LIST_TRIG = CAND_CAP - 128 - Q_CAP) and, at a refresh
            // point, every list that has grown since its last compaction, so that it publishes a fresh bound
            {
                // tracked mode, int8: every grown list is compacted once, after the first tile, to give it a local
                // threshold for the tiles that run before the first shared bounds arrive (more early compactions
                // cost more than they filter: -2 % at 200 tiles per split)
                const int early_last = radix_first_split ? 0 : 1;          // 0: the radix select above did it
                const bool grown = refresh && (!track || ti <= early_last);
                const bool n0 = cnt[0] > LIST_TRIG || (grown && cnt[0] > base[0]);
                const bool n1 = cnt[1] > LIST_TRIG || (grown && cnt[1] > base[1]);
                uint64_t need = (__ballot(n0) & 0xFFFFFFFFull) | (__ballot(n1) << 32);
                if (need) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    while (need) {
                        const int jb = __builtin_ctzll(need);
                        need &= need - 1;
                        const int nbq = jb >> 5, j = jb & 31;
                        const int n = __shfl(nbq ? cnt[1] : cnt[0], j, 64);
                        uint64_t* bj = lists0 + (int64_t)jb * CAND_CAP;
                        float nt, pv; int nc2;
                        dbg_comp += 1; dbg_compn += n;
                        compact_list(bj, n, p.k, p.pub_m, lane, nt, nc2, pv, false);
                        if (lane == 0)
                            __hip_atomic_store(p.pub + (qrow0 + wn * 64 + jb) * p.n_lists + ss, pv,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (l31 == j) {
                            if (nbq) { thr[1] = nt; cnt[1] = nc2; base[1] = nc2; } else { thr[0] = nt; cnt[0] = nc2; base[0] = nc2; }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    drained_upto = is_cnt - 1;              // the fence drained every load (and store)
                    theff[0] = fmaxf(thr[0], thr_sh[0]);
                    theff[1] = fmaxf(thr[1], thr_sh[1]);
                }
            }
            STAMP(2);
            // ---- at a refresh point: publish this list's bound (tracked mode; in legacy mode the compaction
            // above did) and refresh the shared lower bound of each query's global k-th best from what all the
            // lists (other waves, other workgroups) have published.  Stale values are valid.
            if (refresh && !(p.debug_flags & 8)) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int64_t qr = qrow[nb] < p.nq ? qrow[nb] : 0;
                    if (track) {
                        // merge the two lanes' sorted 4 (bitonic: a_i vs b_{3-i}), sort both halves, pick rank pub_m
                        float o[4], hi4[4], lo4[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = __shfl_xor(top4[nb][i], 32, 64);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            hi4[i] = fmaxf(top4[nb][i], o[3 - i]);
                            lo4[i] = fminf(top4[nb][i], o[3 - i]);
                        }
                        auto sort4_bitonic = [](float (&t)[4]) {        // bitonic input -> descending
                            float a = fmaxf(t[0], t[2]), c = fminf(t[0], t[2]), b = fmaxf(t[1], t[3]), d = fminf(t[1], t[3]);
                            t[0] = fmaxf(a, b); t[1] = fminf(a, b); t[2] = fmaxf(c, d); t[3] = fminf(c, d);
                        };
                        sort4_bitonic(hi4);
                        sort4_bitonic(lo4);
                        float pv = lo4[3];
                        const int pm = p.pub_m;
                        pv = pm == 1 ? hi4[0] : pm == 2 ? hi4[1] : pm == 3 ? hi4[2] : pm == 4 ? hi4[3]
                           : pm == 5 ? lo4[0] : pm == 6 ? lo4[1] : pm == 7 ? lo4[2] : lo4[3];
                        if (h == 0 && qrow[nb] < p.nq)
                            __hip_atomic_store(p.pub + qr * p.n_lists + ss, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const float* pp = p.pub + qr * p.n_lists;
                    float mn = INFINITY;
                    if ((p.n_lists & 15) == 0) {
                        // a query's bounds are one or more 64-byte rows: four system-scope (L2-bypassing) 16-byte
                        // loads per row instead of sixteen 4-byte ones -- every such load is its own memory request
                        const __amdgpu_buffer_rsrc_t rpub = make_rsrc((const char*)p.pub, (uint32_t)(p.nq_pad * p.n_lists * 4));
                        const int vo = (int)(qr * p.n_lists) * 4;
                        for (int j0 = 0; j0 < p.n_lists; j0 += 16) {
                            v4i t4[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                t4[u] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rpub, vo + (j0 + 4 * u) * 4, 0, 16));
#pragma unroll
                            for (int u = 0; u < 4; ++u)
#pragma unroll
                                for (int e = 0; e < 4; ++e) mn = fminf(mn, __int_as_float(t4[u][e]));
                        }
                    } else {
                        for (int j0 = 0; j0 < p.n_lists; j0 += 16) {     // 16 agent-scope loads in flight
                            float t16[16];
#pragma unroll
                            for (int u = 0; u < 16; ++u)
                                t16[u] = __hip_atomic_load(pp + (j0 + u < p.n_lists ? j0 + u : 0), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                            for (int u = 0; u < 16; ++u) mn = fminf(mn, t16[u]);
                        }
                    }
                    // v >= mn  <=>  v > pred(mn): keep ties with the bound (the merge settles them by index)
                    uint32_t u = __float_as_uint(mn);
                    if (mn == 0.f) u = 0x80000001u;
                    else if (u != 0xFF800000u && u != 0x7F800000u) u = (u >> 31) ? u + 1 : u - 1;
                    thr_sh[nb] = (qrow[nb] < p.nq) ? __uint_as_float(u) : INFINITY;
                }
                // every register-destination load of this epilogue has returned; saying so with the builtin
                // (which hipcc's waitcnt pass accounts for, unlike inline asm) keeps it from guarding the
                // fragment registers with a vmcnt(0) INSIDE the phase loop, which would drain the ring
                __builtin_amdgcn_s_waitcnt(0x0F70);
                drained_upto = is_cnt - 1;
            }
            theff[0] = fmaxf(thr[0], thr_sh[0]);
            theff[1] = fmaxf(thr[1], thr_sh[1]);
            STAMP(3);
            }
            // ---- end of the epilogue segment E
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            STAMP(4);
            if (wm == 1) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();          // balances group 1's extra barrier
    }

    if (p.debug_flags & 16) {
        unsigned a = dbg_app;
        for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
        if (lane == 0) {
            atomicAdd((unsigned long long*)p.dbg + 0, (unsigned long long)a);
            atomicAdd((unsigned long long*)p.dbg + 1, (unsigned long long)dbg_comp);
            atomicAdd((unsigned long long*)p.dbg + 2, (unsigned long long)dbg_compn);
            for (int i = 0; i < 5; ++i) atomicAdd((unsigned long long*)p.dbg + 3 + i, (unsigned long long)dbg_t[i]);
            for (int i = 0; i < 5; ++i) atomicAdd((unsigned long long*)p.dbg + 14 + i, (unsigned long long)dbg_e[i]);
            atomicAdd((unsigned long long*)p.dbg + 8, (unsigned long long)(__builtin_amdgcn_s_memtime() - dbg_start));
            atomicAdd((unsigned long long*)p.dbg + 9, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - dbg_rstart));
#ifdef ABC_PHASE_STAMPS
            for (int i = 0; i < 4; ++i) atomicAdd((unsigned long long*)p.dbg + 10 + i, (unsigned long long)ps_t[i]);
#endif
        }
    }
    // ---- end of the sweep: sort every list, pad to k, publish the rank counts -----------------------
    flush_queue();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        for (int j = 0; j < 32; ++j) {
            const int64_t qr = qrow0 + wn * 64 + nb * 32 + j;
            if (qr >= p.nq) break;                         // uniform
            const int n = __shfl(cnt[nb], j, 64);
            uint64_t* bj = p.cand + ((int64_t)ss * p.nq_pad + qr) * CAND_CAP;
            float nt, pv; int nc2;
            compact_list(bj, n, p.k, p.pub_m, lane, nt, nc2, pv, true);
        }
        if constexpr (COUNT) {
            const uint32_t tot = rank[nb] + __shfl_xor(rank[nb], 32, 64);
            if (h == 0 && qrow[nb] < p.nq) p.rank_part[(int64_t)ss * p.nq_pad + qrow[nb]] = tot;
        }
    }
}


#include "topk_sweep4.inc"
