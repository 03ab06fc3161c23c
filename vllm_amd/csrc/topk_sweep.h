// Internal interface of the fused score-GEMM + top-k sweep (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/abcsearch.h"

#define CAND_CAP 1024  // candidate-list capacity per (split, query): 8 KiB of 8-byte keys


struct SweepParams {
    const char* q;          // [nq, rowbytes] queries of this launch
    const char* c;          // [nc, rowbytes] corpus
    int64_t nq, nc;
    int rowbytes;           // d * sizeof(elem), multiple of 128
    int k;
    int n_qblocks;          // ceil(nq / 256) of this launch
    int n_splits;           // corpus splits; grid = n_qblocks * n_splits
    int n_tiles;            // ceil(nc / 256)
    int64_t nq_pad;         // list stride between list sets (queries per launch, padded)
    const float* ub;        // nullable [nq]
    const float* q_scale;   // int8 only
    const float* c_scale;   // int8 only
    const float* rank_ref;  // COUNT only
    int64_t rank_excl;      // COUNT only: column (query row + rank_excl) is not counted
    int debug_flags;        // ABC_DEBUG_FLAGS at load time / abc_debug_set("flags") (timing experiments only; 0 in production;
                            // 1 and 2 make results WRONG and exist only in -DABC_DIAG builds):
                            //   1      contraction only, no epilogue       2      quick reject only, nothing appended
                            //   8      never refresh the shared bounds     16     debug counters + cycle stamps to stderr
                            //   bits 8-9   refresh spacing 5/4 (0), 9/8, 3/2, 2
                            //   bits 12-13 XCDs that share a group of query blocks: 2 (0, default), 1, 4, 8
                            // ABC_PLAN_QB=<n> / abc_debug_set("plan_qb") overrides the query blocks per launch; ABC_LIB picks
                            // another .so (A/B runs)
    int n_lists;            // = n_splits: one candidate list per (split, query)
    int pub_m;              // ceil(k / n_lists): rank each list publishes
    float* pub;             // [nq_pad][n_lists] published bounds (start at -inf)
    unsigned long long* dbg; // debug counters
    uint64_t* cand;         // [n_splits][nq_pad][CAND_CAP] packed keys
    uint32_t* rank_part;    // [n_splits][nq_pad]
    uint32_t* list_cnt;     // [n_splits][nq_pad] entries each list holds when the sweep ends (unsorted; the merge selects)
};

struct SweepPlan {
    int n_tiles;
    int qblocks_per_launch;
    int n_splits;
    int64_t nq_pad;
};

void sweep_plan(int64_t nq, int64_t nc, int k, SweepPlan* pl);
size_t sweep_workspace_bytes(int64_t nq, int64_t nc, int k);
int sweep_run(int dt, const void* q, const float* q_scale, int64_t nq, const void* c, const float* c_scale, int64_t nc,
              int d, int k, const float* upper_bound, int64_t idx_offset, float* out_val, int64_t* out_idx, uint64_t* out_keys,
              const float* rank_ref, int64_t rank_self_offset, int64_t* out_rank, void* workspace, hipStream_t st,
              hipError_t* herr);
hipError_t launch_sweep_bf16(bool count, const SweepParams& p, int grid, hipStream_t st);
hipError_t launch_sweep_i8(bool count, const SweepParams& p, int grid, hipStream_t st);
hipError_t launch_sweep_f32(bool count, const SweepParams& p, int grid, hipStream_t st);
int rowdot_run(int dt, const void* q, const float* qs, const void* c, const float* cs, int64_t n, int d, float* out,
               hipStream_t st, hipError_t* herr);
int merge_keys_run(const uint64_t* keys, int parts, int64_t nq, int k, float* out_val, int64_t* out_idx, hipStream_t st,
                   hipError_t* herr);
int sweep_debug_set(const char* name, long value);
long sweep_debug_get(const char* name);
int merge_lists_run(const float* vals, const int64_t* idx, int parts, int64_t nq, int k, float* out_val,
                    int64_t* out_idx, hipStream_t st, hipError_t* herr);
