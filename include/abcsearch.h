/*
 * abcsearch.h -- C ABI of libabcsearch.so (MI355X / gfx950 only).
 *
 * Exact similarity search over embeddings (score = Q.C^T, per-row top-k fused into the
 * MFMA GEMM epilogue, optional per-row upper bound = the hard-negative rule) and the
 * temperature-scaled, label-smoothed InfoNCE forward/backward.
 *
 * The reference (Ben-Schneider-code/vllm @ 2025-03-01) has no FFI layer: its boundary is a set of
 * Python functions calling torch ops.  Each entry point below names the reference call site whose
 * arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates, passes data_ptr());
 *     the library borrows it for the enqueued work and never frees or retains it;
 *   - matrices are row-major contiguous, base 16-byte aligned; d*sizeof(elem) must be a multiple
 *     of 128 (bf16: d%64==0, int8: d%128==0, f32: d%32==0);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no implicit device sync;
 *   - return 0 = ok, negative = error (ABC_E_*); message via abc_last_error() (thread-local);
 *   - no exceptions, no C++ types, no torch types cross this boundary.
 */
#ifndef ABCSEARCH_H
#define ABCSEARCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABC_OK 0
#define ABC_E_INVALID (-1)   /* bad argument (shape, alignment, k, null pointer) */
#define ABC_E_WORKSPACE (-2) /* workspace too small */
#define ABC_E_HIP (-3)       /* HIP runtime error at enqueue time */

#define ABC_DT_BF16 0
#define ABC_DT_I8 1
#define ABC_DT_F32 2

#define ABC_MAX_K 128
#define ABC_NO_SELF INT64_MIN

const char* abc_version(void);
const char* abc_last_error(void);

/* Bytes of device workspace abc_topk_* needs for this problem (0 on invalid arguments). */
size_t abc_topk_workspace_bytes(int64_t nq, int64_t nc, int d, int k, int dtype);

/*
 * Exact top-k of q @ c^T per query row, best first; ties: larger score, then smaller index.
 * Replaces  scores = query_stack @ candidate_stack.T; torch.topk(scores, k, dim=1)
 *   evaluate/eval_mscoco.py:40-41 (same in eval_flickr30k.py:40-41, eval_vg_instruct.py, mmeb_*.py),
 *   visualization/neg_mine.py:26-29,
 * and, with upper_bound, the mask + sort of pretrain/create_negative_index.py:47-51
 *   (scores[scores > thr] = -inf; argsort(descending)[:k]).
 *
 *   q [nq,d], c [nc,d]       inputs of the dtype in the name
 *   upper_bound [nq] f32     nullable; scores strictly greater than upper_bound[i] are dropped
 *   idx_offset               added to every returned index (global index of c's row 0 on a shard)
 *   out_val [nq,k] f32       scores; missing entries (fewer than k survivors) = -inf
 *   out_idx [nq,k] i64       indices; missing entries = -1
 *   rank_ref [nq] f32 / out_rank [nq] i64   nullable pair; out_rank[i] = #{j != i + rank_self_offset :
 *                            score_ij > rank_ref[i]} (visualization/neg_mine.py:37-38, the compare-count of
 *                            batch()); column i + rank_self_offset is query i's own positive, whose score
 *                            equals rank_ref[i] up to rounding and is never counted; ABC_NO_SELF = none
 *   workspace                >= abc_topk_workspace_bytes(...)
 * The score matrix is never written to memory.
 */
int abc_topk_bf16(const void* q, int64_t nq, const void* c, int64_t nc, int d, int k,
                  const float* upper_bound, int64_t idx_offset, float* out_val, int64_t* out_idx,
                  const float* rank_ref, int64_t rank_self_offset, int64_t* out_rank,
                  void* workspace, size_t workspace_bytes, void* stream);

int abc_topk_f32(const void* q, int64_t nq, const void* c, int64_t nc, int d, int k,
                 const float* upper_bound, int64_t idx_offset, float* out_val, int64_t* out_idx,
                 const float* rank_ref, int64_t rank_self_offset, int64_t* out_rank,
                 void* workspace, size_t workspace_bytes, void* stream);

/* int8 with per-vector scales: score_ij = float(sum_k q_ik*c_jk) * (q_scale[i]*c_scale[j])
 * (the build's int8 rule; no counterpart in the reference). */
int abc_topk_i8(const void* q, const float* q_scale, int64_t nq, const void* c, const float* c_scale,
                int64_t nc, int d, int k,
                const float* upper_bound, int64_t idx_offset, float* out_val, int64_t* out_idx,
                const float* rank_ref, int64_t rank_self_offset, int64_t* out_rank,
                void* workspace, size_t workspace_bytes, void* stream);

/*
 * out[i] = dot(q[i,:], c[i,:]) in fp32, i < n.  Replaces scores[idx] of
 * pretrain/create_negative_index.py:48 and the bmm of visualization/neg_mine.py:32-35.
 * For every shape abc_topk_* accepts (d*sizeof(elem) % 128 == 0) the value is contracted by the same MFMA
 * instruction (16 x 16 tiles), fragments and k order as the sweep, i.e. it is BIT-IDENTICAL to the score abc_topk_* sees for
 * column i of query i -- the reference takes scores[idx] out of the same GEMV it masks (:47-49).
 */
int abc_rowdot_bf16(const void* q, const void* c, int64_t n, int d, float* out, void* stream);
int abc_rowdot_f32(const void* q, const void* c, int64_t n, int d, float* out, void* stream);
int abc_rowdot_i8(const void* q, const float* q_scale, const void* c, const float* c_scale, int64_t n, int d,
                  float* out, void* stream);

/*
 * Merge `parts` sorted top-k lists per query into one (the cross-shard / cross-GPU merge after the
 * all-gather; no counterpart in the reference, whose search is single-GPU).
 *   vals [parts,nq,k] f32, idx [parts,nq,k] i64 (already global, -1 = missing) -> out_val/out_idx [nq,k]
 *   idx_limit   exclusive upper bound of the indices in idx (the caller's total row count).  The merge packs
 *               (score, index) into 64-bit keys with a 32-bit index field: idx_limit > 2^32-1 is rejected
 *               (ABC_E_INVALID) instead of wrapping silently.
 */
int abc_topk_merge(const float* vals, const int64_t* idx, int parts, int64_t nq, int k, int64_t idx_limit,
                   float* out_val, int64_t* out_idx, void* stream);

/*
 * The same search with PACKED output, and the merge of packed lists: the multi-GPU exchange (no
 * counterpart in the reference, whose search runs on one GPU).  A shard's result travels as ONE tensor of 8 bytes per
 * entry, so a chunk's exchange is one all-gather of k * 8 bytes per query:
 *   key = orderable(score) << 32 | (0xFFFFFFFF - global index),  0 = no entry;  larger key = better (ties: smaller index)
 *   orderable(s) = bits(s) ^ (bits(s) >> 31 ? 0xFFFFFFFF : 0x80000000)
 * abc_topk_packed: dtype = ABC_DT_*, q_scale / c_scale only for int8; idx_offset + nc must not exceed 2^32 - 1
 *   (ABC_E_INVALID otherwise: the index field has 32 bits).  out_keys [nq,k] u64, best first.
 * abc_topk_merge_packed: keys [parts,nq,k] u64 (each list sorted descending) -> out_val [nq,k] f32 (-inf = missing),
 *   out_idx [nq,k] i64 (-1 = missing).
 */
int abc_topk_packed(int dtype, const void* q, const float* q_scale, int64_t nq, const void* c, const float* c_scale,
                    int64_t nc, int d, int k, const float* upper_bound, int64_t idx_offset, uint64_t* out_keys,
                    void* workspace, size_t workspace_bytes, void* stream);
int abc_topk_merge_packed(const uint64_t* keys, int parts, int64_t nq, int k, float* out_val, int64_t* out_idx,
                          void* stream);

/*
 * Diagnostics hook for tests and timing scripts (no counterpart in the reference; not for production use: it
 * changes process-wide settings, do not call it while another thread is inside abc_topk_*):
 *   "flags"   SweepParams::debug_flags (cycle stamps, refresh spacing, XCD grouping; result-altering bits are
 *             masked out unless the library was built with -DABC_DIAG)
 *   "plan_qb" query blocks per launch (0 = the library's plan)
 * The same knobs are read from ABC_DEBUG_FLAGS / ABC_PLAN_QB once, when the library is loaded.
 */
int abc_debug_set(const char* name, int64_t value);
int64_t abc_debug_get(const char* name);

/*
 * Symmetric per-vector absmax int8 quantiser: scale = max|x|/127 (1 for an all-zero
 * row), q = clamp(rint(x/scale), -127, 127), round-half-even.  x [n,d] f32 -> out_q [n,d] i8, out_scale [n].
 */
int abc_quantize_i8(const float* x, int64_t n, int d, int8_t* out_q, float* out_scale, void* stream);

/* f32 -> bf16 (round-to-nearest-even) cast of an [n] array; embeddings saved as fp32 query.pt/cand.pt. */
int abc_cast_bf16(const float* x, int64_t n, void* out_bf16, void* stream);

/*
 * InfoNCE forward.  Replaces model/abc_util.py:47-54:
 *   score = (q @ p^T) / temp; loss = cross_entropy(score, arange(B), label_smoothing=eps); argmax.
 *   q [B,d] f32, p [C,d] f32 (C >= B; row i of p is the positive of query i)
 *   temp [1] f32 DEVICE scalar (Temperature.temp, model/modeling_abc.py:36-43; score = s / temp)
 *   out_loss [1] f32 (mean over rows), out_lse [B] f32 (logsumexp of score rows),
 *   out_argmax [B] i32, out_acc [1] f32 or NULL: accuracy = (argmax == arange(B)).sum() / B (abc_util.py:52-54),
 *   workspace >= abc_infonce_workspace_bytes(B,C,d)
 */
size_t abc_infonce_workspace_bytes(int B, int C, int d);
int abc_infonce_fwd(const float* q, const float* p, const float* temp, float eps, int B, int C, int d,
                    float* out_loss, float* out_lse, int32_t* out_argmax, float* out_acc,
                    void* workspace, size_t workspace_bytes, void* stream);
/*
 * The forward for callers that cannot allocate.  No counterpart in the reference; it exists for callers that must
 * not call hipMalloc or hand over a workspace: code under HIP graph capture, tight evaluation loops, harnesses that
 * forbid allocation.  Same arithmetic and results as abc_infonce_fwd, bit for bit; the scratch abc_infonce_fwd takes
 * as `workspace` lives in a 16 MiB device array inside the library (one copy per GPU, held until the process ends).
 * Shapes whose abc_infonce_workspace_bytes exceeds abc_infonce_noalloc_max_bytes() are refused with ABC_E_WORKSPACE
 * (B=128 x C=8192, the 8-GPU form of BASELINE configs[3], needs 8.3 MiB).  Calls no hipMalloc / hipFree.
 * Every call on one GPU shares that GPU's array: calls on ONE stream run in order; a call on another stream while the
 * previous call has not finished (an event recorded behind every call) is refused with ABC_E_INVALID: it would race on
 * the array.  Two host threads calling at the same moment are not detected: give the entry point one owner per GPU.
 * The array used is the CURRENT device's: `stream` must belong to the current device, else ABC_E_INVALID.
 * The first use on a device resolves the array's address (hipGetSymbolAddress: loads the code object, which may
 * allocate and must not happen under stream capture): call abc_infonce_noalloc_max_bytes() once per device, with that
 * device current, before capturing.
 * There is no backward for it: the logits stay in the library's array only until the next call.
 */
size_t abc_infonce_noalloc_max_bytes(void);
int abc_infonce_fwd_noalloc(const float* q, const float* p, const float* temp, float eps, int B, int C, int d,
                            float* out_loss, float* out_lse, int32_t* out_argmax, float* out_acc, void* stream);

/*
 * InfoNCE backward (closed form of the autograd of model/abc_util.py:47-51):
 *   g = (softmax(z) - [(1-eps)1{j=i} + eps/C]) * grad_out / B
 *   dq = (g/temp) p ; dp_local = ((g/temp)^T q)[:local_rows] ; d_temp = -sum(g*s)/temp^2
 * `workspace` must be the buffer abc_infonce_fwd was given for the same (q, p): it carries the logits.
 * Only the first local_rows rows of p receive a gradient (remote all-gathered blocks are constants,
 * model/abc_util.py:19-21).  grad_out is a device scalar.
 */
int abc_infonce_bwd(const float* q, const float* p, const float* lse, const float* temp, float eps,
                    const float* grad_out, int B, int C, int d, int local_rows,
                    float* dq, float* dp_local, float* d_temp,
                    void* workspace, size_t workspace_bytes, void* stream);

/*
 * Embedding epilogue: the step that produces the vectors the search and the loss consume.
 *
 * Mean pool.  Replaces get_mean_token_embed, model/abc_util.py:33-41:
 *   out[b,h] = (1/S) sum_s [ids[b,s] != pad_id] * instr[b,s] * hidden[b,s,h]     (divides by S, as torch.mean)
 *   hidden [B,S,H] bf16 or f32, ids [B,S] i64, instr [B,S] f32 nullable (the instruction mask), out [B,H] f32.
 *   round_bf16 != 0 (bf16 form only): round the instruction-mask product and the mean to bf16, as the
 *   reference's bf16 tensors do; 0 keeps the fp32 mean.  H % 8 == 0.
 */
int abc_mean_pool_bf16(const void* hidden, const int64_t* ids, int64_t pad_id, const float* instr, int B, int S, int H,
                       int round_bf16, float* out, void* stream);
int abc_mean_pool_f32(const void* hidden, const int64_t* ids, int64_t pad_id, const float* instr, int B, int S, int H,
                      float* out, void* stream);

/*
 * Projection head + normalisation.  Replaces MLP.forward, model/modeling_abc.py:28-34, and
 * F.normalize(mlp_head(x).float(), dim=-1), model/modeling_abc.py:151-154 / :178-179 (inference):
 *   y = x + W2 selu(W1 x + b1) + b2 ;  out = y / max(||y||_2, 1e-12)   (normalize = 0: out = y)
 *   x [B,H] f32, w1 [Hd,H], b1 [Hd] nullable, w2 [H,Hd], b2 [H] nullable (nn.Linear layouts), fp32 throughout.
 *   out_f32 [B,H]; optional out_bf16 [B,H] (RNE) and out_i8 [B,H] + out_scale [B] (the abc_quantize_i8 rule),
 *   i.e. the corpus forms abc_topk_bf16 / abc_topk_i8 contract in.  H % 8 == 0, Hd % 8 == 0.
 */
size_t abc_embed_head_workspace_bytes(int B, int H, int Hd);
int abc_embed_head_f32(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                       int B, int H, int Hd, int normalize, float* out_f32, void* out_bf16, int8_t* out_i8,
                       float* out_scale, float* out_h1, float* out_norm, void* workspace, size_t workspace_bytes,
                       void* stream);
/*
 * out_h1 [B,Hd] and out_norm [B] (both nullable) are what the backward needs from the forward: selu(W1 x + b1) and ||y||_2.
 *
 * Backward of the head and of the mean pool: what autograd runs behind model/modeling_abc.py:148-154 in every training
 * step (mean-pool -> mlp_head -> .float() -> F.normalize), between the loss's d emb (abc_infonce_bwd) and the backbone.
 *   dy = (g - out (out . g)) / ||y||  (g / 1e-12 where the clamp is active; g when normalize = 0)
 *   dW2 = dy^T h1, db2 = colsum(dy), da1 = (dy W2) * selu'(a1), dW1 = da1^T x, db1 = colsum(da1), dx = dy + da1 W1
 *   grad_out, out [B,H], norm [B], x [B,H], h1 [B,Hd] f32 (out, h1, norm as abc_embed_head_f32 returned them);
 *   dx [B,H], dw1 [Hd,H], db1 [Hd], dw2 [H,Hd], db2 [H]: each nullable (not computed); fp32, overwritten (not accumulated).
 *   d hidden[b,s,:] = ((dx[b,:] / S) * [ids[b,s] != pad]) * instr[b,s], every position written, in hidden's dtype.
 */
size_t abc_embed_head_bwd_workspace_bytes(int B, int H, int Hd);
int abc_embed_head_bwd(const float* grad_out, const float* out, const float* norm, const float* x, const float* h1,
                       const float* w1, const float* w2, int B, int H, int Hd, int normalize,
                       float* dx, float* dw1, float* db1, float* dw2, float* db2,
                       void* workspace, size_t workspace_bytes, void* stream);
int abc_mean_pool_bwd_bf16(const float* dx, const int64_t* ids, int64_t pad_id, const float* instr, int B, int S, int H,
                           void* dhidden, void* stream);
int abc_mean_pool_bwd_f32(const float* dx, const int64_t* ids, int64_t pad_id, const float* instr, int B, int S, int H,
                          float* dhidden, void* stream);

/*
 * negatives.json text, host code (no GPU, no stream).  Replaces the serialisation half of
 * pretrain/create_negative_index.py:40-43 (json.dump(negative_dict, f) over {int idx: [100 ints]}): the bytes json.dump
 * writes for rows first_row .. first_row + n_rows - 1 -- `"17": [3, 5, 8]` entries joined by ", ", keys as strings, no
 * braces (the caller writes "{", "}" and the ", " between blocks).  idx [n_rows, k] i64 HOST memory, entries < 0 dropped.
 * Returns the byte count written to out[0 .. cap), or -(bytes needed) when cap is too small (nothing written).
 */
int64_t abc_format_negatives(const int64_t* idx, int64_t n_rows, int k, int64_t first_row, char* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* ABCSEARCH_H */
