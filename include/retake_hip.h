/*
 * retake_hip.h — C ABI of libretake_hip.so: the MI355X (gfx950) implementation of ReTaKe's
 * DPSelect + PivotKV hot path.
 *
 * The reference (SCZwangxiao/video-ReTaKe) is pure Python: its "FFI" for this path is the set of
 * torch ATen calls inside two functions.  Each entry point below replaces one group of those call
 * sites; the reference-side binding is a ctypes stub, and the shipped Python
 * package `retake` (video-retake_amd/retake) is that binding behind the reference's own names
 *   retake.visual_compression.memory_bank_compress_keyframe   (visual_compression.py:86-177)
 *   retake.longvideo_cache.PivotKVCache.update                (longvideo_cache.py:217-323)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless marked host;
 *   - the library never allocates or frees device memory: outputs and workspaces are caller-owned;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     re-entrant.  Process-wide state, all of it thread-safe: the thread-local error string; the opt-in
 *     measurement hooks at the end of this header (off by default, mutex-guarded); a memo of which devices
 *     have had the > 64 KiB dynamic-LDS opt-in applied to the score kernels (one atomic bit per device);
 *     a mutex-guarded memo of exhaustive bf16 reciprocal checks keyed by attention_scaling^2.  No results
 *     depend on any of it, and the library reads no environment variables;
 *   - return value 0 = success, negative = rtk_status; no C++ exception crosses the ABI;
 *   - dtype: RTK_F32, RTK_BF16 or RTK_F16 for the frame / q / k / v payloads (the scoring entry points take the
 *     mode-carrying codes as well); scores, distances and RoPE
 *     tables are always fp32; indices are int64 and masks are 1 byte per element (torch.bool layout);
 *   - strides are in ELEMENTS; the innermost (channel / head_dim) axis is always contiguous.
 *
 * Contents (56 entry points; search for the section title).  A binding of the hot path needs §A, §B, §E and the four
 * calls of §G; §D / §F are the stage-by-stage forms the same kernels are also reachable through.
 *   §A  version / errors                rtk_version, rtk_last_error, rtk_arch
 *   §B  "DPSelect"                      rtk_dpselect_dis, rtk_dpselect_select, rtk_gather_frames
 *   §C  "MA-LLM / MA-LLM-hard merges"   rtk_adjacent_cosine, rtk_mallm_argmax, rtk_mallm_merge, rtk_mallm_hard_chain
 *   §D  "RoPE tables"                   rtk_rope_merge, rtk_rope_table, rtk_rope_rotate_rows, rtk_rope_shift(_segments)
 *   §E  "PivotKV" (one unit, by stage)  rtk_pivotkv_score(_stages, _stages_masked, _passes_batched(_q), _partials),
 *                                       rtk_pivotkv_prepare, rtk_pivotkv_select(_batched), rtk_pivotkv_evict,
 *                                       rtk_pivotkv_commit, rtk_copy_rows, *_workspace_bytes
 *   §F  "Chunk-batched cache maintenance"  rtk_pivotkv_append, rtk_pivotkv_evict_batched(_rope), _commit_batched,
 *                                       _place_batched, rtk_pivotkv_compact_batched (+ _compact_sync_ints)
 *   §G  "One-call update and one-call flush"  rtk_pivotkv_update, rtk_pivotkv_flush, rtk_pivotkv_append_rope,
 *                                       rtk_position_shift (+ rtk_pivotkv_shift_ticket_ints): what the shipped
 *                                       package calls per update / per chunk / per decode step
 *   §H  "Direct peer-to-peer all-gather over xGMI"  rtk_p2p_alloc / _free / _export / _open / _close / _push / _wait
 *   §I  "Measurement support"           rtk_profile_* (off by default)
 */
#ifndef RETAKE_HIP_H
#define RETAKE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rtk_stream_t; /* hipStream_t */

enum rtk_dtype {
    RTK_F32 = 0,
    RTK_BF16 = 1,
    /* Scoring entry points only (rtk_pivotkv_score*, rtk_pivotkv_select_batched): bf16 payloads AND the reference's
     * bf16 rounding chain for the score (longvideo_cache.py:264-270 on a bf16 model rounds the logits, the
     * probabilities, the per-head column sums and both means to bf16).  RTK_BF16 keeps exact bf16 products with fp32
     * accumulation, softmax and sums - more accurate than the reference; this code reproduces the reference's own
     * quantised scores (head_dim 128 only).  Workspace / partial sizes differ: query them with the same code. */
    RTK_BF16_REFROUND = 2,
    /* Scoring entry points and rtk_pivotkv_prepare only: bf16 payloads, scored through the fp16 matrix instruction.  The
     * un-rotated q~ is stored as fp16(q~ * log2(e)/sqrt(D)) - ONE extra rounding, to 11 significant bits - and k~ is
     * re-encoded as fp16 exactly (a second copy inside the score workspace; the bf16 k~ the eviction re-rotates is
     * unchanged), so the accumulators are base-2 logits and both passes spend two instructions per logit instead of
     * three.  Scores move by ~1e-5 relative against RTK_BF16 (far inside the reference's own bf16 rounding of the
     * logits); everything downstream of the score is identical.  Opt in: score_rounding="fast".  head_dim 128 only;
     * |k~| beyond fp16's range (65504) saturates. */
    RTK_BF16_FAST = 3,
    /* fp16 payloads (a float16 model).  Like RTK_BF16, every entry point restates the reference's chain of torch ops
     * with one rounding to the tensor dtype per op; the score uses exact fp16 products with fp32 accumulation.
     * Wherever an argument is called `round_bf16`, 2 selects fp16 tables (1 = bf16, 0 = fp32). */
    RTK_F16 = 4,
    /* Scoring entry points only: fp16 payloads AND the reference's rounding chain on a float16 model - RTK_BF16_REFROUND's
     * chain with every rounding to fp16 instead of bf16 (logits, probabilities, per-head sums, both means); head_dim 128.
     * rtk_pivotkv_prepare takes RTK_F16 for such a batch. */
    RTK_F16_REFROUND = 5
};

/* Flag for the `dtype` argument of the SCORING entry points (rtk_pivotkv_score_workspace_bytes, _score_partials,
 * _score, _score_stages[_masked], _score_passes_batched, rtk_pivotkv_prepare, and score_dtype of rtk_pivotkv_select_batched):
 * the caller batches many (layer, chunk) units per launch, so the key / row splits - which fix the workspace layout, the
 * partial layout AND the reduction order, i.e. the bits of the score - are chosen for the length of a workgroup's stream
 * instead of for the workgroup count of one unit (one key split, about eight row tiles per row split).  Every call that
 * touches the same workspace must carry the same flag. */
#define RTK_SCORE_MANY_UNITS 0x100
/* Flag for the `dtype` argument of rtk_pivotkv_prepare only: the chunk keeps every token (compression_ratio 1, what
 * `dynamic_compression_ratio` gives every prompt within max_input_length, qwen2_vl.py:553-554) and is not scored, so
 * the un-rotated queries are not produced - k~ (for the re-rotation) and the cache tail still are. */
#define RTK_PREPARE_K_ONLY 0x200

enum rtk_status {
    RTK_OK = 0,
    RTK_EINVAL = -1,      /* bad argument (NULL pointer, non-positive size, unsupported dtype) -> ValueError   */
    RTK_EUNSUPPORTED = -2,/* shape outside what the kernels support                      -> NotImplementedError */
    RTK_EWORKSPACE = -3,  /* workspace too small                                          -> ValueError         */
    RTK_EHIP = -4,        /* a HIP runtime call failed (message in rtk_last_error)        -> RuntimeError       */
    RTK_EREFCRASH = -5    /* input on which the reference itself raises (N==1 async)      -> IndexError         */
};

/* ABI version (bumped on any signature change) and the thread-local message of the last failure. */
int rtk_version(void);
const char* rtk_last_error(void);
/* Name of the code-object architecture this library was built for ("gfx950"). */
const char* rtk_arch(void);

/* ---------------------------------------------------------------------------------------------
 * DPSelect  — replaces retake/visual_compression.py:100-175
 * ------------------------------------------------------------------------------------------- */

/* D1-D2  visual_compression.py:100-106  F.cosine_similarity(mb[:, :-1], mb[:, 1:], dim=-1),
 * `1 - sim.float()`, cat(ones).   x [T,N,C] (contiguous) -> dis [T,N] fp32, row 0 = 1.0.
 * RTK_BF16 reproduces the reference's bf16 rounding chain (norm, quotient, product, sum). */
int rtk_dpselect_dis(const void* x, int T, int N, int C, int dtype, float* dis, rtk_stream_t stream);

/* D3-D6  visual_compression.py:108-135 (sync) / :142-169 (async): patch mean, argrelmax peaks
 * (max_pool1d_with_indices window, first index wins), +2 bonus, topk(tgt) + ascending sort.
 *   dis   [T,N] fp32 (not modified)
 *   idx   sync: [tgt] int64;  async: [tgt,N] int64 (column n = patch n)
 *   mask  [tgt,N] bytes (keypatches_mask before .flatten())
 *   keys  workspace AND output: async [N,T] fp32 — the keys after the +2 bonus; sync [2,T]: row 0 the
 *         keys, row 1 the patch-mean distance dis.mean(1)
 * Ties at the k-th boundary are taken lowest index first (torch's order is backend-specific).
 * async with N == 1 returns RTK_EREFCRASH (the reference raises IndexError, :153-156). */
int rtk_dpselect_select(const float* dis, int T, int N, int tgt, int window, int sync,
                        int64_t* idx, uint8_t* mask, float* keys, rtk_stream_t stream);

/* D7  visual_compression.py:138 (sync `mb[:, peaks]`) / :173 (async `mb.gather(1, ...)`).
 * x [T,N,C] -> out [tgt,N,C]; idx as produced by rtk_dpselect_select. */
int rtk_gather_frames(const void* x, int T, int N, int C, int dtype, const int64_t* idx, int tgt, int sync,
                      void* out, rtk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * MA-LLM / MA-LLM-hard merges — replaces retake/visual_compression.py:5-47 and :50-83 (one merge step;
 * the caller loops until the target frame count is reached, qwen2_vl.py:402-410)
 * ------------------------------------------------------------------------------------------- */

/* :19 / :63  F.cosine_similarity(mb[:, :-1], mb[:, 1:], dim=-1): x [T,N,C] -> cos_out [T-1,N] fp32 holding
 * the value in the INPUT dtype (RTK_BF16: the reference's bf16 rounding chain). */
int rtk_adjacent_cosine(const void* x, int T, int N, int C, int dtype, float* cos_out, rtk_stream_t stream);

/* :20-24 / :64-67  idx[n] = first arg-max over t of cos[t,n]; sync != 0: arg-max of the patch-mean row
 * (rounded to bf16 when round_bf16 != 0), written to every n.  cos [T1,N] fp32, idx [N] int64. */
int rtk_mallm_argmax(const float* cos, int T1, int N, int sync, int round_bf16, int64_t* idx, rtk_stream_t stream);

/* :26-46 / :69-82  the merge: x [T,N,C], sizes [T,N] (dtype of x; soft merge only) -> out [T-1,N,C],
 * sizes_out [T-1,N].  hard == 0: out[t] = (x[d]*s[d]) / s[d] with d = t + (t > idx), and at t = idx the
 * size-weighted mean of frames idx and idx+1, one rounding per torch op; hard != 0: out[t] = x[t + (t >= idx)]
 * (sizes / sizes_out may be NULL). */
int rtk_mallm_merge(const void* x, const void* sizes, const int64_t* idx, int T, int N, int C, int dtype,
                    int hard, void* out, void* sizes_out, rtk_stream_t stream);

/* MA-LLM-hard run to its target length in one call (the reference loops the single step, qwen2_vl.py:406-408: one
 * full pass over the bank per dropped frame).  A hard merge only drops the first frame of the most similar pair, so
 * every other adjacent cosine of the next step is the one already computed: per step ONE new pair is scored (at every
 * patch position when sync) with rtk_adjacent_cosine's arithmetic, bit for bit.  x [T,N,C] -> the surviving frame
 * indices, ascending: async idx [tgt,N] int64 (column n = patch n), sync idx [tgt]; gather the rows with
 * rtk_gather_frames (sync flag alike).  cos_ws: [T-1,N] fp32 scratch.  RTK_EUNSUPPORTED for rows that are not 16-byte
 * vectors or T beyond the LDS-resident frame list (~12 000): loop the single step. */
int rtk_mallm_hard_chain(const void* x, int T, int N, int C, int dtype, int tgt, int sync, float* cos_ws,
                         int64_t* idx_out, rtk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * RoPE tables — replaces longvideo_cache.py:68-74 (M-RoPE section merge) and, optionally, the
 * rotary_emb_fn(...) calls at :249 and :298 when the rotary module is the standard
 * inv_freq/attention_scaling kind.
 * ------------------------------------------------------------------------------------------- */

/* Merge the [P,L,D] cos and sin returned by rotary_emb_fn (P = 3 for M-RoPE, 1 otherwise; dtype of
 * q) into fp32 [L,D] tables.  sections = mrope_section (host pointer, nsec ints; NULL for P = 1);
 * channel block i of `mrope_section*2` takes row i % 3. */
int rtk_rope_merge(const void* cos_in, const void* sin_in, int P, int L, int D, int dtype,
                   const int* sections_host, int nsec, float* cos_out, float* sin_out, rtk_stream_t stream);

/* Build the merged fp32 [L,D] tables directly from position ids: cos(pos*inv_freq)*scaling.
 * pos [P,L] int64 (row p at pos + p*pos_stride, pos_stride >= L), inv_freq [D/2] fp32.  round_bf16 != 0
 * rounds the table entries to bf16 (what a bf16 model's rotary module returns). */
int rtk_rope_table(const int64_t* pos, int64_t pos_stride, int P, int L, const float* inv_freq, int D,
                   float attention_scaling,
                   const int* sections_host, int nsec, int round_bf16, float* cos_out, float* sin_out,
                   rtk_stream_t stream);

/* Forward rotation in place of UN-rotated key rows at given ids (longvideo_cache.py:297-306 for rows whose re-rotation
 * was deferred, see rtk_pivotkv_batch.defer_rot): k [layers, H, rows, D] (element strides stride_layer, stride_h; rows
 * contiguous), ids of layer l at pos + l*pos_stride_layer, row p of them at + p*pos_stride_p (P = 1 or 3).
 * k' = (k*cos) + (rotate_half(k)*sin), the tables and roundings of rtk_pivotkv_evict_batched_rope, bit for bit. */
int rtk_rope_rotate_rows(void* k, int64_t stride_layer, int64_t stride_h, int layers, int H, int rows, int D, int dtype,
                         const int64_t* pos, int64_t pos_stride_layer, int64_t pos_stride_p, int P, const float* inv_freq,
                         float attention_scaling, const int* sections_host, int nsec, int round_mode, rtk_stream_t stream);

/* In-place k <- R(delta) k on the temporal channels (position row 0; every channel when P = 1) of keys
 * that are already rotated: k [H, n, D] with head stride `stride_h`, delta = one int64 in device
 * memory.  Used by the multi-GPU sharding (retake/sharded.py): ranks compress their chunks at
 * provisional temporal ids; once the global offsets are known, R(p + delta) = R(delta) R(p)
 * (longvideo_cache.py:80-81 composed with the continuity shift of qwen2_vl.py:68-73). */
int rtk_rope_shift(void* k, int64_t stride_h, int H, int n, int D, int dtype, const int64_t* delta_dev,
                   const float* inv_freq, int P, const int* sections_host, int nsec, rtk_stream_t stream);

/* The same for a whole assembled cache in one launch: k [layers, H, world * seg, D] (element strides stride_layer,
 * stride_h; rows contiguous), the segment of rank r in layer l rotated by table_dev[r * layers + l] temporal steps
 * (table_dev: device int64 [world, layers], what retake/sharded.exchange_temporal_offsets returns).  Replaces
 * layers x world calls of rtk_rope_shift in ShardedPivotKV.finalize. */
int rtk_rope_shift_segments(void* k, int64_t stride_layer, int64_t stride_h, int layers, int H, int world, int seg, int D,
                            int dtype, const int64_t* table_dev, const float* inv_freq, int P,
                            const int* sections_host, int nsec, rtk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PivotKV — replaces retake/longvideo_cache.py:248-318
 * ------------------------------------------------------------------------------------------- */

/* Workspace (bytes) needed by rtk_pivotkv_score for these sizes. */
size_t rtk_pivotkv_score_workspace_bytes(int Hq, int Hkv, int L, int D, int dtype);

/* P2-P5  longvideo_cache.py:248-270.
 *   q [Hq,L,D], k [Hkv,L,D]: element (h,l,d) at h*stride_h + l*stride_l + d  (post-RoPE, as the
 *   attention patch hands them over).
 *   cos/sin: fp32 [L,D] merged tables or NULL.  Non-NULL = pos_embed_reforge: q and k are first
 *   un-rotated, x~ = ((x*cos) - (rotate_half(x)*sin)) / attention_scaling^2   (:76-78, :109-111).
 *   score [L] fp32 = mean_g mean_{h in g} sum_i softmax_j(q~_h,i . k~_g,j / sqrt(D))  — keys are the
 *   current chunk only, no causal mask (:264-270).
 *   k_unrot (optional, may be NULL): receives k~ [Hkv,L,D] contiguous in `dtype` for rtk_pivotkv_evict.
 * The [Hq,L,L] probability tensor is never materialised. */
int rtk_pivotkv_score(const void* q, int64_t q_stride_h, int64_t q_stride_l,
                      const void* k, int64_t k_stride_h, int64_t k_stride_l,
                      int Hq, int Hkv, int L, int D, int dtype,
                      const float* cos, const float* sin, float attention_scaling,
                      float* score, void* k_unrot, void* workspace, size_t workspace_bytes,
                      rtk_stream_t stream);

/* The same computation cut into its three stream-ordered stages, so that a caller can put the two big
 * matrix kernels of consecutive updates back to back on one stream while the small pre / post kernels of the
 * neighbouring updates run on other streams (PivotKVCache pipeline_streams): same arguments, same workspace.
 *   RTK_SCORE_PREPARE   un-rotate + pack q~ (workspace) and k~ (k_unrot or workspace)
 *   RTK_SCORE_PASSES    pass 1, lse combine, pass 2 (reads q~/k~, writes the column partials)
 *   RTK_SCORE_FINALIZE  fixed-order reduction of the partials -> score */
enum rtk_score_stage { RTK_SCORE_PREPARE = 1, RTK_SCORE_PASSES = 2, RTK_SCORE_FINALIZE = 4 };
int rtk_pivotkv_score_stages(const void* q, int64_t q_stride_h, int64_t q_stride_l,
                             const void* k, int64_t k_stride_h, int64_t k_stride_l,
                             int Hq, int Hkv, int L, int D, int dtype,
                             const float* cos, const float* sin, float attention_scaling,
                             float* score, void* k_unrot, void* workspace, size_t workspace_bytes,
                             int stages, float* partial_out /* NULL = inside workspace */, rtk_stream_t stream);

/* The same with the chunk's key-patch mask ([L] bytes, may be NULL) and int32 scratch of L + 1 entries: pass 2 then
 * computes the column partials of the UNMASKED tokens only (the caller's selection overwrites the masked tokens' score
 * with 1.0, longvideo_cache.py:272-274) and the masked entries of partial / score hold no meaning until that override.
 * head_dim 128 kernels; other shapes ignore the mask. */
int rtk_pivotkv_score_stages_masked(const void* q, int64_t q_stride_h, int64_t q_stride_l,
                                    const void* k, int64_t k_stride_h, int64_t k_stride_l,
                                    int Hq, int Hkv, int L, int D, int dtype,
                                    const float* cos, const float* sin, float attention_scaling,
                                    float* score, void* k_unrot, void* workspace, size_t workspace_bytes,
                                    int stages, float* partial_out, const void* key_mask, int32_t* key_index_ws,
                                    rtk_stream_t stream);

/* Fused form of {rtk_rope_table, RTK_SCORE_PREPARE, rtk_pivotkv_append} for the standard inv_freq rotary
 * modules (longvideo_cache.py:238, :248-259): one pass over the chunk's q, k, v that builds each token's
 * cos/sin in registers (same arithmetic and rounding as rtk_rope_table), writes q~ into `workspace` (where
 * rtk_pivotkv_score_stages(RTK_SCORE_PASSES) expects it) and k~ into k_unrot, and appends k and v to the cache
 * tail.  Needs 16-byte aligned pointers / strides; returns RTK_EUNSUPPORTED otherwise (use the three calls). */
int rtk_pivotkv_prepare(const void* q, int64_t q_stride_h, int64_t q_stride_l,
                        const void* k, int64_t k_stride_h, int64_t k_stride_l,
                        const void* v, int64_t v_stride_h, int64_t v_stride_l,
                        int Hq, int Hkv, int L, int D, int dtype,
                        const int64_t* pos, int64_t pos_stride, int P, const float* inv_freq, float attention_scaling,
                        const int* sections_host, int nsec, int round_bf16,
                        void* k_unrot, void* workspace, size_t workspace_bytes,
                        void* k_tail, void* v_tail, int64_t tail_stride_h,
                        int64_t* pos_copy /* optional [P, L]: private copy of the ids for a deferred selection */,
                        rtk_stream_t stream);

/* RTK_SCORE_PASSES for n_units (layer, chunk) units in ONE launch per kernel (blockIdx.y = unit): the units'
 * score workspaces (holding q~ from RTK_SCORE_PREPARE / rtk_pivotkv_prepare), their k~ buffers (k_unrot0 == NULL:
 * inside the workspaces) and their partial outputs lie workspace_stride / k_unrot_stride bytes and
 * partial_stride_floats floats apart.  bf16, head_dim 128 (RTK_EUNSUPPORTED otherwise: run the stage per unit).
 * What PivotKVCache runs from after_forward for all layers of a chunk: 28x larger grids, no per-layer tails.
 * Live keys (optional; both NULL = every column): key_masks_host is a HOST array of n_units device pointers to the units'
 * key-patch masks ([L] bytes, entries may be NULL); key_index_ws is device scratch of n_units * (L + 1) int32.  The
 * reference overwrites the score of every masked token with 1.0 after the scoring (longvideo_cache.py:272-274), so
 * pass 2 - one column mass per key - runs on the unit's unmasked keys only and leaves the masked columns of `partial`
 * UNWRITTEN (the selection's mask override never reads them); every other column gets the bits it always got.  The
 * same mask must be handed to rtk_pivotkv_select_batched. */
int rtk_pivotkv_score_passes_batched(void* workspace0, size_t workspace_stride, void* k_unrot0, size_t k_unrot_stride,
                                     float* partial0, size_t partial_stride_floats, int n_units,
                                     int Hq, int Hkv, int L, int D, int dtype,
                                     const void* const* key_masks_host, int32_t* key_index_ws, rtk_stream_t stream);

/* The same with the units' queries read in place: q_units_host is a HOST array of n_units device pointers to tensors the
 * caller keeps alive until the launches have run, element (h, i, d) of unit u at q_units_host[u] + h*q_stride_h +
 * i*q_stride_l + d (elements; 16-byte aligned rows).  For units whose score operands ARE the caller's tensors - the
 * pre-RoPE projections of the attention prologue - no packed copy of the queries is made at all.  NULL: identical to
 * rtk_pivotkv_score_passes_batched.  16-bit payloads, head_dim 128. */
int rtk_pivotkv_score_passes_batched_q(void* workspace0, size_t workspace_stride, void* k_unrot0, size_t k_unrot_stride,
                                       float* partial0, size_t partial_stride_floats, int n_units,
                                       int Hq, int Hkv, int L, int D, int dtype,
                                       const void* const* key_masks_host, int32_t* key_index_ws,
                                       const void* const* q_units_host, int64_t q_stride_h, int64_t q_stride_l,
                                       rtk_stream_t stream);

/* P6-P7, P9-P10  longvideo_cache.py:272-277, :283-295.
 *   score [L] fp32: entries with mask != 0 are overwritten with 1.0 IN PLACE (masked_fill_, :274);
 *   mask may be NULL.
 *   keep_idx [keep] int64 ascending = topk(keep).sort()  (ties: lowest index first)
 *   rank [L] int32: position of token l in keep_idx, or -1 if evicted (inverse map, for callers that
 *        scan tokens in order); may be NULL when a workspace is given
 *   pos [P,L] int64 (P = 3 M-RoPE rows t,h,w or 1), may be NULL together with pos_out;
 *   pos_out [P,keep] int64 (row p at pos_out + p*pos_out_stride, pos_out_stride >= keep) = gathered ids;
 *       if reforge != 0 row 0 becomes
 *       tmin + (int64)((float)(t - tmin) * (float)(keep / (double)L))        (:293-295)
 *   workspace: rtk_pivotkv_select_workspace_bytes(L) bytes of device scratch.  With it the selection runs
 *       chip-wide (rank-by-counting over ~L/32 workgroups + an ordered emit); without it (NULL) on one
 *       workgroup (radix select).  Same exact result either way. */
size_t rtk_pivotkv_select_workspace_bytes(int L);
int rtk_pivotkv_select(float* score, const uint8_t* mask, int L, int keep,
                       const int64_t* pos, int P, int reforge,
                       int64_t* keep_idx, int32_t* rank, int64_t* pos_out, int64_t pos_out_stride,
                       void* workspace, size_t workspace_bytes, rtk_stream_t stream);

/* The selection of n (layer, chunk) units in the same three launches (finalize of the column partials ->
 * rank -> emit): what PivotKVCache runs from after_forward for all layers of a chunk.  Units share L, keep, P
 * and the partial layout (Hkv, RS, G as reported by rtk_pivotkv_score_partials).  RTK_EUNSUPPORTED for L < 512
 * (call rtk_pivotkv_select per unit).  Fewer than 8 units are ranked chip-wide one after the other; 8 or more
 * (L <= 16384) get one radix-select workgroup each (keys in LDS), all side by side - same result, bit for bit.
 * `units` is a HOST array. */
typedef struct rtk_select_unit {
    const float* partial;   /* column partials [Hkv, RS, L] written by RTK_SCORE_PASSES (partial_out), or NULL: score is final */
    float* score;           /* [L] fp32 (written from the partials, masked entries overwritten with 1.0) */
    const uint8_t* mask;    /* [L] key-patch mask or NULL */
    const int64_t* pos;     /* [P, L] position ids (row stride L) or NULL */
    int64_t* keep_idx;      /* [keep] */
    int32_t* rank;          /* [L] or NULL */
    int64_t* pos_out;       /* rows at pos_out + p*pos_out_stride, or NULL */
    void* workspace;        /* rtk_pivotkv_select_workspace_bytes(L), 256-byte aligned */
} rtk_select_unit;
#define RTK_SELECT_MAX_UNITS 28
int rtk_pivotkv_select_batched(const rtk_select_unit* units, int n_units, int Hkv, int RS, int G, int L,
                               int keep, int P, int reforge, int64_t pos_out_stride,
                               int score_dtype /* RTK_BF16_REFROUND / RTK_F16_REFROUND: partials are per head
                                                  [Hkv*G, RS, L] and the finalize applies the reference's roundings; else 0 */,
                               rtk_stream_t stream);

/* Layout of the column partials rtk_pivotkv_score_stages(RTK_SCORE_PASSES) produces for these sizes: returns the
 * number of floats (Hkv * RS * L) and writes RS (row splits actually used) to *rs_out. */
size_t rtk_pivotkv_score_partials(int Hq, int Hkv, int L, int D, int dtype, int* rs_out);

/* P1, P8, P11, P13  longvideo_cache.py:238, :278-280, :297-306, :313-318 — the eviction scan.
 * One launch over the chunk's K and V rows:
 *   every row l is appended to the cache tail   k_tail/v_tail[h][l]      (the uncompressed view the
 *                                                current layer's attention reads, :238)
 *   row keep_idx[r] is also written to          k_kept/v_kept[h][r]      (the compacted cache, :313-318)
 *   keep_idx [keep] int64 ascending, as produced by rtk_pivotkv_select
 * With reforge (cos_new != NULL) the kept K row is taken from k_unrot (un-rotated, as produced by
 * rtk_pivotkv_score) and rotated forward with the fp32 [keep,D] tables of its NEW position:
 *   k' = (k~*cos_new) + (rotate_half(k~)*sin_new)          (:80-81, :113-114)
 * otherwise it is a plain copy of k.  V rows are plain copies.
 * k_tail/v_tail may be NULL (no append wanted).  Destinations: element (h,r,d) at
 * h*stride_h + r*D + d. */
int rtk_pivotkv_evict(const void* k, int64_t k_stride_h, int64_t k_stride_l,
                      const void* v, int64_t v_stride_h, int64_t v_stride_l,
                      const void* k_unrot, int Hkv, int L, int D, int dtype,
                      const int64_t* keep_idx, int keep,
                      const float* cos_new, const float* sin_new,
                      void* k_tail, void* v_tail, int64_t tail_stride_h,
                      void* k_kept, void* v_kept, int64_t kept_stride_h,
                      rtk_stream_t stream);

/* P13 (second half)  longvideo_cache.py:313-318: commits the staged kept K and V rows of one layer over
 * the head of the uncompressed tail, once the layer's attention has consumed that view.  One launch:
 * dst[h][r][:] = stage[h][r][:], r < rows, for both tensors. */
int rtk_pivotkv_commit(const void* k_stage, const void* v_stage, int64_t stage_stride_h,
                       void* k_dst, void* v_dst, int64_t dst_stride_h,
                       int H, int rows, int D, int dtype, rtk_stream_t stream);

/* Row-block copy used to commit staged kept rows into the cache after the layer's attention has
 * consumed the uncompressed view: dst[h][r][:] = src[h][r][:], r < rows. */
int rtk_copy_rows(const void* src, int64_t src_stride_h, void* dst, int64_t dst_stride_h,
                  int H, int rows, int D, int dtype, rtk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Chunk-batched cache maintenance.  The reference rebuilds every layer's cache with two torch.cat per
 * (layer, chunk) (longvideo_cache.py:238, :313-318) and calls cache.after_forward() once per video
 * chunk (qwen2_vl.py:715-716).  Here `update` appends the chunk to the pre-allocated cache tail and
 * records which rows survive; the gather / re-rotation / compaction of every layer of the chunk runs
 * as ONE launch each from after_forward (or from whatever touches the cache first).
 * ------------------------------------------------------------------------------------------- */

/* P1  longvideo_cache.py:238 (DynamicCache.update -> torch.cat): K and V rows of the chunk
 * (element (h,l,d) at h*stride_h + l*stride_l + d) -> k_tail/v_tail (element (h,l,d) at
 * h*tail_stride_h + l*D + d), the uncompressed view the current layer's attention reads. */
int rtk_pivotkv_append(const void* k, int64_t k_stride_h, int64_t k_stride_l,
                       const void* v, int64_t v_stride_h, int64_t v_stride_l,
                       int Hkv, int L, int D, int dtype,
                       void* k_tail, void* v_tail, int64_t tail_stride_h, rtk_stream_t stream);

/* One (layer, chunk) unit of a batched eviction.  Strides in elements. */
typedef struct rtk_evict_unit {
    const void* k_src;        /* K rows of the chunk, element (h,l,d) at h*k_src_stride_h + l*D + d.  With cos_new:
                                 the UN-rotated k~ written by rtk_pivotkv_score; without: the rotated rows */
    int64_t k_src_stride_h;
    const void* v_src;        /* V rows of the chunk, element (h,l,d) at h*v_src_stride_h + l*D + d */
    int64_t v_src_stride_h;
    const int64_t* keep_idx;  /* [keep] ascending, from rtk_pivotkv_select */
    const float* cos_new;     /* fp32 [keep,D] tables of the NEW positions of the kept rows, or NULL (no reforge) */
    const float* sin_new;
    void* k_dst;              /* kept row (h,r) at h*k_dst_stride_h + r*D; must not alias k_src.  NULL: K is left alone */
    int64_t k_dst_stride_h;
    void* v_dst;              /* kept row (h,r) at h*v_dst_stride_h + r*D; must not alias v_src.  NULL: V is left alone
                                 (keep == L: every row already sits where the append put it) */
    int64_t v_dst_stride_h;
    const int64_t* pos_src;   /* optional: ids of the kept rows [P,keep] (row stride pos_src_stride) ... */
    int64_t pos_src_stride;
    int64_t* pos_dst;         /* ... copied to the layer's position cache tail (row stride pos_dst_stride) */
    int64_t pos_dst_stride;
} rtk_evict_unit;
#define RTK_EVICT_MAX_UNITS 28   /* units per launch (the array travels as a kernel argument); more = more launches */

/* P8, P9, P11, P12  longvideo_cache.py:278-288, :297-310 for n_units (layer, chunk) units that share
 * Hkv, D, keep, P and dtype:  K: k' = (k~*cos_new) + (rotate_half(k~)*sin_new) (or a copy), V: copy,
 * position ids: copy.  `units` is a HOST array. */
int rtk_pivotkv_evict_batched(const rtk_evict_unit* units, int n_units, int Hkv, int D, int keep, int P,
                              int dtype, int stage_low_only, rtk_stream_t stream);
/* stage_low_only bit 1 (value 3 together with bit 0): K is copied verbatim from a buffer of its own (k_src is not the
 * tail k_dst lies in - the un-rotated rows of a deferred re-rotation): EVERY kept K row is written to k_dst.
 * stage_low_only != 0 (what PivotKVCache uses, together with rtk_pivotkv_place_batched): rows that are copied verbatim
 * (V; K when there are no tables) are written to k_dst / v_dst ONLY when keep_idx[r] < keep, i.e. when their source
 * lies inside the destination range of the compaction and has to be parked; the other rows are moved in place by
 * rtk_pivotkv_place_batched.  Re-rotated K rows are always written. */

/* The same with the tables of the NEW ids computed in the kernel (reference :297-298: rotary_emb(compressed ids) and
 * the M-RoPE section merge :68-74, i.e. rtk_rope_table's arithmetic: fp32 sincosf(id * inv_freq[d mod D/2]) *
 * attention_scaling, rounded to bf16 when round_bf16): every unit needs pos_src (the kept rows' new ids, [P, keep])
 * and cos_new = sin_new = NULL.  P = 1 (plain RoPE) or 3 (M-RoPE with `sections`).  Saves the table launch and
 * 2 x keep x D x 4 bytes of write + read per unit. */
int rtk_pivotkv_evict_batched_rope(const rtk_evict_unit* units, int n_units, int Hkv, int D, int keep, int P,
                                   int dtype, const float* inv_freq, float attention_scaling,
                                   const int* sections_host, int nsec, int round_bf16, int stage_low_only,
                                   rtk_stream_t stream);

typedef struct rtk_copy_unit {
    const void* src;
    int64_t src_stride_h_bytes;
    void* dst;
    int64_t dst_stride_h_bytes;
} rtk_copy_unit;
#define RTK_COPY_MAX_UNITS 64

/* P13  longvideo_cache.py:313-318 for n_units staged row blocks: dst[h][r][:] = src[h][r][:], r < rows.
 * `units` is a HOST array. */
int rtk_pivotkv_commit_batched(const rtk_copy_unit* units, int n_units, int H, int rows, int D, int dtype,
                               rtk_stream_t stream);

/* P13 without a full staging copy: for every unit, kept row r (r < keep) of the tail block `tail` (element (h, l, d) at
 * h*tail_stride_h_bytes + l*D*es) becomes row r of the same block.  Its source is chunk row keep_idx[r]: read in place
 * when keep_idx[r] >= keep (outside the destination range), from stage[h][r] otherwise (parked there by
 * rtk_pivotkv_evict_batched with stage_low_only).  Only ~ratio of the kept rows take the staging hop.  One unit per
 * tensor (V of a layer; K of a layer when it is not re-rotated).  `units` is a HOST array. */
typedef struct rtk_place_unit {
    const void* stage;             /* [H, keep, D] staged rows (only rows with keep_idx[r] < keep are read) */
    int64_t stage_stride_h_bytes;
    void* tail;                    /* the uncompressed chunk inside the cache: rows [0, L) per head */
    int64_t tail_stride_h_bytes;
    const int64_t* keep_idx;       /* [keep] ascending */
} rtk_place_unit;
#define RTK_PLACE_MAX_UNITS 64
int rtk_pivotkv_place_batched(const rtk_place_unit* units, int n_units, int H, int keep, int D, int dtype,
                              rtk_stream_t stream);

/* P8-P13 in ONE launch, in place (ABI 14): longvideo_cache.py:278-288, :297-310, :313-318 for n_units (layer, chunk)
 * units that share Hkv, D, keep, P and dtype - what rtk_pivotkv_evict_batched[_rope] + rtk_pivotkv_place_batched do in
 * two launches with a staging hop.  Kept row r of a unit's V tail (and K tail, k_mode RTK_COMPACT_K_INPLACE) comes from
 * chunk row keep_idx[r] >= r of the SAME tail: the workgroups of a unit run in ticket order and store a row block only
 * after every lower block has its source rows in registers, so nothing is staged and the kernel moves
 * 4 x keep x Hkv x D x es bytes per unit.  Same arithmetic, same bits as rtk_pivotkv_evict_batched_rope. */
typedef struct rtk_compact_unit {
    const void* k_src;        /* RTK_COMPACT_K_ROTATE / _COPY: K rows of the chunk in a buffer of their own (the un-rotated
                                 k~), element (h,l,d) at h*k_src_stride_h + l*D + d.  _INPLACE: ignored */
    int64_t k_src_stride_h;
    void* k_tail;             /* the chunk inside the layer's cache: chunk row l of head h at h*k_tail_stride_h + l*D;
                                 kept row r is written to row r of the same block */
    int64_t k_tail_stride_h;
    void* v_tail;
    int64_t v_tail_stride_h;
    const int64_t* keep_idx;  /* [keep] ascending, from rtk_pivotkv_select[_batched] */
    const int64_t* pos_src;   /* ids of the kept rows [P, keep] (row stride pos_src_stride): the NEW ids the rotation
                                 uses (RTK_COMPACT_K_ROTATE: required) ... */
    int64_t pos_src_stride;
    int64_t* pos_dst;         /* ... copied to the layer's position cache tail (row stride pos_dst_stride), or NULL */
    int64_t pos_dst_stride;
} rtk_compact_unit;
#define RTK_COMPACT_MAX_UNITS 28   /* units per launch (the array travels as a kernel argument) */
enum rtk_compact_k_mode {
    RTK_COMPACT_K_ROTATE = 0,   /* kept K = k_src row re-rotated at its new ids (pos_embed_reforge, :297-306) */
    RTK_COMPACT_K_COPY = 1,     /* kept K = k_src row verbatim (deferred re-rotation: rtk_pivotkv_batch.defer_rot) */
    RTK_COMPACT_K_INPLACE = 2   /* kept K = row keep_idx[r] of k_tail, compacted in place like V (no reforge, :279) */
};
/* ints of the sync workspace for this geometry: device memory, zero-initialised once by the caller (tickets and flags of
 * the workgroups live in it; every launch that completes leaves it zeroed again, and a call that fails at a launch
 * re-zeroes it on `stream` before returning).  A kernel that aborts on the device (its bounded flag wait traps instead
 * of hanging) leaves the HIP context in error: the workspace - like every other buffer - is then to be re-created. */
size_t rtk_pivotkv_compact_sync_ints(int n_units, int Hkv, int keep, int D, int dtype);
/* `units` is a HOST array.  inv_freq / attention_scaling / sections / round_mode as rtk_pivotkv_evict_batched_rope
 * (RTK_COMPACT_K_ROTATE only).  One launch at a time per workspace: calls that share a workspace must be ordered on
 * ONE stream (PivotKVCache flushes a batch on the caller's current stream only). */
int rtk_pivotkv_compact_batched(const rtk_compact_unit* units, int n_units, int Hkv, int D, int keep, int P, int dtype,
                                int k_mode, const float* inv_freq, float attention_scaling, const int* sections_host,
                                int nsec, int round_mode, int32_t* sync_ws, size_t sync_ws_ints, rtk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * One-call update and one-call flush (ABI 13).  PivotKVCache.update runs 1,792 times per 2048-frame video and
 * the reference calls cache.after_forward() once per chunk (qwen2_vl.py:715-716): on MI355X the per-call HOST work
 * around the launches decides the real-geometry step, so the argument blocks are bound ONCE per chunk geometry
 * (rtk_pivotkv_batch, HOST memory owned by the caller) and per layer (rtk_layer_state, HOST memory owned by the
 * caller, updated by the library), and each update / flush is ONE call that fills in the launches the
 * entry points above would have been handed one by one.  Nothing new is computed here:
 *   rtk_pivotkv_update  = rtk_pivotkv_prepare (rotated q, k)  or the attention prologue (pre-RoPE q, k; below)
 *   rtk_pivotkv_flush   = rtk_pivotkv_score_passes_batched + rtk_pivotkv_select_batched +
 *                         rtk_pivotkv_compact_batched (or, without batch.compact_sync, rtk_pivotkv_evict_batched[_rope] +
 *                         rtk_pivotkv_place_batched) for the pending layers
 * ------------------------------------------------------------------------------------------- */

/* One layer's pre-allocated cache (longvideo_cache.py: key_cache[l] / value_cache[l] / position_cache[l]). */
typedef struct rtk_layer_state {
    void* k;                 /* [1, Hkv, cap, D] keys   (element (h, r, d) at (h*cap + r)*D + d) */
    void* v;                 /* [1, Hkv, cap, D] values */
    int64_t cap;             /* rows allocated per head */
    int64_t length;          /* committed (compressed) rows                                  - advanced by rtk_pivotkv_flush */
    int64_t pending;         /* uncompressed chunk rows at [length, length + pending)        - set by update, cleared by flush */
    int64_t pending_keep;    /* rows of them that survive                                                                      */
    int64_t* pos;            /* [P, pos_cap] position ids of the cached rows (pos_embed_reforge) or NULL */
    int64_t pos_cap;
    int64_t pos_len;         /*                                                              - advanced by rtk_pivotkv_flush */
    const uint8_t* mask;     /* the pending chunk's key-patch mask [L] or NULL (cache.keypatches_mask_chunk, :272-274) */
} rtk_layer_state;

/* The per-chunk batch: geometry + the slot-strided scratch of all layers (slot = layer index).  Every buffer is
 * caller-allocated device memory; sizes as the per-stage entry points document them. */
typedef struct rtk_pivotkv_batch {
    int32_t Hq, Hkv, L, D, keep, P, slots;
    int32_t dtype;            /* payload: RTK_F32 / RTK_BF16 / RTK_F16 */
    int32_t score_dtype;      /* what the scoring entry points are told (may carry RTK_SCORE_MANY_UNITS) */
    int32_t prep_dtype;       /* what rtk_pivotkv_prepare is told */
    int32_t reforge;          /* pos_embed_reforge */
    int32_t keep_all;         /* keep == L and no scores wanted: selection is the identity, no score passes */
    int32_t round_mode;       /* 0 fp32 / 1 bf16 / 2 fp16 rotary tables */
    int32_t nsec;
    int32_t sections[8];      /* mrope_section (nsec entries; nsec = 0: plain RoPE) */
    int32_t rs_n;             /* row splits of the column partials (rtk_pivotkv_score_partials) */
    int32_t skip_masked;      /* pass 2 on the unmasked keys only */
    float attention_scaling;
    int32_t defer_rot;        /* pos_embed_reforge with the re-rotation of the kept keys deferred: the flush copies the
                                 UN-rotated kept rows into the cache and their ids into the position cache; the owner
                                 rotates them once, at their final ids, with rtk_rope_rotate_rows (chunk-sharded prefill:
                                 a block's temporal offset is known only when the blocks before it are done) */
    const float* inv_freq;    /* [D/2] fp32 */
    void* score_ws;           /* slot s at score_ws + s*score_ws_stride (256-byte aligned): q~, lse, ... */
    uint64_t score_ws_stride, score_ws_bytes;
    void* k_unrot;            /* [slots, Hkv, L, D] un-rotated keys */
    float* partials;          /* [slots, partial_floats] */
    uint64_t partial_floats;
    float* score;             /* [slots, L] */
    int64_t* pos_old;         /* [slots, P, L] ids of the pending chunks (after the continuity shift) */
    int64_t* keep_idx;        /* [slots, keep] */
    int64_t* pos_new;         /* [P, slots, keep] */
    void* sel_ws;             /* slot s at sel_ws + s*sel_ws_stride */
    uint64_t sel_ws_stride;
    int32_t* key_index;       /* [slots, L + 1] */
    void* v_stage;            /* [slots, Hkv, keep, D] */
    void* k_stage;            /* [slots, Hkv, keep, D] (no reforge only, else NULL) */
    int64_t* shift_row;       /* RTK_UPDATE_PRE_ROPE: the caller's temporal-id row; rtk_pivotkv_flush applies the last
                                 pending layer's continuity shift to it in place (qwen2_vl.py:73), then clears this field */
    const void** q_units;     /* HOST array [slots] or NULL.  Non-NULL entries of the pending slots: the unit's queries are
                                 scored where they are (RTK_UPDATE_Q_IN_PLACE; element strides below), all pending units alike */
    int64_t q_stride_h, q_stride_l;
    int32_t pre_rope;         /* the pending units were appended from pre-RoPE projections: 1 k~ == k0 (pre-RoPE operands),
                                 2 k~ = the reference's un-rotation of the rotated row (RTK_UPDATE_ROUNDTRIP); 0 otherwise */
    int32_t batched_passes;   /* 1: the score passes of all pending layers run from rtk_pivotkv_flush, one launch per kernel
                                 (16-bit payloads, head_dim 128); 0: rtk_pivotkv_update runs them per unit */
    int32_t* compact_sync;    /* zero-initialised sync workspace of rtk_pivotkv_compact_batched for (slots, Hkv, keep, D,
                                 dtype), or NULL: the flush then stages through v_stage / k_stage (two launches) */
    uint64_t compact_sync_ints;
} rtk_pivotkv_batch;

/* The tensors of one update call.  Strides in elements; element (h, l, d) at h*stride_h + l*stride_l + d. */
typedef struct rtk_update_io {
    const void* q; int64_t q_stride_h, q_stride_l;   /* [Hq, L, D] */
    const void* k; int64_t k_stride_h, k_stride_l;   /* [Hkv, L, D] */
    const void* v; int64_t v_stride_h, v_stride_l;   /* [Hkv, L, D] */
    const int64_t* pos; int64_t pos_stride;          /* [P, L] ids of the chunk, row p at pos + p*pos_stride */
    void* q_rot; int64_t qr_stride_h, qr_stride_l;   /* RTK_UPDATE_PRE_ROPE: rotated queries out (may alias q) */
    int32_t flags;
    int32_t pad0;
    const int64_t* next_prev;                        /* RTK_UPDATE_SHIFT_NEXT (ABI 16): the NEXT layer's last cached temporal id */
    int32_t* ticket; int64_t ticket_ints;            /*   (device; NULL = -1); rtk_pivotkv_shift_ticket_ints(L, D) device words, */
                                                     /*   zeroed ONCE by the caller: word 0 launch count, word 31 the run-out   */
                                                     /*   LATCH (below), then the arrival counters, a cache line each           */
    int32_t* status;                                 /* (ABI 17) optional HOST-VISIBLE word (pinned host memory the device can   */
} rtk_update_io;                                     /*   write), zeroed by the caller: incremented when a launch's bounded     */
                                                     /*   wait runs out, so the host can see the latch without a device sync    */
size_t rtk_pivotkv_shift_ticket_ints(int L, int D);
enum rtk_update_flags {
    /* q, k are the PRE-RoPE projections (what q_proj / k_proj return).  One launch then does the whole prologue of the
     * attention patch (qwen2_vl.py:55-86, llava_onevision.py:59-141) and of PivotKVCache.update (:238, :248-259):
     *   continuity shift  t' = t + (prev + 1 - t[0]) on the temporal row, prev = the layer's last cached temporal id
     *                     (qwen2_vl.py:68-73; applied to the ids the kernel uses and to pos_old - the caller's tensor is
     *                     shifted in place by rtk_pivotkv_flush, see rtk_pivotkv_batch.shift_row);
     *   rotary tables     cos / sin of the shifted ids (rtk_rope_table's arithmetic, M-RoPE section merge :68-74);
     *   q_rot             = (q*cos) + (rotate_half(q)*sin), one rounding per torch op  -> the layer's attention;
     *   k tail, v tail    rotated k and v appended to the cache (:238);
     *   q~, k~            the un-rotated operands of the score passes are the inputs themselves (un-rotating
     *                     a rotation returns the pre-RoPE value up to rounding), copied to the score workspace / k_unrot.
     * Needs pos_embed_reforge and an inv_freq rotary (batch.inv_freq). */
    RTK_UPDATE_PRE_ROPE = 1,
    /* with RTK_UPDATE_PRE_ROPE and q_rot != q: no copy of the queries is made - the score passes of the flush read
     * io->q itself (the caller keeps it alive and unmodified until rtk_pivotkv_flush has run; the library records the
     * pointer and strides in batch->q_units).  16-bit payloads scored by the batched passes only. */
    RTK_UPDATE_Q_IN_PLACE = 2,
    /* with RTK_UPDATE_PRE_ROPE (ABI 15): q~ / k~ are what the REFERENCE scores and re-rotates - the un-rotation
     * ((x*cos) - (rotate_half(x)*sin)) / a^2 of the rotated rows this launch produced, one rounding of the model dtype
     * per torch op (longvideo_cache.py:76-78, :248-259) - instead of the inputs themselves.  On a bf16 / fp16 model the
     * round trip moves an operand by up to a few ulps, which is what the reference's scores and kept keys carry; with this
     * flag the score operands and the kept keys equal the reference's bit for bit (tables within one fp32 ulp of the rotary
     * module's).  Excludes RTK_UPDATE_Q_IN_PLACE; allows the reference-rounding score modes. */
    RTK_UPDATE_ROUNDTRIP = 4,
    /* without RTK_UPDATE_PRE_ROPE (ABI 16; the reference's protocol: q, k rotated by the caller): the launch that
     * un-rotates and appends the chunk also applies the continuity shift the NEXT layer's attention patch would launch
     * before its RoPE (qwen2_vl.py:68-73):  pos[0, 0:L] += (*next_prev + 1) - pos[0, 0], in place, once every workgroup
     * has read the ids this layer works with (each counts itself in on one of the counters in io->ticket; one extra
     * workgroup watches them, shifts and zeroes them again: nobody else waits).  `pos` is written despite its const.
     * Launches that share the ticket words must be stream ordered.
     * The watcher's wait is bounded by a poll count (~seconds; the workers need microseconds).  If it runs out - the
     * counters were not zero at launch: words shared between streams, or never zeroed - the ids are NOT shifted and the
     * counters NOT zeroed: the launch increments ticket[31] (and *io->status) and returns, and so does the watcher of every
     * later launch on these words (without waiting) until the caller has zeroed ticket and status again.  A caller that
     * skips its own rtk_position_shift on the strength of this flag must check the latch before it trusts the ids
     * (PivotKVCache raises RuntimeError).  Needs the chunk-batched passes or a keep-all batch: otherwise
     * RTK_EUNSUPPORTED is returned BEFORE anything is launched (nothing touched).  With this flag EVERY RTK_EUNSUPPORTED of
     * rtk_pivotkv_update is returned before the launch (alignment, extents, shape): a caller may always fall back.
     * The caller's next rtk_position_shift for that layer is then a no-op and may be skipped - the attention patch of a
     * 28-layer model launches it 27 times per chunk otherwise.  Only with the native prepare kernel (batch.inv_freq). */
    RTK_UPDATE_SHIFT_NEXT = 8
};
/* longvideo_cache.py:217-310 up to the deferred selection, for layer slot `slot`.  Rows go to the layer's tail
 * (ls->k/v + length rows; the caller has made sure length + L <= cap), ls->pending / pending_keep are set. */
int rtk_pivotkv_update(rtk_pivotkv_batch* batch, rtk_layer_state* layer, int slot, const rtk_update_io* io,
                       rtk_stream_t stream);

/* The attention patch's prologue for a segment that is NOT compressed - text prefill, decode (qwen2_vl.py:68-86 with the
 * cache's else-branch, longvideo_cache.py:319-321): continuity shift against the layer's last cached temporal id, rotary
 * tables, RoPE of q (to io->q_rot, which may alias io->q) and of k, the append of the rotated k and of v at the layer's
 * tail, and the shifted ids appended to the layer's position cache - one launch.  shift_ids_in_place (Qwen2-VL's in-place
 * semantics; LLaVA shifts a clone: 0): the caller's ids are shifted too - by the same launch for n == 1 (a decode step),
 * by rtk_position_shift after it otherwise.  io: q [Hq,n,D], k / v [Hkv,n,D],
 * pos [P,n].  The caller has made room (rows and ids); length / pos_len are advanced.  pos_embed_reforge caches only. */
int rtk_pivotkv_append_rope(rtk_layer_state* layer, const rtk_update_io* io, int Hq, int Hkv, int n, int D, int dtype, int P,
                            const float* inv_freq, float attention_scaling, const int* sections_host, int nsec,
                            int round_mode, int shift_ids_in_place, rtk_stream_t stream);

/* longvideo_cache.py:260-318 for the n pending layers slots[0..n) (ascending) of one chunk: score passes (unless
 * keep_all), selection, eviction (re-rotated K straight into the cache, V compacted in place) and the bookkeeping of
 * rtk_layer_state (length, pos_len, pending).  layers[i] is the state of slot slots[i].  Native RoPE (batch.inv_freq)
 * when reforging.  RTK_EUNSUPPORTED when the shape needs the per-stage entry points (L < 512, head_dim != 128 for the
 * batched passes). */
int rtk_pivotkv_flush(rtk_pivotkv_batch* batch, rtk_layer_state* const* layers, const int32_t* slots, int n,
                      rtk_stream_t stream);

/* G1  qwen2_vl.py:68-73 / llava_onevision.py:68-72, without the host round trip of the reference's
 * `if position_ids[0,0,0] != prev + 1`:  t[0:n] += (prev + 1) - t[0]  in place, where t is the temporal
 * row of the chunk's position ids and prev = *prev_dev (device memory: the last temporal id the layer's
 * cache holds) or -1 when prev_dev is NULL. */
int rtk_position_shift(int64_t* temporal_ids, int n, const int64_t* prev_dev, rtk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Direct peer-to-peer all-gather over xGMI (SURVEY §5 "Distributed backend", §8(b) rtk_allgather_*; used by
 * retake/p2p.py and, opt-in, retake/sharded.py).  xGMI is point to point: a rank that has mapped its peers'
 * receive buffers pushes its block into all of them at once (one hop per link) instead of walking a ring.
 *   - buffers: rtk_p2p_alloc (device memory; uncached != 0 for the flag words, which are polled while
 *     kernels of other ranks write them), exported / opened as hipIpc handles between the processes of a node;
 *   - rtk_p2p_push: nseg segments of seg_bytes from src (+ s * src_stride) to byte dst_offset + s * dst_stride
 *     of EVERY mapped buffer (the own one included), then - from a second launch behind the copy, so that a kernel
 *     boundary orders it after every workgroup of it; system-scope release - the value `epoch` into word `rank` of
 *     every peer's flag array  (ABI 17: the unused `counter` argument of ABI <= 16 is gone);
 *   - rtk_p2p_wait: returns (on the stream) once every sender has published an epoch >= `epoch`; bounded:
 *     after ~timeout_ms the kernel gives up and stores 1 + (first missing sender) into *status (device word,
 *     0 = fine), so that a lost peer is an error, not a hung GPU.
 * Epochs count up from 1; receivers that reuse a landing zone alternate two of them (a sender can be one
 * epoch ahead of a receiver, never two).  What landed in a zone at epoch n is guaranteed intact only for reads
 * the receiver enqueued BEFORE its own push n+1: a peer rewrites the zone at epoch n+2 as soon as its wait n+1
 * has seen that push, and later reads are ordered behind nothing.  All sizes, strides and offsets are multiples
 * of 16 bytes.
 * ------------------------------------------------------------------------------------------- */
#define RTK_IPC_HANDLE_BYTES 64
#define RTK_P2P_MAX_RANKS 16
typedef struct rtk_p2p_peers {
    void* buf[RTK_P2P_MAX_RANKS];        /* rank q's receive buffer as mapped in THIS process (own: the local pointer) */
    uint32_t* flag[RTK_P2P_MAX_RANKS];   /* rank q's flag array [world] */
} rtk_p2p_peers;
int rtk_p2p_alloc(size_t bytes, int uncached, void** ptr);      /* zero-filled */
int rtk_p2p_free(void* ptr);
int rtk_p2p_export(const void* ptr, void* handle_out /* RTK_IPC_HANDLE_BYTES */, size_t* offset_out);
int rtk_p2p_open(const void* handle, void** base_out);          /* ptr in this process = *base_out + offset */
int rtk_p2p_close(void* base);
int rtk_p2p_push(const void* src, size_t seg_bytes, int nseg, size_t src_stride_bytes, const rtk_p2p_peers* peers,
                 int rank, int world, size_t dst_offset_bytes, size_t dst_stride_bytes, uint32_t epoch,
                 rtk_stream_t stream);
int rtk_p2p_wait(const uint32_t* own_flags, int world, uint32_t epoch, int timeout_ms, uint32_t* status,
                 rtk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement support (bench.py).  When enabled, every kernel launch of this library is bracketed
 * by two hipEvents recorded on the launch stream; rtk_profile_collect() waits for them and folds
 * the elapsed times into per-kernel totals.  Process-wide, off by default.
 * ------------------------------------------------------------------------------------------- */
int rtk_profile_enable(int on);
int rtk_profile_enable_mask(unsigned kernel_id_mask);   /* bit k = time kernel id k only; 0 = off */
int rtk_profile_collect(void);
int rtk_profile_reset(void);
int rtk_profile_num_kernels(void);
const char* rtk_profile_kernel_name(int kernel_id);
int rtk_profile_read(int kernel_id, long long* launches, double* total_ms);
/* Calibration for the HBM rooflines: dst = src over 16-byte vectors with non-temporal loads and stores - the device
 * copy the "achievable" bandwidth beside the nominal 8 TB/s is measured with (bench.py: hbm_achievable). */
int rtk_profile_copy(void* dst, const void* src, size_t bytes, rtk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RETAKE_HIP_H */
