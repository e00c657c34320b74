/*
 * include/swat_hip.h — C ABI of libswat_hip.so, the gfx950 (MI355X) kernels behind swat_amd.
 *
 * The reference (tian1327/SWAT) is pure Python and has no FFI of its own: every entry point below
 * replaces a PyTorch / open_clip op that the reference calls from Python, cited as
 * "ref: <file>:<lines>" relative to the reference tree.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers (tensor.data_ptr()), sizes, a hipStream_t passed as void*.
 *  - row-major contiguous tensors; the caller allocates every output and workspace (a *_workspace_bytes
 *    query per op); every call is stream-ordered and re-entrant.
 *  - the steady state of every entry point makes NO allocation and NO host synchronisation.  What the library
 *    owns beyond that is small, per device or per shape, created on FIRST use and kept for the life of the
 *    process (a caller that captures hipGraphs or measures allocator traffic warms each shape up once):
 *      * swat_vit_forward / swat_vit_backward in the ViT-B/32 regime (csrc/vit_mega.hip): 2 176 B of device
 *        hand-off counters + status, one 128-B pinned mapped host word (the give-up flag swat_vit_mega_poll
 *        reads) and one event per device; persistent launches of a device are ordered behind one another by
 *        that event (hipStreamWaitEvent, never a host wait).
 *      * swat_vit_backward / swat_vit_lane_stream / swat_vit_aux_stream: three side streams (weight-gradient
 *        lane, two auxiliary) and a ring of <= 1 024 ordering events, created once per process, back to back.
 *      * grouped / table-driven product launches (swat_linear_wgrad_bias_group, the one-launch weight
 *        gradients inside swat_vit_backward; csrc/gemm_pc.hip, csrc/gemm_xl.hip): a device descriptor table per
 *        distinct problem list (a few tens of KB + the same pinned; at most 8 table slots per device).  Re-filling a table
 *        slot whose previous contents may still be in flight calls hipStreamSynchronize on THAT slot's stream
 *        — only when a ninth distinct problem list evicts the least recently used one, never in a training
 *        loop (one list per model shape).
 *      * swat_jpeg_decode_batch: a pinned host staging arena and two device arenas per device, grown
 *        geometrically to the largest batch seen (growth = hipDeviceSynchronize + reallocation; a steady loader
 *        reaches its size in the first batches); the pinned buffer's re-use waits on the previous batch's H2D
 *        copy event (hipEventSynchronize: the host half of the decode fills that buffer), the device arenas'
 *        re-use across streams is ordered by an event; host entropy-decode threads are kept in a pool.
 *      * swat_vit_mega_status (a test / diagnostic call) synchronises the device; swat_gemm_prof_* (tuning
 *        header) own timing events.
 *    Nothing else allocates, and no compute entry point waits on the host for the GPU.
 *  - return 0 on success, a negative value on error; swat_last_error() has the text
 *    (thread-local).
 *  - dtype enum: SWAT_F32 = 0, SWAT_BF16 = 1.
 */
#ifndef SWAT_HIP_H
#define SWAT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* swat_stream_t; /* hipStream_t */

enum { SWAT_F32 = 0, SWAT_BF16 = 1 };

int swat_version(void);
const char* swat_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Path B — retrieval (ref: retrieval/sample_retrieval.py)
 * ------------------------------------------------------------------------------------------ */

/* B4  sims[i] = <feats[i,:], queries[row_seg[i],:]> as ONE fp32 fma chain over k = 0..d-1
 *     (the canonical order of oracle/retrieval_oracle.c; bit-exact with it).
 *     ref: t2t_similarity :397-416, cal_t2i_similarity :335-353 (a [n_c,d] @ [d,1] GEMV per
 *     class).  feats [n_rows,d] fp32 class-grouped; queries [n_seg,d]; row_seg [n_rows] int32
 *     (NULL = every row uses queries[0]).  d must be a multiple of 64. */
int swat_sim_segmented(const float* feats, int64_t n_rows, int d, const float* queries,
                       const int32_t* row_seg, float* sims, swat_stream_t stream);

/*     the same chain on a row list: s_r = <feats[row_index[r] - row_base], queries[row_seg[r]]> for r < n_idx
 *     (row_index[r] < 0 -> -inf).  The exact re-scoring step of the dense scan: candidates chosen by
 *     swat_sim_candidates_bf16 get the value the full scan would have produced, bit for bit.  sims (fp32, may be NULL)
 *     and / or keys_out (uint64 sort keys ordered(s_r) << 32 | ~row_index[r], 0 for padding, may be NULL). */
int swat_sim_gathered(const float* feats, const int64_t* row_index, int64_t n_idx, int d, const float* queries,
                      const int32_t* row_seg, float* sims, void* keys_out, int64_t row_base, swat_stream_t stream);
/*     completeness proof of the dense top-k: fails[0] (device or host-mapped pinned memory; stored, not added) = number of queries whose need-th exact value is not strictly
 *     above (K2-th best approximate score, or thr_dev[q] when cand_cnt[q] < K2) + eps_qnorm * sqrt(max_norm2[0]) [*
 *     qnorm_max[0] when that device scalar — max |q| over the batch — is given instead of folding it into
 *     eps_qnorm on the host: no host synchronisation for a new query batch], that kept fewer than `need` rows, or whose
 *     candidate list overflowed `cap`.  0 = the kept rows ARE the exact top-k. */
int swat_dense_topk_proof(const void* cand_cnt, int cap, int K2, const float* approx_val, const float* thr_dev,
                          const float* exact_val, int k, const int64_t* out_cnt, int need, float eps_qnorm,
                          const float* max_norm2, const float* qnorm_max, int n_q, int* fails, swat_stream_t stream);

/* out[r] = the rank-th largest (1-based) of scores[r, 0 .. n_cols): the per-query pruning threshold of the dense top-k, taken at a
 *     small rank of a row sample (the candidate pass keeps what scores above it; ref: the per-class ranking of
 *     retrieval/sample_retrieval.py:754,807 generalised to N_query x N_db).  Scores must hold no NaN.
 *     Rider (q_rows != null): q_rows [n_rows, d] are the queries; qnorm_max[0] = max_r |q_r| rounded up — the `qnorm_max` device scalar
 *     of swat_dense_topk_proof / swat_candidates_topk_proof — from the same launch.  norm_workspace: (1 + n_rows) floats, word 0
 *     zero before the first use (the launch leaves it zero). */
int swat_rank_select(const float* scores, int n_rows, int n_cols, int rank, float* out, const float* q_rows, int d,
                     void* norm_workspace, float* qnorm_max, swat_stream_t stream);
/* B5  filtered segmented top-k: for every segment s = rows [seg_offsets[s], seg_offsets[s+1])
 *     keep row iff sims >= thr [and t2i >= t2i_thr when t2i != NULL] [and !excluded[row]],
 *     order by (sims desc, row asc) and emit the first k.
 *     ref: sorted(..., reverse=True) :754/:807 + add_to_split :439-482 /
 *     add_t2t_ranked_t2i_tshd_to_split :492-540.
 *     out_idx [n_seg,k] int64 global row ids (-1 padded), out_val [n_seg,k] fp32 (NULL ok),
 *     out_cnt [n_seg] int64.  1 <= k <= 2048. */
int swat_segment_topk(const float* sims, const float* t2i, const uint8_t* excluded,
                      const int64_t* seg_offsets, int64_t n_seg, float thr, float t2i_thr, int k,
                      int64_t* out_idx, float* out_val, int64_t* out_cnt, swat_stream_t stream);

/* B4/B5 dense form (BASELINE config 4): S = Q DB^T on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 is a k-ordered
 *     fp32 fma chain: bit-identical to swat_sim_segmented and the oracle).  Entries >= thr_dev[q] are appended to
 *     cand_keys [n_q, cap] (uint64: ordered(sim) << 32 | ~(row_base + row)), cand_cnt [n_q] (uint32, zero it first);
 *     S_out [n_q, ld_s] optionally receives the full product (threshold sampling, tests).
 *     ref: the per-class GEMV loop :724-825 generalised to N_query x N_db. */
int swat_sim_topk_candidates(const float* queries, int n_q, const float* db, int64_t n_db, int d,
                             const float* thr_dev, int64_t row_base, void* cand_keys, int cap,
                             void* cand_cnt, float* S_out, int64_t ld_s, swat_stream_t stream);
/*     candidate pass at the HBM rate: database rows stream once as fp32, are rounded to bf16 in registers and multiplied
 *     with bf16 queries [n_q, d] on v_mfma_f32_16x16x32_bf16; (q, row) with approximate score >= thr_dev[q] go to the
 *     candidate lists (same key format; cand_cnt keeps counting past cap).  |approx - exact| <= (2^-8 + 2^-12) |q| |d|;
 *     max_norm2 (device float, zero it first) receives max |d_row|^2 so the caller can prove its top-k complete after the
 *     exact re-scoring (swat_sim_gathered).  csrc/retrieval_bf16.hip. */
int swat_sim_candidates_bf16(const void* queries_bf16, int n_q, const float* db, int64_t n_db, int d,
                             const float* thr_dev, int64_t row_base, void* cand_keys, int cap, void* cand_cnt,
                             float* max_norm2, swat_stream_t stream);
/*     the same pass in its fastest form (d % 128 == 0, n_q <= 4096): queries as the packed bf16 image made by
 *     swat_dense_pack_queries (fp32 [n_q, d] -> per 32-deep chunk and 16-query tile the 1 KiB LDS image the kernel reads
 *     linearly; swat_dense_packed_bytes(n_q, d) bytes, 0 when d is not supported) and a device workspace of
 *     swat_dense_workspace_bytes(n_q) bytes (0 when n_q is not supported) in which every wave stages its candidate
 *     records without atomics; a second launch files them into cand_keys / cand_cnt.  Same outputs, same error bound;
 *     an overflowing staging area is reported as cand_cnt[0] >= 2^31. */
int64_t swat_dense_packed_bytes(int n_q, int d);
int64_t swat_dense_workspace_bytes(int n_q);
int swat_dense_pack_queries(const float* queries, int n_q, int d, void* packed, swat_stream_t stream);
int swat_sim_candidates_packed(const void* packed, int n_q, const float* db, int64_t n_db, int d,
                               const float* thr_dev, int64_t row_base, void* cand_keys, int cap, void* cand_cnt,
                               float* max_norm2, void* workspace, int64_t workspace_bytes, swat_stream_t stream);
/*     The same pass over a bf16 SHADOW of the shard, built once per index (the reference keeps the fp32 matrix of mined.pth in
 *     memory for every class query, retrieval/sample_retrieval.py:1387-1415; this is its scan-side image): rows rounded to bf16
 *     once and stored in the scan's order — for d % 256 == 0 tiled: blocks of 256 rows, inside a block one 64-deep k-step of
 *     all its rows is contiguous (32 KB, in the order the scan's lanes load it: 1 KiB per 16-row fragment and 32-deep half),
 *     rows padded with zeros to whole blocks; swat_dense_shadow_bytes(n_db, d) = 2 d x
 *     n_db rounded up to 256 (d % 256 == 0) or 2 n_db d (other multiples of 128); 0: d not supported.
 *     swat_dense_build_shadow also leaves max |row|^2 of the fp32 rows in *max_norm2 (zero it first).  The candidate lists are
 *     those of swat_sim_candidates_packed bit for bit (same roundings, same order of accumulation) from half the HBM bytes;
 *     the exact re-scoring and the completeness proof still read the fp32 rows.  swat_sim_candidates_shadow wants the first
 *     16 KiB of `workspace` (the staging-slot counters) ZERO on entry and leaves them zero: clear the workspace once when it
 *     is allocated. */
int64_t swat_dense_shadow_bytes(int64_t n_db, int d);
int swat_dense_build_shadow(const float* db, int64_t n_db, int d, void* shadow, float* max_norm2, swat_stream_t stream);
int swat_sim_candidates_shadow(const void* packed, int n_q, const void* shadow, int64_t n_db, int d, const float* thr_dev,
                               int64_t row_base, void* cand_keys, int cap, void* cand_cnt, void* workspace,
                               int64_t workspace_bytes, swat_stream_t stream);
/*     sort each query's candidates, emit the (value desc, row asc) top-k (cap <= 8192). */
int swat_candidates_topk(const void* cand_keys, int cap, const void* cand_cnt, int n_q, int k,
                         int64_t* out_idx, float* out_val, int64_t* out_cnt, swat_stream_t stream);
/*     the k2 best candidates of every query UNORDERED (radix select instead of a sort: the exact re-scoring step orders them
 *     again): slots 0 .. k2-2 in arbitrary order, slot k2-1 = the k2-th best (the bound the completeness proof reads);
 *     fewer than k2 candidates: all of them, then (-1, -inf).
 *     zero_fails (device int, may be NULL) is set to 0: the failure counter swat_candidates_topk_proof adds to later in the call. */
int swat_candidates_select(const void* cand_keys, int cap, const void* cand_cnt, int n_q, int k2,
                           int64_t* out_idx, float* out_val, int64_t* out_cnt, int* zero_fails, swat_stream_t stream);
/*     swat_candidates_topk over the re-scored keys with swat_dense_topk_proof folded into the same launch: fails (DEVICE int,
 *     zeroed earlier in the call) += queries whose proof does not hold; scan_cnt / scan_cap = the candidate pass's counts / cap;
 *     zero_scan_cnt != 0: scan_cnt[q] is cleared once read (cand_cnt must then be the same array or no longer needed). */
int swat_candidates_topk_proof(const void* cand_keys, int cap, const void* cand_cnt, int n_q, int k, int64_t* out_idx,
                               float* out_val, int64_t* out_cnt, void* scan_cnt, int scan_cap, int K2,
                               const float* approx_val, const float* thr_dev, int need, float eps, const float* max_norm2,
                               const float* qnorm_max, int* fails, int zero_scan_cnt, swat_stream_t stream);
/*     SURVEY 8(e) retrieval row: k-way merge of the per-shard top-k lists after the all_gather (the sharded form of the
 *     reference's single sorted() per class, retrieval/sample_retrieval.py:754,807).  idx / val [n_q, m]: the shards' lists
 *     side by side in rank order (global rows, -1 padded), m <= 8192; output = the k best by (value desc, global row asc).
 *     workspace: swat_merge_topk_workspace_bytes(n_q, m) bytes. */
int64_t swat_merge_topk_workspace_bytes(int n_q, int m);
int swat_merge_topk_lists(const int64_t* idx, const float* val, int n_q, int m, int k, void* workspace,
                          int64_t* out_idx, float* out_val, int64_t* out_cnt, swat_stream_t stream);
/* B6  remove_near_duplicates2 (ref :237-275): marks[j] = 1 iff <f_i, f_j> > thr for some i < j (S = F F^T on the
 *     fp32 matrix cores, exact canonical order). */
int swat_near_dup_marks(const float* feats, int64_t n, int d, float thr, uint8_t* marks,
                        swat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Path A — OpenCLIP ViT finetuning step (ref: utils/training.py train_ce :326-347,
 * train_cutmix :1452-1489; the ViT arithmetic is open_clip 2.26.1's VisionTransformer)
 * ------------------------------------------------------------------------------------------ */

/* A2a,c,e,f,h  bf16 MFMA GEMM, fp32 accumulate, fused epilogue.  layout 0 = NT (Y = X W^T, forward),
 *     1 = NN (dX = dY W, dgrad), 2 = TN (dW = dY^T X, wgrad).  Replaces nn.Linear / conv1 /
 *     nn.MultiheadAttention matmuls and their autograd (cuBLAS in the reference).
 *     Batched: block z uses offsets (z / inner) * s?0 + (z % inner) * s?1 (elements).
 *     Epilogue: v = alpha*acc (+bias[n]) (+residual[m,n] fp32, may alias C) (*act'(actin[m,n]));
 *     C is bf16 or fp32 (out_f32); C2 (bf16, optional) = act(C) with act 1 = QuickGELU, 2 = erf-GELU; C may be NULL with a
 *     C2: only the activated output is stored (inference: nothing re-reads the pre-activation), same values.
 *     lda/ldb multiples of 8, ldc multiple of 4 and >= roundup(N,4). */
int swat_gemm_bf16(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda,
                   int ldb, int ldc, int batch, int inner, int64_t sA0, int64_t sA1, int64_t sB0,
                   int64_t sB1, int64_t sC0, int64_t sC1, const float* bias, const float* residual,
                   const void* actin, int ldact, int act, void* C2, int out_f32, float alpha,
                   swat_stream_t stream);

/*     Measurement and tuning entry points (per-launch GEMM timing for bench.py's roofline leg, tile pinning for the
 *     tests, ablation switches that make results WRONG on purpose) are NOT part of this interface: see
 *     include/swat_hip_tuning.h. */

/* A2b,f,g  LayerNorm (eps inside rsqrt, biased variance = torch.nn.LayerNorm). x fp32 with an explicit
 *     row stride (so the cls rows of [B,T,d] can be normalised in place); y in bf16 and/or fp32. */
int swat_layernorm_fwd(const float* x, int64_t x_row_stride, const float* gamma, const float* beta,
                       void* y_bf16, float* y_f32, float* mean, float* rstd, int rows, int d,
                       float eps, swat_stream_t stream);
/*     same, with x given as `nsplit` split-K partial planes (plane 0 carries bias + residual): sums them in plane order,
 *     writes the materialised x to x_sum_out (same row stride) and normalises it. */
int swat_layernorm_fwd_split(const float* x_planes, int64_t x_row_stride, int nsplit, int64_t split_stride,
                             float* x_sum_out, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                             float* mean, float* rstd, int rows, int d, float eps, swat_stream_t stream);
int64_t swat_layernorm_bwd_workspace_bytes(int rows, int d);
/*     dx_out = (dx_add ? dx_add : 0) + LN'(dy); optional bf16 copy; dgamma/dbeta reduced in a fixed
 *     order through `workspace` (deterministic).  dcolsum (optional) = column sums of the bf16 output = the bias
 *     gradient of the linear layer whose output stream this is (saves a separate pass over dx). */
int swat_layernorm_bwd(const float* dy, int64_t dy_stride, const float* x, int64_t x_stride,
                       const float* gamma, const float* mean, const float* rstd, const float* dx_add,
                       int64_t dxa_stride, float* dx_out, int64_t dxo_stride, void* dx_bf16,
                       int64_t dxb_stride, float* dgamma, float* dbeta, float* dcolsum, void* workspace,
                       int rows, int d, swat_stream_t stream);

/*     same, with dy given as `nsplit` partial planes (split_stride floats apart) that are summed on the fly: the
 *     split-K dgrad GEMM that feeds it writes partial sums instead of paying a reduction pass. */
int swat_layernorm_bwd_split(const float* dy, int64_t dy_stride, int nsplit, int64_t split_stride, const float* x,
                             int64_t x_stride, const float* gamma, const float* mean, const float* rstd,
                             const float* dx_add, int64_t dxa_stride, float* dx_out, int64_t dxo_stride,
                             void* dx_bf16, int64_t dxb_stride, float* dgamma, float* dbeta, float* dcolsum,
                             void* workspace, int rows, int d, swat_stream_t stream);
/*     the same in two calls — the row pass, then the reduction of its per-workgroup partial sums (`workspace`) to
 *     dgamma / dbeta / dcolsum; the reduction may be issued on another stream that is ordered after the row pass. */
int swat_layernorm_bwd_partial(const float* dy, int64_t dy_stride, int nsplit, int64_t split_stride, const float* x,
                               int64_t x_stride, const float* gamma, const float* mean, const float* rstd,
                               const float* dx_add, int64_t dxa_stride, float* dx_out, int64_t dxo_stride,
                               void* dx_bf16, int64_t dxb_stride, void* workspace, int rows, int d,
                               swat_stream_t stream);
int swat_layernorm_bwd_finish(const void* workspace, int rows, int d, float* dgamma, float* dbeta, float* dcolsum,
                              swat_stream_t stream);
/*     swat_layernorm_bwd_partial for LayerNorm rows that sit `period` rows apart in a contiguous d-wide output (dx_out fp32 and / or
 *     dx_bf16, [rows * period, d]; LayerNorm row j -> output row j * period): the same launch also writes ZEROS to every output row
 *     in between — the tower's top gradient, where only the class-token row of an image carries one (ref: `pooled = x[:, 0]` in
 *     open_clip's VisionTransformer.forward behind model.encode_image, utils/training.py:1474).  Partial sums as
 *     swat_layernorm_bwd_partial (reduce with swat_layernorm_bwd_finish(workspace, rows, d, ...)). */
int swat_layernorm_bwd_partial_fill(const float* dy, int64_t dy_stride, const float* x, int64_t x_stride, const float* gamma,
                                    const float* mean, const float* rstd, float* dx_out, void* dx_bf16, int period,
                                    void* workspace, int rows, int d, swat_stream_t stream);

/* A2d  attention softmax over score rows (fp32 S, ld = padded T): P = softmax(scale*S[:, :T]) bf16,
 *     pad columns zeroed; causal!=0 for the text tower.  Backward: dS = scale*P o (dP - rowsum(P o dP)). */
int swat_softmax_fwd(const float* S, void* P_bf16, int64_t rows, int Tq, int T, int ld, float scale,
                     int causal, swat_stream_t stream);
int swat_softmax_bwd(const void* P_bf16, const float* dP, void* dS_bf16, int64_t rows, int T, int ld,
                     float scale, swat_stream_t stream);

/* A5 + A2a  CutMix box paste fused with patch extraction and the bf16 cast:
 *     patches[b*g*g + p, k] = images[src(b,y,x), c, y, x], src = perm[b] inside [x1,x2) x [y1,y2)
 *     (dims 2,3; ref utils/training.py:1466), k = c*ps*ps + (y%ps)*ps + (x%ps).  perm NULL = no mix. */
int swat_patchify_cutmix(const float* images, const int64_t* perm, void* patches_bf16, int B, int H,
                         int W, int patch, int ldp, int x1, int x2, int y1, int y2,
                         swat_stream_t stream);
/* A5  the paste alone (out of place; dst must not alias src). */
int swat_cutmix_paste(const float* src, const int64_t* perm, float* dst, int B, int H, int W, int x1,
                      int x2, int y1, int y2, swat_stream_t stream);
/* A2b  x[b,0,:] = class_embedding + positional_embedding[0]. */
int swat_cls_row(float* x, const float* cls, const float* pos, int B, int T, int d,
                 swat_stream_t stream);

/* A2h  bias gradients: out[n] = sum over rows of X[:, n] (bf16 or fp32), fixed summation order.
 *     Optional row grouping (logical row r -> (r/group_rows)*group_stride_rows + r%group_rows + row_off). */
int swat_colsum(const void* X, int dtype, int64_t rows, int n, int ld, float* out, int64_t group_rows,
                int64_t group_stride_rows, int64_t row_off, swat_stream_t stream);
/*     two-stage form for the large bias gradients (rows = B*T): workspace of swat_colsum_bf16_workspace_bytes(n). */
int64_t swat_colsum_bf16_workspace_bytes(int n);
int swat_colsum_bf16(const void* X_bf16, int64_t rows, int n, int ld, float* out, void* workspace,
                     swat_stream_t stream);
/* A2h  weight gradient and bias gradient of one linear layer (autograd of nn.Linear in the reference):
 *     dW [n_out, n_in] = dY^T X, dbias [n_out] = column sums of dY; dY [rows, ld_dy], X [rows, ld_x] bf16.
 *     One launch when rows % 64 == 0 (the GEMM's n-tile-0 workgroups add an MFMA against a ones fragment per k-step),
 *     else GEMM + the two-stage column sum; workspace: swat_colsum_bf16_workspace_bytes(n_out). */
int swat_linear_wgrad_bias(const void* dY_bf16, const void* X_bf16, float* dW, float* dbias, int rows, int n_out,
                           int n_in, int ld_dy, int ld_x, void* workspace, swat_stream_t stream);
/*     the same for n <= 4 layers in ONE persistent launch over the joint tile list (the four weight gradients of a
 *     transformer block are independent; the reference's autograd runs them as separate cuBLAS calls).  Arrays are HOST
 *     arrays of length n; dbias[i] may be null; rows[i] % 64 != 0 falls back to one launch per layer. */
int swat_linear_wgrad_bias_group(int n, const void* const* dY_bf16, const void* const* X_bf16, float* const* dW,
                                 float* const* dbias, const int* rows, const int* n_out, const int* n_in,
                                 const int* ld_dy, const int* ld_x, void* workspace, swat_stream_t stream);
int swat_gather_cast_bf16(const float* in, void* out_bf16, int64_t rows, int d, int64_t group_rows,
                          int64_t group_stride_rows, int64_t row_off, swat_stream_t stream);
int swat_cast_bf16(const float* in, void* out_bf16, int64_t n, swat_stream_t stream);
/*     bf16 -> fp32 (the receiving side of the data-parallel step's optional bf16 gradient wire) */
int swat_widen_bf16(const void* in_bf16, float* out, int64_t n, swat_stream_t stream);

/* A7  fused multi-group AdamW over one flat fp32 buffer (torch.optim.AdamW semantics, ref
 *     utils/optimizers.py:131-146) + refresh of the bf16 shadow weights.  seg_begin: HOST array of
 *     n_seg group start offsets (first = 0); lrs_dev [n_seg] and bias_corr_dev {1-b1^t, 1-b2^t} live
 *     on the device so a captured step replays with new values. */
int swat_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
               int64_t n, const int64_t* seg_begin, int n_seg, const float* lrs_dev, float beta1,
               float beta2, float eps, float weight_decay, const float* bias_corr_dev,
               float grad_scale, swat_stream_t stream);

/* A3  L2 normalise rows, no eps (ref utils/training.py:333,1475) and its backward. */
int swat_l2norm_fwd(const float* x, float* y, float* norms, int rows, int d, swat_stream_t stream);
int swat_l2norm_bwd(const float* y, const float* norms, const float* dy, float* dx, int rows, int d,
                    swat_stream_t stream);
/* A4  small strided fp32 GEMM (head / proj and their gradients; M = batch):
 *     C[m,n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] (+= if accumulate). */
int swat_sgemm_small(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam,
                     int64_t sak, int64_t sbk, int64_t sbn, int64_t ldc, float alpha, int accumulate,
                     swat_stream_t stream);
/* A6  mixed cross entropy on logits = z * exp(logit_scale) (ref mixup_criterion :1157-1158 over
 *     nn.CrossEntropyLoss mean reduction): loss, d loss/d z, d loss/d logit_scale; lam = 1, yb = ya
 *     gives plain CE (train_ce).  per_sample_ws: 2*B floats.  loss_accum (optional, float64) += loss. */
int swat_ce_mixed(const float* z, const float* logit_scale_dev, const int64_t* ya, const int64_t* yb,
                  float lam, float* per_sample_ws, float* loss, float* dlogit_scale, float* dz,
                  float* logits_out, double* loss_accum, int B, int C, swat_stream_t stream);

/* A9 / f.4  validate's per-batch tail on the device (ref testing.py:449-458 argmax + torchmetrics ConfusionMatrix.update,
 *     :417): pred[b] = first index of the maximum of row b of z [B, C]; confusion[label, pred] += 1 ([C, C] int64, rows =
 *     true class, accumulated across calls — the caller zeroes it once per validate); *correct += (pred == label).
 *     pred_out / max_out / confusion / correct may each be null. */
int swat_argmax_confusion(const float* z, const int64_t* labels, int B, int C, int64_t* pred_out, float* max_out,
                          int64_t* confusion, int64_t* correct, swat_stream_t stream);

/* A2d  nn.MultiheadAttention core for short sequences (T <= 64, head width 64: ViT-B/32 at 224 px has T = 50): one
 *     workgroup per (image, head) does QK^T, softmax, PV on the matrix cores; the backward recomputes P from Q and K.
 *     qkv [B*T, 3d] packed as in_proj leaves it, out / dout [B*T, d], dqkv [B*T, 3d]; all bf16. */
int swat_attn_fwd_small(const void* qkv_bf16, void* out_bf16, int B, int T, int d, int H, swat_stream_t stream);
int swat_attn_bwd_small(const void* qkv_bf16, const void* dout_bf16, void* dqkv_bf16, int B, int T, int d, int H,
                        swat_stream_t stream);
/*     64 < T <= 320: one workgroup per (image, head, 64-query tile) walks the key tiles with the running-maximum
 *     softmax; lse [B*H, T] (optional) receives the row log-sum-exp the backward needs; causal = 1 masks keys > query
 *     (text tower). */
int swat_attn_fwd_tiled(const void* qkv_bf16, void* out_bf16, float* lse, int B, int T, int d, int H, int causal,
                        swat_stream_t stream);
/*     its backward in two launches (dQ per query tile; dK, dV per key tile), each recomputing P from Q, K and lse;
 *     out = the forward's output (for delta = rowsum(dO * O), which the first launch leaves in `workspace`:
 *     swat_attn_bwd_tiled_workspace_bytes(B, T, H) bytes). */
int64_t swat_attn_bwd_tiled_workspace_bytes(int B, int T, int H);
int swat_attn_bwd_tiled(const void* qkv_bf16, const void* out_bf16, const void* dout_bf16, const float* lse,
                        void* dqkv_bf16, int B, int T, int d, int H, void* workspace, swat_stream_t stream);

/* A2 (whole)  model.encode_image(images) as one call: CutMix-aware patch embed, cls/pos, ln_pre, L
 *     pre-LN blocks, ln_post(cls), proj -> feats [B,E] fp32 (un-normalised).
 *     dims (HOST int32[10]): B, image_size, patch, width d, layers L, heads H, mlp width, embed E,
 *     act (1 QuickGELU / 2 erf-GELU), save_for_backward.
 *     offs (HOST int64[8 + 12*L], elements into params / shadow): conv1.weight, class_embedding,
 *     positional_embedding, ln_pre.{w,b}, ln_post.{w,b}, proj, then per block ln_1.{w,b},
 *     attn.in_proj_{weight,bias}, attn.out_proj.{weight,bias}, ln_2.{w,b}, mlp.c_fc.{weight,bias},
 *     mlp.c_proj.{weight,bias}.  params = fp32 master, shadow_bf16 = bf16 copy at the same offsets.
 *     perm/x1..y2: CutMix permutation and box (ref utils/training.py:1462-1466), perm NULL = none.
 *     workspace: swat_vit_workspace_bytes(dims) bytes, ZEROED ONCE by the caller before its first use (the buffers a
 *     weight gradient reduces over are padded to a multiple of 64 rows; the kernels never write the padding). */
int64_t swat_vit_workspace_bytes(const int32_t* dims);
int swat_vit_forward(const int32_t* dims, const int64_t* offs, const float* params,
                     const void* shadow_bf16, const float* images, const int64_t* perm, int x1, int x2,
                     int y1, int y2, void* workspace, float* feats, swat_stream_t stream);
/* A2h  backward of the above in stages (L+1: proj+ln_post, l+1: block l, 0: stem), run from stage_hi
 *     down to stage_lo, so the host can reduce a stage's gradients while the next one runs.
 *     Gradients are written (not accumulated) into `grads` at the parameters' offsets. */
int swat_vit_backward(const int32_t* dims, const int64_t* offs, const float* params,
                      const void* shadow_bf16, void* workspace, const float* dfeats, float* grads,
                      int stage_hi, int stage_lo, swat_stream_t stream);
/* Weight-gradient lane of swat_vit_backward.  mode 1 (default): the weight-gradient GEMMs and bias column
 *     sums run on an internal lower-priority stream, ordered by events against the data-gradient chain on
 *     `stream`, and the call joins that lane before it returns.  mode 2: as 1 but the caller joins
 *     (swat_vit_backward_join) after its last stage call; after each call, work ordered behind
 *     swat_vit_lane_stream() sees every gradient of the stages issued so far, so a per-stage all-reduce /
 *     optimizer slice can trail the backward without stalling it.  mode 0: everything on `stream`.
 *     Same kernels and buffers in every mode: the gradients are bit-identical. */
int swat_vit_set_overlap(int mode);
swat_stream_t swat_vit_lane_stream(void);
swat_stream_t swat_vit_aux_stream(int i); /* i = 0, 1: two more process-wide streams created together with the lane (the
                                            * data-parallel step's all-reduce and optimizer streams) */
int swat_vit_lane_order(swat_stream_t stream); /* `stream` waits for what the lane holds now; the lane is not joined */
int swat_vit_lane_follow(swat_stream_t stream); /* the lane waits for what `stream` holds now (then work may be queued on swat_vit_lane_stream()) */
int swat_vit_backward_join(swat_stream_t stream);
/* swat_vit_backward_join that leaves the join OWED instead of making `stream` wait now: what the lane and the second internal stream
 *     still hold after a one-GPU step (the blocks' weight gradients and their AdamW) is read again only by the blocks of the next
 *     forward, so the next step's stem (CutMix / patch extraction, patch embedding, class row, ln_pre: stage-0 parameters, stepped on
 *     `stream` itself) may run beside the optimizer's tail.  swat_vit_forward pays the debt behind its stem; swat_vit_backward,
 *     swat_vit_backward_join and swat_vit_join_pending(stream) pay it at once.  A caller that reads parameters, moments or gradients
 *     itself (checkpoint, torch ops on the arena) calls swat_vit_join_pending(its stream) first.
 *     ref: one process, one stream in the reference (utils/training.py:1486-1489): nothing there to order. */
int swat_vit_backward_join_later(swat_stream_t stream);
int swat_vit_join_pending(swat_stream_t stream);
/* Data-parallel form of the blocks' backward (gradient all-reduce overlapped with backward work; ref: the
 *     reference is one process, utils/training.py:1486-1488).  swat_vit_backward_set_buckets names buckets of transformer blocks from
 *     the top of the tower down (n <= 16; the last one takes what remains).  The NEXT swat_vit_backward call that runs the blocks as
 *     the persistent launch (stage_hi >= L, stage_lo <= 1, swat_vit_backward_blocks_fused) then issues on the weight-gradient lane
 *     the LayerNorm reductions of all blocks and, per bucket, one weight-gradient launch with an event behind it;
 *     swat_vit_backward_bucket_wait(i, stream) orders `stream` behind bucket i's event, so bucket i's all-reduce and AdamW run
 *     beside the weight gradients of the buckets below.  swat_vit_backward_buckets_recorded(): events of the last such call
 *     (0: the blocks ran op by op and the list was dropped).  One call per arming; no optimizer work inside. */
int swat_vit_backward_set_buckets(const int32_t* blocks, int n);
int swat_vit_backward_bucket_wait(int i, swat_stream_t stream);
int swat_vit_backward_buckets_recorded(void);
/* Data-parallel runs of the persistent launches: a hand-off give-up on ANY rank voids the step on EVERY rank.
 *     swat_vit_mega_void_flag(dst): dst[0] (device float) = 1 if this device's guard is up (a persistent launch gave up), else 0;
 *     the caller reduces dst over the ranks (MAX or SUM) on the stream its gradient collectives run on, before the step's first
 *     optimizer launch, then swat_vit_mega_void_absorb(src, step_seq): src[0] != 0 raises the local guard too (every swat_adamw /
 *     swat_loss_accumulate from here on is skipped, as after a local give-up) and publishes step_seq (>= 1, the caller's step
 *     count) in a pinned host word; swat_vit_mega_poll_void() reads it without synchronising (0: no step voided since the last
 *     swat_vit_mega_status).  The host then replays from that step on every rank alike (swat_amd/step.py). */
int swat_vit_mega_void_flag(float* dst, swat_stream_t stream);
int swat_vit_mega_void_absorb(const float* src, int step_seq, swat_stream_t stream);
int swat_vit_mega_poll_void(void);
/* AdamW (ref utils/optimizers.py:131-146, torch.optim.AdamW) of the blocks' parameters INSIDE the next swat_vit_backward call that
 *     covers all blocks (stage_hi >= L, stage_lo <= 1) and runs them as the persistent backward: block by block beside the next
 *     block's weight-gradient launch, or in the epilogue of the weight-gradient launch (swat_vit_set_block_adamw) — same
 *     arithmetic as swat_adamw, one learning-rate group.
 *     Pointers are the arena bases (params / grads / shadow must be the ones the backward call is given; lr_dev, bias_corr_dev:
 *     DEVICE scalars as for swat_adamw).  The arming is used or dropped by that one call; swat_vit_fused_adamw_consumed() then says
 *     (and clears) whether the blocks' range has been stepped — if not (shape not served, data-parallel run), the caller launches
 *     swat_adamw for it as usual.  Single-process training only: a data-parallel step reduces the gradients first. */
/* loss_accum (float64, device) += loss (device scalar), stream-ordered; skipped while the device guard of a persistent launch that
 *     gave up stands.  For a step whose backward runs as the persistent launch the running loss sum is fed here, behind that
 *     backward, instead of by swat_head_loss (pass it loss_accum = null): a give-up then voids the step as a whole. */
int swat_loss_accumulate(const float* loss, double* loss_accum, swat_stream_t stream);
int swat_vit_arm_fused_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                             const float* lr_dev, const float* bias_corr_dev, float beta1, float beta2, float eps,
                             float weight_decay, float grad_scale);
int swat_vit_fused_adamw_consumed(void);
/* How an armed AdamW runs: 1 (default; env SWAT_BLOCK_ADAMW) piped — per block on a second internal stream beside the next
 *     block's weight-gradient launch; 2 inside the epilogue of ONE weight-gradient launch over all blocks (the gradient tile never
 *     goes to memory); 0 not at all (the arming is dropped, the caller's own launch follows).  Results are bit-identical. */
int swat_vit_set_block_adamw(int mode);
/* The next swat_vit_backward call for stage L+1 alone may leave d proj (read only by the optimizer) pending; the blocks' call that
 *     follows issues it behind the persistent backward, any other backward / join call flushes it first.  One call; for callers that
 *     read that stage's gradients only after the blocks' call. */
int swat_vit_backward_defer_top(void);
/* A3+A4+A6 fused with their backward (ref utils/training.py:1475-1481): loss, d/dfeats, d/dW_head,
 *     d/dlogit_scale.  want_grad = 0: forward only (validate, testing.py:430-447); 2 then 3: the same call in two parts on the
 *     same workspace and arguments — 2 issues what the tower's backward waits for (d/dfeats), 3 the batch means (loss,
 *     d/dlogit_scale, loss_accum) and d/dW_head, which only the optimizer and the logger read; z_out / feat_norm_out
 *     optionally receive the un-scaled head output and the normalised features.  normalize = 0: feats are
 *     pre-extracted unit-norm features and are used as they are (train_probing, utils/training.py:96-97). */
int64_t swat_head_loss_workspace_bytes(int B, int E, int C);
int swat_head_loss(const float* feats, const float* W, const float* logit_scale_dev, const int64_t* ya,
                   const int64_t* yb, float lam, void* workspace, float* loss, float* dfeats, float* dW,
                   float* dlogit_scale, float* z_out, float* feat_norm_out, double* loss_accum, int B,
                   int E, int C, int want_grad, int normalize, swat_stream_t stream);

/* §8(f).1  CLIP text tower pieces (ref: model.encode_text — utils/features.py:27-36, extract_mined_feature.py:119-121):
 *     token-embedding gather + positional add, and a forward-only stack of L pre-LN blocks on x [B*T, d] fp32 (in
 *     place) with an optional causal mask.  dims (HOST int32[8]): B, T, d, L, H, mlp, act, causal; offs 12 per block. */
int swat_embed_tokens(const int64_t* tokens, const float* table, const float* pos, float* out, int64_t rows,
                      int T, int d, swat_stream_t stream);
int64_t swat_transformer_workspace_bytes(const int32_t* dims);
int swat_transformer_forward(const int32_t* dims, const int64_t* offs, const float* params,
                             const void* shadow_bf16, float* x, void* workspace, swat_stream_t stream);

/* §8(f).2, decode half (ref: `self.loader(path)` = PIL `Image.open(path).convert('RGB')` — utils/datasets/dataset_utils.py:130-180,
 *     in 8 worker processes: utils/dataloader.py:107-135; retrieval/extract_mined_feature.py:103-117).  Baseline JPEG (SOF0, 8 bit,
 *     1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, restart intervals): Huffman entropy decode on host threads, dequantisation + libjpeg's
 *     "islow" inverse DCT + fancy chroma upsampling + YCbCr -> RGB on the GPU, bytes identical to Pillow's, written in the layout
 *     swat_preprocess_batch reads.
 *     swat_jpeg_info           : width / height from the headers.  -> 0, 1 = not a file this path decodes (progressive, CMYK, other
 *                                sampling, not a JPEG: the caller keeps such a file on Pillow), 2 = corrupt.
 *     swat_jpeg_entropy_decode : the host half alone — quantised coefficient blocks, int16, natural order,
 *                                [component][block row][block column][64], block grids padded to whole MCUs; dims (HOST int32[8]) = W, H,
 *                                components, luma h, luma v, MCUs across, MCUs down, 0; coef_i16 may be null (dims only).
 *     swat_jpeg_decode_batch   : files / nbytes / offsets / status are HOST arrays; pixels_u8 is DEVICE memory, image i (HWC RGB) at byte
 *                                offsets[i]; status[i] = 0 decoded, 1 / 2 as above (nothing written).  threads <= 0: chosen by the library. */
int swat_jpeg_info(const void* file, int64_t nbytes, int32_t* width, int32_t* height);
int swat_jpeg_entropy_decode(const void* file, int64_t nbytes, int32_t* dims, void* coef_i16, int64_t coef_capacity);
int swat_jpeg_decode_batch(const void* const* files, const int64_t* nbytes, int n, void* pixels_u8, const int64_t* offsets,
                           int32_t* status, int threads, swat_stream_t stream);

/* §8(f).2  Image side of the input pipeline (ref: utils/extras.py:66-85 `transform` = RandomResizedCrop(224, bicubic)
 *     + RandomHorizontalFlip + ToTensor + Normalize for training, Resize(224, bicubic) + CenterCrop + ToTensor +
 *     Normalize for validation; applied per sample at utils/datasets/dataset_utils.py:165-180).  Decoded RGB images go
 *     in as bytes, the normalised fp32 batch [n_images, 3, n_px, n_px] comes out; the resampling is Pillow's 8-bit
 *     antialiased bicubic (two passes, 22-bit fixed-point weights), bit for bit.  The random draws stay on the host.
 *     pixels_u8 : all images back to back, HWC uint8, image i at byte offsets[i]; the allocation must extend 4 bytes
 *                 past the last image (rows are fetched as aligned 32-bit words).
 *     desc      : DEVICE int32 [n_images][12] = { width, height (stored image), x0, y0, in_w, in_h (region that is
 *                 resampled), out_w, out_h (size it is resampled to), off_x, off_y (corner of the n_px window kept from
 *                 that result), flip (mirror the window left-right), 0 }.
 *                 training: region = the random crop, out = n_px x n_px, off = 0; validation: region = whole image,
 *                 out = (nw, nh) with the short side n_px, off = the centre-crop corner.
 *     row_start : DEVICE int64 [n_images + 1], prefix sums of in_h.       lut: DEVICE float [3][256], the value of
 *                 (v / 255 - mean_c) / std_c for every byte v (computed once by the host in float32).
 *     max_in_w / max_in_h / max_ksize: maxima over the batch; ksize = 2 * ceil(2 * max(in / out, 1)) + 1 per axis. */
int64_t swat_preprocess_workspace_bytes(int n_images, int n_px, int max_ksize, int64_t total_rows);
int swat_preprocess_batch(const void* pixels_u8, const int64_t* offsets, const int32_t* desc, const int64_t* row_start,
                          const float* lut, float* out, int n_images, int n_px, int max_in_w, int max_in_h,
                          int max_ksize, void* workspace, int64_t workspace_bytes, swat_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SWAT_HIP_H */
