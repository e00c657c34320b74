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
 *  - row-major contiguous tensors; the caller allocates every output and workspace.
 *  - no hidden allocation, no host synchronisation, stream-ordered, re-entrant, graph-capturable.
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

#ifdef __cplusplus
}
#endif
#endif /* SWAT_HIP_H */
