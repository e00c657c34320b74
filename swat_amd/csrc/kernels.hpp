// swat_amd/csrc/kernels.hpp — internal declarations shared between the kernel files and the composite.
#pragma once
#include "common.hpp"

namespace swat {

enum { L_NT = 0, L_NN = 1, L_TN = 2 };

struct GemmArgs {
    const unsigned short* A;
    const unsigned short* B;
    int M, N, K;
    int lda, ldb, ldc;
    int inner;  // batched: z = blockIdx.z ; off = (z / inner) * s0 + (z % inner) * s1   (elements)
    int64_t sA0, sA1, sB0, sB1, sC0, sC1;
    void* C;                       // primary output (bf16 or f32)
    void* C2;                      // optional second output (bf16): act(C)
    int c2_only;                   // with C2: do not store C (inference: nothing re-reads the pre-activation)
    const float* bias;             // [N] fp32 or null
    const float* residual;         // [M,N] fp32 (ld = ldc) or null (may alias C)
    const unsigned short* actin;   // [M,N] bf16 (ld = ldact): dgrad through the activation
    int ldact;
    int act;                       // 0 none, 1 quick_gelu, 2 gelu(erf)
    int out_f32;
    float alpha;
    int rpg, gstride, roff;        // output row remap: out_row = (m / rpg) * gstride + (m % rpg) + roff (rpg = 0: identity)
    const float* rowbias;          // [.., N] fp32 added as rowbias[(m % rpg) + roff, n]
    // XCD-aware tile order (filled by the launcher): 0 = 2-D grid as is; 1 / 2 = 1-D grid, consecutive tiles of one
    // XCD walk M first / N first, so the tiles an XCD's L2 serves share operand panels.
    int swz, tiles_m, tiles_n;
    float* colsum;  // TN only, lean kernels: colsum[m] = sum_k A[k, m] (the bias gradient that goes with a weight gradient
                    // dW = dY^T X), produced by the n-tile-0 workgroups with one extra MFMA against a ones fragment per k-step
    int epi0;  // split-K: apply bias / residual / rowbias on batch slice 0 only (the other slices are raw partial planes)
    unsigned long long* trace;  // tuning only (exp bit 16): per-workgroup time stamps of the producer / consumer kernel
    int exp;  // tuning experiments (bit 0: no loads after the prologue, bit 1: no MFMA, bit 2: no epilogue); 0 in production
};

int gemm_dispatch(int layout, const GemmArgs& a, int batch, hipStream_t s);
// grid + XCD-aware tile order for bm x bn tiles (fills swz / tiles_m / tiles_n / exp of b); gemm.hip
dim3 plan_grid(GemmArgs& b, int bm, int bn, int batch);
// producer / consumer kernel, tile configurations >= 50 (gemm_pc.hip); K % 64 == 0 (batch > 1: split-K planes)
int launch_gemm_pc(int layout, const GemmArgs& a, int batch, int cfg, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
// swat_gemm_prof_begin / _end: take the next timing slot for a launch of `kind` (see swat_gemm_prof_end) — gemm.hip
bool gemm_prof_take(int kind, double flops, hipEvent_t& ev0, hipEvent_t& ev1);
// 256 x 256 tile, NT / NN, one unbatched product, K % 64 == 0 (gemm_big.hip)
int launch_gemm_big(int layout, const GemmArgs& a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
// up to 64 products of one layout in ONE persistent launch (gemm_pc.hip); every product: batch 1, K % 64 == 0, fp32 or
// bf16 output with 16-B rows (what gemm_group_dispatch checks before it comes here)
int launch_gemm_pc_group(int layout, const GemmArgs* probs, int n, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, int cfg = 0);
// n <= 8 weight gradients dW_i = dY_i^T X_i (TN, fp32 out, optional column sums of dY) as ONE launch of 256 x 128 tiles
// (gemm_wgrad.hip); -> 0 launched, 1 shapes not served (run them one by one), < 0 error
int launch_wgrad_group(const GemmArgs* probs, int n, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, int exp = 0, unsigned long long* trace = nullptr);
// 256 x 128 tiles, loader / consumer waves, persistent over the joint tile list of n <= 8 products of one layout (gemm_xl.hip)
bool gemm_xl_serves(int layout, const GemmArgs& g);
int launch_gemm_xl(int layout, const GemmArgs* probs, int n, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, int exp = 0, unsigned long long* trace = nullptr,
                   const int* tails = nullptr);   // tails[i] (NT / NN): 0 .. 32 rows beyond probs[i].M computed by the launch's tail phase
bool gemm_skinny_serves(int layout, const GemmArgs& a);   // the remainder-row scheme's shape conditions (gemm_skinny.hip)
// AdamW applied to a weight-gradient tile while it is still in the accumulators (torch.optim.AdamW semantics and the arithmetic of
// adamw_kernel, csrc/elementwise.hip — ref utils/optimizers.py:131-146): the gradient is never written; parameter, moments and the
// bf16 shadow are read / written at the arena offset of the product's output (C - gbase).  One learning-rate group.
struct XlAdam {
    int on;
    const float* gbase;           // arena.grad
    float *p, *m, *v;             // arena.flat, exp_avg, exp_avg_sq
    unsigned short* shadow;
    const float *lr, *bc;         // device scalars: lr of the group; {1 - b1^t, 1 - b2^t}
    float b1, b2, eps, wd, gscale;
    const unsigned* guard;        // non-zero: a persistent launch of the step gave up — leave everything as it is
    int stagger;                  // start offset between the four phase groups of workgroups, 100 MHz ticks (gemm_xl.hip)
};
// up to 64 weight gradients (TN) as ONE launch, the product list in device memory (gemm_xl.hip); adam != null: consumed by AdamW in
// the epilogue.  wgrad_all_try: policy + timing slot
int launch_gemm_xl_table(const GemmArgs* probs, int n, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const XlAdam* adam = nullptr);
int wgrad_all_try(const GemmArgs* probs, int n, hipStream_t s, const XlAdam* adam = nullptr);
// AdamW of up to 128 short arena ranges (biases, LayerNorm weights between the matrices) in one launch (elementwise.hip)
int launch_adamw_ranges(const int64_t* begin, const int* len, int n, const XlAdam& ad, hipStream_t s);
// M <= 64 rows (the remainder of a many-row product): one wave per 16 output columns (gemm_skinny.hip); -> 0 / 1 not served / < 0
int launch_gemm_skinny(int layout, const GemmArgs& a, hipStream_t s);
int wgrad_big_try(const GemmArgs* probs, int n, hipStream_t s, bool dry_run = false);   // policy + launch_wgrad_group (gemm.hip); same
                                                                                         // return values; dry_run: only say whether it would launch
// n products of one layout: one grouped launch when every product qualifies, else one launch each
int gemm_group_dispatch(int layout, const GemmArgs* probs, int n, hipStream_t s);

// the transformer blocks of the ViT forward as one persistent launch (vit_mega.hip); -> 0 launched, 1 not eligible, < 0 error
int vit_mega_forward(int B, int T, int d, int H, int mlp, int L, int act, int save, int M, int Mp, const float* params,
                     const unsigned short* shadow, const int64_t* layer_offs, float* xs, float* x1, unsigned short* h1,
                     unsigned short* h2, unsigned short* a, unsigned short* qkv, unsigned short* f, unsigned short* gact, float* mean1,
                     float* rstd1, float* mean2, float* rstd2, hipStream_t s);

// data gradients + LayerNorm / attention backward of all blocks as one persistent launch (vit_mega.hip); same return values
bool vit_mega_backward_fits(int B, int T, int d, int H, int mlp);   // shape only: the workspace carve
bool vit_mega_backward_shape(int B, int T, int d, int H, int mlp, int L);   // shape + modes: will it run
int64_t vit_mega_backward_lnp_bytes(int B, int d);
int vit_mega_backward(int B, int T, int d, int H, int mlp, int L, int act, int M, int Mp, const float* params, const unsigned short* shadow,
                      const int64_t* layer_offs, const float* xs, const float* x1, const unsigned short* qkv, const unsigned short* f,
                      const float* mean1, const float* rstd1, const float* mean2, const float* rstd2, float* dx, unsigned short* dxb,
                      unsigned short* dfb, unsigned short* dxb2, unsigned short* dqkv, float* dh2, float* dh1, float* lnp1, float* lnp2,
                      hipStream_t s);

// fixed-order reduction of `nblk` blocks [3][d] of LayerNorm-backward partial sums (elementwise.hip)
int ln_bwd_finish_blocks(const float* part, int nblk, int d, float* dgamma, float* dbeta, float* dcolsum, hipStream_t s);
// ... of `nsets` independent sets (same nblk, d) in one launch per 64 sets
int ln_bwd_finish_sets(const float* const* part, float* const* dgamma, float* const* dbeta, float* const* dcolsum, int nsets, int nblk, int d,
                       hipStream_t s);

// swat_ce_mixed's two launches (tail.hip): per-sample losses and d logits; their batch means (loss, d logit_scale, running sum)
int launch_ce_rows(const float* z, const float* logit_scale_dev, const int64_t* ya, const int64_t* yb, float lam, float* per_sample_ws, float* dz,
                   float* logits_out, int B, int C, hipStream_t s);
int launch_ce_finish(const float* per_sample_ws, int B, float* loss, float* dlogit_scale, double* loss_accum, hipStream_t s);
// features -> normalise -> logits -> per-sample mixed CE -> d logits -> d features as one launch, a workgroup per sample (tail.hip)
int launch_head_rows(const float* feats, const float* W, const float* ls, const int64_t* ya, const int64_t* yb, float lam, float* fn, float* nrm,
                     float* z, float* dz, float* per_sample_ws, float* dfeats, int B, int E, int C, hipStream_t s);

// device word that is non-zero while a persistent forward's give-up has not been read by the host (vit_mega.hip); null when no
// persistent forward has run on the current device.  AdamW and the running loss sum skip their writes while it is set.
const unsigned* mega_guard_word();

}  // namespace swat
