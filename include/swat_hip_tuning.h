/* include/swat_hip_tuning.h — measurement and tuning entry points of libswat_hip.so.
 *
 * NOT part of the drop-in boundary (include/swat_hip.h): nothing the reference does maps onto these.  They exist for
 * bench.py's roofline leg (per-launch kernel times), the tile-configuration tests and the tools under tools/.  The
 * ablation switches (swat_gemm_set_exp) make results WRONG on purpose; production code never calls this header.
 */
#ifndef SWAT_HIP_TUNING_H
#define SWAT_HIP_TUNING_H
#include "swat_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Time every GEMM launch with start / stop hipEvents attached to the kernel's own dispatch packet (kernel execution
 *     time, as rocprofv3 reports it).  end() synchronises and fills 16 slots — layout * 3 + {0: 64-row tile,
 *     1: 128- / 256-row tile, 2: producer / consumer kernel} for NT / NN / TN, 9: the persistent forward (csrc/vit_mega.hip:
 *     all blocks in one launch), 10: a block's weight gradients as one launch of 256 x 128 tiles, 11: the 256 x 128 loader /
 *     consumer kernel on NT / NN products, 12: the persistent backward, 13-15: unused — with total milliseconds, launch counts
 *     and flops; returns the launches seen.  The three arrays hold 16 entries. */
int swat_gemm_prof_begin(int max_launches);
int swat_gemm_prof_end(double* ms_by_kind, int64_t* count_by_kind, double* flops_by_kind);
/* Pin one tile configuration (-1 = automatic) — tests and tuning.  14-19, 25-28, 43: LDS-DMA ring (64x64 / 128x128 /
 *     128x64 tiles, 2-8 stages, 4 or 8 waves); 30-34: register ring on whole k-steps; 20-24: register ring with K tails
 *     and batch strides (the fallback of every other number); 50-65: producer / consumer kernel (csrc/gemm_pc.hip:
 *     128x128, 160x128, 96x64, 96x96, 128x64, 64x64 tiles, loader-wave and ring-depth variants); 80: 256 x 256, NT / NN
 *     (csrc/gemm_big.hip). */
int swat_gemm_set_tile(int cfg);
/* Ablation bits (results WRONG): 1 no operand loads after the prologue, 2 no LDS reads / MFMA, 4 no epilogue, 8 no
 *     k-order skew, 16 per-workgroup time stamps into the trace buffer, 32 per-wave epilogue instead of the LDS-staged
 *     one, 64 loader waves read into registers instead of LDS-DMA, 512 never pick the producer / consumer kernel, 2048
 *     (results right) let pick_cfg take the 256 x 256 tile for deep many-row products.  SWAT_GEMM_EXP presets the bits. */
int swat_gemm_set_exp(int bits);
int swat_gemm_set_trace(void* dev_buf); /* exp bit 16: 4 (LDS-DMA kernel) or 8 (producer / consumer) uint64 per workgroup */
int swat_gemm_set_swizzle(int mode);    /* 0 = plain 2-D grid, -1 = XCD-aware tile order (default) */
/* Split-K planes of the deep, narrow products (c_proj forward, c_fc / in_proj dgrads: K = 4d / 3d onto N = d): the
 *     reduction runs as `planes` slices into fp32 planes that the next LayerNorm kernel sums in a fixed order (default 3;
 *     0 or 1 = one launch, no planes).  Deterministic either way; the per-element summation order differs. */
int swat_vit_set_split_k(int planes);
/* Weight gradients of a block through the grouped 256 x 128-tile launch (csrc/gemm_wgrad.hip): 0 = never, 1 = when the joint
 *     tile list is one round of the chip (default), 2 = whenever the shapes allow (tests). */
int swat_gemm_set_wgrad_big(int mode);
/* The transformer blocks of the forward as ONE persistent launch (csrc/vit_mega.hip; ViT-B/32 regime: d = 768, 12 heads of 64,
 *     38 <= T <= 56): 1 = on when the shapes qualify and the batch fills whole rounds of 16 two-image panels (B = 27-32, 59-64;
 *     default), 2 = on for every batch of a qualifying shape (tests), 0 = always op by op.  Same buffers, same rounding points. */
int swat_vit_set_mega(int mode);
/* Roles of the persistent launches' waves (SWAT_VIT_MEGA_ROLES presets it): 2 (default) = the four loader waves move operand bytes
 *     only for the whole launch and carry up to 192 KB of the NEXT product's weight stages in their registers across the epilogue,
 *     hand-off and LayerNorm in between (the four consumer waves do everything else); 1 = the kernels of rounds 3-5, where every
 *     wave takes part in every phase.  Outputs are bit-identical. */
int swat_vit_set_mega_roles(int mode);
/* XCDs the 16 members of a team are spread over under round-robin workgroup placement (SWAT_VIT_MEGA_XSPLIT presets it): 1 = an XCD
 *     hosts two whole teams (rounds 3-5); 2 / 4 / 8 = it hosts the same 8 / 4 / 2 members of 4 / 8 / 16 teams, i.e. sees a half / a
 *     quarter / an eighth of every weight matrix.  Default 4.  Placement is for speed only: outputs do not depend on it. */
int swat_vit_set_mega_xsplit(int xcds);
/* Synchronises the device, returns and clears the status word of the persistent forwards since the last call: 0 = every team
 *     hand-off completed; otherwise bit 30 is set and the low bits name the hand-off a workgroup gave up on (0.5 s). */
int swat_vit_mega_status(void);
/* No synchronisation: 0, or the 1-based launch number (swat_vit_mega_launches) of the first persistent forward that gave up on a
 *     hand-off since the last swat_vit_mega_status.  While that status is non-zero every swat_adamw launch and the running loss
 *     sum of swat_ce_mixed on the device leave their outputs untouched (the step's gradients are void), so the host can re-run
 *     the steps from that launch on (swat_amd/step.py) once it has noticed. */
int swat_vit_mega_poll(void);
/* Persistent forwards launched on the current device so far. */
int swat_vit_mega_launches(void);
/* Tests: in each of the next `launches` persistent forwards one team member skips its arrival at hand-off number `handoff`
 *     (1-based, 6 per block) and the spin bound is 2 ms: the launch gives up as if that member were not resident. */
int swat_vit_mega_inject_failure(int launches, int handoff);
/* Per-workgroup phase stamps of the next persistent forwards (64 x uint64 per workgroup, 100 MHz ticks); null = off. */
int swat_vit_mega_trace(void* device_buffer);
/* The data-gradient chain of the blocks' backward (swat_vit_backward with stage_hi >= L, stage_lo <= 1) as one persistent launch:
 * 0 = op by op, 1 = persistent when eligible (default; env SWAT_VIT_MEGA_BWD).  _inject_failure / _trace: as the forward's. */
int swat_vit_set_mega_backward(int mode);
/* swat_head_loss with gradients: 1 (default; env SWAT_HEAD_ROWS) the sample-local path (normalise, logits, per-sample CE, d logits,
 *     d features) as ONE launch with a workgroup per sample, 0 the five separate launches. */
int swat_head_set_fused(int on);
int swat_vit_mega_backward_inject_failure(int launches, int handoff);
int swat_vit_mega_backward_launches(void);
/* 1 when swat_vit_backward(dims, .., stage_hi >= L, stage_lo <= 1) would run the blocks as the persistent launch (shape and modes;
 * the caller then makes that ONE call for all blocks instead of one per stage). */
int swat_vit_backward_blocks_fused(const int32_t* dims);
int swat_vit_mega_backward_trace(void* device_buffer);

/* Candidate epilogue of the dense top-k's shadow scan (csrc/retrieval_bf16.hip, A/B runs).  Both forms read the lane's thresholds once
 *     per block and keep the records in a per-wave LDS buffer until the scan is over; they differ in how a hit becomes a record:
 *     2 = a group of 4 x 64 scores is four compares into scalar masks, a group with a hit runs vector code under exec masks per row;
 *     3 = a group is one max + compare, a group with a hit parks its hit lanes (four scores + a tag) in LDS and one rolled loop
 *     after the groups converts them (default; env SWAT_DENSE_EPI).  Same set of records; the order inside a group differs. */
int swat_dense_set_epilogue(int form);
/*     where a step of that scan issues its row loads: 1 = one between every few MFMAs of the step (default), 0 = all at the step's
 *     start (measured 2 % slower; env SWAT_DENSE_SPREAD).  Same records. */
int swat_dense_set_spread(int on);
/*     with env SWAT_DENSE_PROF=1 the scan runs its cycle-accounting build; this copies the last launch's per-wave sums of shader
 *     cycles to host_out[256 workgroups][8 waves][5] = {counted wait, barrier, MFMA phase, candidate epilogue, whole kernel}
 *     (device synchronised first).  tools/prof_dense_waves.py. */
int swat_dense_debug_profile(unsigned long long* host_out);

/* swat_adamw's launch shape: 0 = wide (up to 4096 blocks, the fastest alone), n > 0 = n blocks of 256 threads handling four
 *     16-byte quads per array and trip (n = 256: one block per CU, 4 of its 32 wave slots) so that GEMM workgroups on another
 *     stream share the CUs with it.  Same arithmetic per element.  Also SWAT_ADAMW_BLOCKS in the environment. */
int swat_adamw_set_blocks(int blocks);

#ifdef __cplusplus
}
#endif
#endif
