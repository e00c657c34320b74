// swat_amd/csrc/vit_mega.hip — the transformer blocks of the ViT forward as ONE persistent launch (ViT-B/32 regime:
// T <= 56 tokens, d = 768, head width 64, B*T rows in the low thousands).
//
// ref: the 12 x [ln_1, in_proj, attention, out_proj, ln_2, c_fc, act, c_proj] of open_clip's VisionTransformer.forward
// behind model.encode_image (call sites utils/training.py:332,1474, testing.py:430; arithmetic SURVEY App. A).
//
// Why.  Launched op by op (csrc/vit.hip) a block is seven launches of 5-18 us each, of which 6-7 us per GEMM launch
// are fixed costs (dispatch, workgroup entry, first operand bytes, epilogue tail) — at B = 32 they are
// 40 % of the step.  Every op of the forward is local to an image (attention) or to a row (LayerNorm, the linear
// layers), so a PANEL of images can run the whole tower without ever waiting for another panel.  The launch is 256
// workgroups, one per CU, in TEAMS of 16: a team owns one panel (P images, P*T <= 112 rows = seven 16-row MFMA fragments)
// at a time and walks it through all layers.  Inside a layer the team alternates six phases with a team-wide
// hand-off between them (nothing is chip-wide):
//     LN1 : member i normalises rows 7i .. 7i+6 of the panel                               -> h1 (bf16)
//     QKV : member h < H multiplies the panel by head h's 3 x 64 rows of in_proj (112 x 192 x K tile), keeps q, k, v in
//           LDS and runs the attention of its head for the panel's images right there       -> qkv (for backward), a
//     OUT : member i computes columns 48i .. 48i+47 of out_proj, + bias + residual          -> x1 (fp32)
//     LN2 : as LN1                                                                         -> h2
//     FC  : member i computes columns 192i .. of c_fc, + bias, activation                   -> f, g = act(f)
//     PRJ : member i computes columns 48i .. of c_proj, + bias + residual                   -> x (next block's input)
// The GEMM phases use the producer / consumer structure of csrc/gemm_pc.hip (loader waves issue LDS-DMA into a ring with
// counted vmcnt, consumer waves read fragments and issue v_mfma_f32_16x16x32_bf16, one raw s_barrier per 64-deep
// k-step); what bounds them is the same per-CU L2 -> LDS rate, but a layer now pays six ~2 us hand-offs instead of
// seven kernel boundaries + launch fixed costs.
//
// Hand-off protocol (MI355X guide, Guideline 16 recipe R1, counter form; placement-independent — the team <-> XCD
// mapping below — mg_place, round 6: a team's members spread over four XCDs, so that an XCD's L2 sees a quarter of every weight
// matrix — is for speed only): every byte another workgroup will read is stored write-through (sc1); each storing
// wave drains its stores (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane adds 1 to the team's arrival counter (agent
// scope) and polls it relaxed until all 16 members of the phase have arrived; then ONE agent-scope acquire
// (buffer_inv sc1), its wait, workgroup barrier, and only then the next phase's loads.  Counters are zeroed by a
// hipMemsetAsync in front of every launch; every spin is bounded (give-up code in the status block).
//
// Round 6 (DESIGN.md sections 4 and 8): what bounds these kernels was measured — the k-loops by the CU's vector-memory byte rate
// (~50 GB/s per CU), the chain between them by memory round trips that cannot share the pipe with bulk traffic.  Live from that round:
// the placement above, and epilogues without a global load (mg_epi_residual_keep: the residual slice stays in the thread's registers).
// Kept as a tested switch, not the default: vit_fwd_mega2_kernel (fixed wave roles, weight stages carried in loader registers).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "gemm_device.hpp"
#include "attention.hpp"
#include "../../include/swat_hip.h"
#include "../../include/swat_hip_tuning.h"

namespace swat {

constexpr int MG_ROWS = 112;      // panel rows held by one team (seven 16-row fragments)
constexpr int MG_FM = 7;
constexpr int MG_TEAM = 16;       // workgroups per team
constexpr int MG_THREADS = 512;   // 4 consumer + 4 loader waves
constexpr int MG_NC = 4, MG_NL = 4;
constexpr int MG_LDS = 156 * 1024;

struct MegaFwd {
    // shapes
    int B, T, d, H, mlp, L, act, save, M, Mp;
    int P;          // images per panel
    int npanels, nteams;
    int xsplit;     // XCDs a team is spread over (mg_place)
    float eps;
    // parameters: fp32 master + bf16 shadow at the same element offsets; layer l adds l * layer_stride
    const float* params;
    const unsigned short* shadow;
    int64_t lo[12];         // ln1_w ln1_b qkv_w qkv_b out_w out_b ln2_w ln2_b fc_w fc_b proj_w proj_b of layer 0
    int64_t layer_stride;
    // workspace (csrc/vit.hip carve): per-layer strides are 0 when the forward does not save activations
    float* xs;              // [(L+1) or 2][M, d]
    float* x1;
    unsigned short *h1, *h2, *a, *qkv, *f, *gact;
    float *mean1, *rstd1, *mean2, *rstd2;
    // hand-off state (zeroed before every launch)
    unsigned* ctr;          // [nteams][32] : word 0 of each 128-B line = arrivals
    unsigned* status;       // [0] = first give-up code (0 = ok); sticky until the host reads it (swat_vit_mega_status): the
                            // AdamW launches and the loss accumulation of the process skip their work while it is non-zero
    unsigned* host_flag;    // pinned host word: launch number of the first launch that gave up (the host polls it without a sync)
    unsigned launch_no;     // 1-based number of this launch on its device
    unsigned long long spin_ticks;  // bound of a hand-off wait (100 MHz ticks)
    unsigned ctr_base;      // the team counters' value when this launch starts: they only grow (no memset in front of a launch);
                            // hand-off number q of the launch completes at ctr_base + 16 q
    int fail_sync;          // tuning (swat_vit_mega_inject_failure): member 1 of team 0 skips its arrival at hand-off number
                            // fail_sync of the launch (1-based); 0 in production
    unsigned long long* trace;  // tuning: per workgroup phase stamps (null in production)
};

enum { LN1W = 0, LN1B, QKVW, QKVB, OUTW, OUTB, LN2W, LN2B, FCW, FCB, PRW, PRB };

// ---- write-through (sc1) stores / loads of bytes that cross workgroups ---------------------------------------------
__device__ __forceinline__ void st_sc1_16(void* p, u32x4_t v) {
    // one 16-B store with sc1: inline asm (a raw-buffer descriptor would cap the workspace at 4 GiB); the trailing s_nop keeps
    // the data registers intact until the store has read them (guide 5.7 item 1)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v));
}
__device__ __forceinline__ void st_sc1_8(void* p, unsigned lo, unsigned hi) {
    const unsigned long long x = ((unsigned long long)hi << 32) | lo;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mg_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned long long mg_clock() { return wall_clock64(); }  // 100 MHz

// ---- the weight slice of a GEMM phase: tile row n of the phase -> row of W ([N][K] row-major, K contiguous) -------------------
// (backward, data gradients dX = dY W: the reduction runs down the ROWS of W — W points at the member's first column, ldw is
// still the row stride, and the slice [64 k][BN] of a stage is laid out in LDS as gemm_device.hpp's stp_off image)
struct MegaW {
    const unsigned short* W;   // null: this member has no product in the phase
    int ldw, nk;               // row stride (elements), 64-deep k-steps
    int row0, qkv_h, d;        // qkv_h >= 0: in_proj rows of head qkv_h (n >> 6 selects q / k / v), else rows row0 + n
    __device__ __forceinline__ int row(int n) const { return qkv_h >= 0 ? (n >> 6) * d + 64 * qkv_h + (n & 63) : row0 + n; }
};

__device__ __forceinline__ void mg_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// team-wide hand-off: all members' phase outputs are complete and visible when this returns (see the file header).
// `early(lw)` runs on the four loader waves between the two workgroup barriers — while lane 0 waits for the team they already
// stream the first stages of the NEXT phase's weight slice into the LDS ring (weights depend on no hand-off).  The second
// barrier is a raw s_barrier: __syncthreads() would drain those LDS-DMAs (vmcnt(0)).
// lane 0 of the workgroup: arrive at the team's counter and wait (bounded) until every member has
template <class Params>
__device__ __forceinline__ void team_arrive(const Params& p, int team, int member, unsigned target, unsigned long long* tr) {
    if (tr) tr[1] = mg_clock();                   // stores drained, workgroup together
    unsigned* c = p.ctr + team * 32;
    const bool skip = p.fail_sync && target == (unsigned)p.fail_sync * MG_TEAM && team == 0 && member == 1;
    target += p.ctr_base;
    if (!skip) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = mg_clock();
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        // a hand-off that has already been given up on anywhere (this launch or an earlier one the host has not seen yet):
        // the launch's values are void, do not wait 0.5 s at each of its 72 hand-offs
        if ((++spins & 63u) == 0 && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        if (mg_clock() - t0 > p.spin_ticks) {  // a member never arrived (not resident / faulted): give up, flag it
            __hip_atomic_store(p.status, 0x80000000u | (target - p.ctr_base), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned expect = 0;
            __hip_atomic_compare_exchange_strong(p.host_flag, &expect, p.launch_no, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    if (tr) tr[2] = mg_clock();                   // every member has arrived
    // The acquire drops lines this CU's L1 (and the L2's non-local lines) may hold from BEFORE they were rewritten.  With
    // save = 1 every handed-off buffer is per layer: each line is written once in a launch and no workgroup loads it before
    // the hand-off that publishes it (the launch itself starts with invalidated caches), so there is nothing stale to
    // drop and the ~1 us invalidate is skipped.  save = 0 re-uses the buffers layer after layer: the acquire stays.
    if (!p.save) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (tr) tr[3] = mg_clock();                   // acquire complete
}

template <class Params, class Early>
__device__ __forceinline__ void team_sync(const Params& p, int team, int member, unsigned target, Early&& early, unsigned long long* tr = nullptr) {
    if (tr && threadIdx.x == 0) tr[0] = mg_clock();   // stores issued
    mg_drain();          // every storing wave
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) team_arrive(p, team, member, target, tr);
    else if (wave >= MG_NC) early(wave - MG_NC);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mg_barrier();
}

template <int N_>
__device__ __forceinline__ void mg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
template <int IMM>
__device__ __forceinline__ bf16x8_t mg_lds_b128(unsigned addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
template <int N_>
__device__ __forceinline__ void mg_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_ > 15 ? 15 : N_) : "memory"); }
__device__ __forceinline__ void mg_pin(bf16x8_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ bf16x8_t mg_lds_tr(unsigned lo_addr, unsigned hi_addr) {
    s16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(lo_addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(hi_addr));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
template <int N_, typename F, int... Is>
__device__ __forceinline__ void mg_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N_, typename F>
__device__ __forceinline__ void mg_static_for(F&& f) { mg_static_for_impl<N_>(f, std::make_integer_sequence<int, N_>{}); }

// dynamic s_waitcnt vmcnt(n) (n is wave uniform; the instruction wants an immediate)
__device__ __forceinline__ void mg_wait_vmcnt_n(int n) {
    switch (n) {
#define MG_W(k_) case k_: asm volatile("s_waitcnt vmcnt(" #k_ ")" ::: "memory"); break;
        MG_W(0) MG_W(1) MG_W(2) MG_W(3) MG_W(4) MG_W(5) MG_W(6) MG_W(7) MG_W(8) MG_W(9) MG_W(10) MG_W(11) MG_W(12) MG_W(13) MG_W(14) MG_W(15)
        MG_W(16) MG_W(17) MG_W(18) MG_W(19) MG_W(20) MG_W(21) MG_W(22) MG_W(23) MG_W(24) MG_W(25) MG_W(26) MG_W(27) MG_W(28) MG_W(29) MG_W(30)
        MG_W(31) MG_W(32) MG_W(33) MG_W(34) MG_W(35) MG_W(36) MG_W(37) MG_W(38) MG_W(39) MG_W(40)
#undef MG_W
        default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;   // more may fly than we need to allow: still correct
    }
}

// ---- the loader waves' piece lists.  A stage of the ring = [A: 14 pieces of 8 rows x 128 B | W: BN / 8 pieces]; loader wave lw
// moves A pieces lw, lw + 4, .. and W pieces lw, lw + 4, ..  (k-contiguous image of gemm_device.hpp: kc_off, swizzle applied
// to the per-lane source address).
template <int BN>
struct MgPieces {
    static constexpr int PA = MG_ROWS / 8, PB = BN / 8;
    static constexpr int MAXA = (PA + MG_NL - 1) / MG_NL, MAXW = (PB + MG_NL - 1) / MG_NL;
    static constexpr int A_BYTES = PA * 1024, STAGE = (PA + PB) * 1024;
    __device__ static __forceinline__ int na(int lw) { return (PA - lw + MG_NL - 1) / MG_NL; }
    __device__ static __forceinline__ int nw(int lw) { return (PB - lw + MG_NL - 1) / MG_NL; }
};

// W pieces of stage `st` into ring buffer `buf` (their k offset added by the caller through `cur`)
template <int BN>
__device__ __forceinline__ void mg_issue_w(unsigned char* smem, int buf, const char* cur, const unsigned (&voff)[MgPieces<BN>::MAXW], int lw) {
    using PC = MgPieces<BN>;
#pragma unroll
    for (int i = 0; i < PC::MAXW; ++i)
        if (lw + MG_NL * i < PC::PB)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cur + voff[i]),
                                             (__attribute__((address_space(3))) void*)(smem + buf * PC::STAGE + PC::A_BYTES + (lw + MG_NL * i) * 1024),
                                             16, 0, 0);
}
template <int BN, bool WKC = true>
__device__ __forceinline__ void mg_w_offsets(const MegaW& w, int lw, int lane, unsigned (&voff)[MgPieces<BN>::MAXW]) {
#pragma unroll
    for (int i = 0; i < MgPieces<BN>::MAXW; ++i) {
        const int byte = (lw + MG_NL * i) * 1024 + lane * 16;
        if constexpr (WKC) {
            const int row = byte >> 7, t = ((byte >> 4) & 7) ^ ((row >> 1) & 7);
            voff[i] = (unsigned)(w.row(row < BN ? row : BN - 1) * w.ldw + 8 * t) * 2u;
        } else {
            int k = 0, col = 0;
            if (byte < BN * 128) stp_inv<BN>(byte, k, col);
            voff[i] = (unsigned)(k * w.ldw + col) * 2u;
        }
    }
}
// bytes from one 64-deep k-step of the weight slice to the next
template <bool WKC>
__device__ __forceinline__ int64_t mg_w_step(const MegaW& w) { return WKC ? (int64_t)128 : (int64_t)128 * w.ldw; }
// the part of a GEMM phase that needs no hand-off: W of stages 0 .. NS-2 (run by the loader waves inside team_sync)
template <int BN, int NS, bool WKC = true>
__device__ __forceinline__ void mg_w_prologue(unsigned char* smem, const MegaW& w, int lw) {
    if (!w.W) return;
    unsigned voff[MgPieces<BN>::MAXW];
    mg_w_offsets<BN, WKC>(w, lw, threadIdx.x & 63, voff);
    const char* cur = reinterpret_cast<const char*>(w.W);
    const int64_t step = mg_w_step<WKC>(w);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) mg_issue_w<BN>(smem, st, cur + step * st, voff, lw);
}

constexpr int MG_CT = 64 * MG_NC;   // threads of the consumer waves
#ifndef MG_WREG_N
#define MG_WREG_N 48
#endif
#ifndef MG_TRIGGER
#define MG_TRIGGER 1
#endif
constexpr int MG_WREG = MG_WREG_N;  // 16-B registers of weight bytes a loader wave carries from one product into the next (measurement
                                    // builds: 1 = none).  MG_TRIGGER 0: the register stages are loaded right behind the product's last
                                    // stage; 1: when the consumer waves have ISSUED the epilogue's stores (mg_signal / mg_wait_flag)

template <int BN, int NS, int NK>
struct MgRegStages {   // which 64-deep k-steps of a product's weight slice ride in the loader waves' registers: [S0, S1)
    static constexpr int MAXW = MgPieces<BN>::MAXW;
    static constexpr int F = MG_WREG / MAXW;
    static constexpr int S0 = NS - 1;                          // stages 0 .. NS-2 go into the free ring during the hand-off, as before
    static constexpr int S1 = NK < S0 + F ? NK : S0 + F;
    __host__ __device__ static constexpr bool reg(int s) { return s >= S0 && s < S1; }
};

// The loader program sits inside the loop over the blocks and every address it forms is block-invariant: the compiler would hoist
// them ALL out of the loop (dozens of registers next to the 192 that carry weights -> spills).  An opaque copy of the lane number at
// the top of each step keeps its address arithmetic (a few instructions) inside the step.
__device__ __forceinline__ int mg_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// Pieces of a loader wave, known at compile time once the wave's class is: LO = loader waves 0, 1 / !LO = loader waves 2, 3 (the
// piece lists of a stage are dealt round-robin over the four waves: both waves of a class own the same NUMBER of pieces)
template <int BN, bool LO>
struct MgLoad {
    using PC = MgPieces<BN>;
    static constexpr int L0 = LO ? 0 : 2;
    static constexpr int NA = (PC::PA - L0 + MG_NL - 1) / MG_NL, NW = (PC::PB - L0 + MG_NL - 1) / MG_NL;
    static_assert(NA == (PC::PA - L0 - 1 + MG_NL - 1) / MG_NL && NW == (PC::PB - L0 - 1 + MG_NL - 1) / MG_NL, "both waves of a class own the same pieces");
};

template <int BN, bool LO, bool WKC>
__device__ __forceinline__ void mg_w_offsets2(const MegaW& w, int lw, int lane, unsigned (&voff)[MgLoad<BN, LO>::NW]) {
#pragma unroll
    for (int i = 0; i < MgLoad<BN, LO>::NW; ++i) {
        const int byte = (lw + MG_NL * i) * 1024 + lane * 16;
        if constexpr (WKC) {
            const int row = byte >> 7, t = ((byte >> 4) & 7) ^ ((row >> 1) & 7);
            voff[i] = (unsigned)(w.row(row) * w.ldw + 8 * t) * 2u;
        } else {
            int k = 0, col = 0;
            stp_inv<BN>(byte, k, col);
            voff[i] = (unsigned)(k * w.ldw + col) * 2u;
        }
    }
}

// W pieces of one stage by LDS-DMA into ring buffer `buf`
template <int BN, bool LO>
__device__ __forceinline__ void mg_issue_w2(unsigned char* smem, int buf, const char* cur, const unsigned (&voff)[MgLoad<BN, LO>::NW], int lw) {
    using PC = MgPieces<BN>;
#pragma unroll
    for (int i = 0; i < MgLoad<BN, LO>::NW; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cur + voff[i]),
                                         (__attribute__((address_space(3))) void*)(smem + buf * PC::STAGE + PC::A_BYTES + (lw + MG_NL * i) * 1024), 16, 0, 0);
}

// the part of a GEMM phase that needs no hand-off: W of stages 0 .. NS-2 into the (free) ring, inside the hand-off
template <int BN, int NS, bool LO, bool WKC>
__device__ __forceinline__ void mg_w_prologue2(unsigned char* smem, const MegaW& w, int lw, int lane) {
    if (!w.W) return;
    lane = mg_opaque(lane);
    unsigned voff[MgLoad<BN, LO>::NW];
    mg_w_offsets2<BN, LO, WKC>(w, lw, lane, voff);
    const char* cur = reinterpret_cast<const char*>(w.W);
    const int64_t step = mg_w_step<WKC>(w);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) mg_issue_w2<BN, LO>(smem, st, cur + step * st, voff, lw);
}

template <int OFF>
__device__ __forceinline__ void mg_lds_store16(unsigned addr, const u32x4_t& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2 ; mg-stage\n\ts_nop 1" ::"v"(addr), "v"(v), "n"(OFF) : "memory");   // (s_nop: the data registers may be re-used right behind it)
}


// ---- the consumer waves' half of a GEMM phase: fragments from the ring, MFMA, one raw s_barrier per k-step; then (behind the
// barrier that tells the loaders the ring is free) the fp32 tile to LDS at smem as [112][BN + 4].  The caller's workgroup barrier
// follows.
template <int BN, int CN, int NS, bool WKC>
__device__ __forceinline__ void mg_gemm_consume(unsigned char* smem, int nk, int wave, int lane) {
    constexpr int NC = MG_NC;
    constexpr int FM = MG_FM, WN = BN / CN, FN = WN / 16;
    using PC = MgPieces<BN>;
    constexpr int A_BYTES = PC::A_BYTES, STAGE = PC::STAGE;
    (void)NC;
    const bool active = wave < CN;   // BN = 48: three column fragments, the fourth consumer wave only keeps the barriers
    const int wn0 = (active ? wave : 0) * WN;
    const int r16 = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
    constexpr int NB = WKC ? 1 : 2;   // LDS read instructions per B fragment (k-strided: two transposed 8-B reads)
    unsigned ra[2], rb[2][WKC ? 1 : FN][NB];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        ra[ks] = lds0 + kc_off(r16, 4 * ks + g);
        if constexpr (WKC) rb[ks][0][0] = lds0 + A_BYTES + kc_off(wn0 + r16, 4 * ks + g);
        else {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int k = 32 * ks + 8 * g + (r16 >> 2);
                rb[ks][j][0] = lds0 + A_BYTES + stp_off<BN>(k, wn0 + 16 * j) + 8 * (r16 & 3);
                rb[ks][j][NB - 1] = lds0 + A_BYTES + stp_off<BN>(k + 4, wn0 + 16 * j) + 8 * (r16 & 3);
            }
        }
    }
    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (active) {
        bf16x8_t fa[FM], fb[2][FN];
        constexpr int RB = FM / 2;
        auto read_a = [&](auto I, unsigned so, int ks) {
            constexpr int i = decltype(I)::value;
            fa[i] = mg_lds_b128<2048 * i>(ra[ks] + so);
        };
        auto read_b = [&](auto J, unsigned so, int ks, bf16x8_t (&bfr)[FN]) {
            constexpr int j = decltype(J)::value;
            if constexpr (WKC) bfr[j] = mg_lds_b128<2048 * j>(rb[ks][0][0] + so);
            else bfr[j] = mg_lds_tr(rb[ks][j][0] + so, rb[ks][j][NB - 1] + so);
        };
        constexpr int NRI = FM + FN * NB;
        auto half = [&](auto WITH_READS, bf16x8_t (&bcur)[FN], bf16x8_t (&bnext)[FN], unsigned sn, int nks) {
            mg_static_for<FM>([&](auto I) {
                constexpr int i = decltype(I)::value;
                // reads issued behind row i-1 of THIS half (with reads): a_{i-1} + the B fragments that ride behind row i-1
                if constexpr (i == 0) {
                    mg_wait_lgkm<(FM - RB)>();
#pragma unroll
                    for (int j = 0; j < FN; ++j) mg_pin(bcur[j]);
                } else if constexpr (decltype(WITH_READS)::value) {
                    mg_wait_lgkm<NRI - 1>();
                } else {
                    constexpr int later_b = i + 1 < RB ? FN - (i + 1) * FN / RB : 0;
                    mg_wait_lgkm<(FM - 1 - i) + later_b * NB>();
                }
                mg_pin(fa[i]);
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bcur[j], fa[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (decltype(WITH_READS)::value) {
                    read_a(I, sn, nks);
                    if constexpr (i < RB) {
                        constexpr int j0 = i * FN / RB, j1 = (i + 1) * FN / RB;
                        mg_static_for<j1 - j0>([&](auto Jd) {
                            read_b(std::integral_constant<int, j0 + decltype(Jd)::value>{}, sn, nks, bnext);
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        };
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        mg_barrier();  // barrier 0: stages 0 and 1 have landed
        mg_static_for<FM>([&](auto I) { read_a(I, 0u, 0); });
        mg_static_for<FN>([&](auto J) { read_b(J, 0u, 0, fb[0]); });
        mg_wait_lgkm<0>();
        unsigned so = 0;
        for (int t = 0; t + 1 < nk; ++t) {
            const unsigned sn = so + STAGE == NS * STAGE ? 0u : so + STAGE;
            half(T1{}, fb[0], fb[1], so, 1);
            half(T1{}, fb[1], fb[0], sn, 0);
            mg_barrier();
            so = sn;
        }
        half(T1{}, fb[0], fb[1], so, 1);
        half(T0{}, fb[1], fb[0], 0u, 0);
    } else {
        for (int t = 0; t < nk; ++t) mg_barrier();
    }
    // accumulators -> LDS fp32 tile, behind a barrier: every consumer has finished reading the ring
    // (the loaders are past their last stage barrier: this extra barrier is matched below)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (active) {
        float* cl = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                *reinterpret_cast<f32x4_t*>(cl + (16 * i + r16) * (BN + 4) + wn0 + 16 * j + 4 * g) = acc[i][j];
    }
}

// ---- one GEMM phase: C[112, BN] = A[rows, K] (row-major, lda) x W^T, W rows given by w.row(n), both operands
// k-contiguous.  Structure and LDS image of gemm_pc_kernel<L_NT> (one tile, FM = 7 row fragments for every consumer
// wave, CN consumer waves side by side in n); the W pieces of the first NS-1 stages were issued by mg_w_prologue before the
// hand-off completed.  On return the fp32 tile sits in LDS at smem as [112][BN + 4] and the workgroup has passed a barrier
// behind it.
template <int BN, int CN, int NS, bool WKC = true>
__device__ __forceinline__ void mg_gemm_nt(unsigned char* smem, const unsigned short* A, int lda, int rows, const MegaW& w) {
    constexpr int NL = MG_NL, NC = MG_NC;
    constexpr int FM = MG_FM, WN = BN / CN, FN = WN / 16;
    static_assert(BN % (16 * CN) == 0 && CN <= NC, "wave tiles are whole fragments");
    using PC = MgPieces<BN>;
    constexpr int A_BYTES = PC::A_BYTES, STAGE = PC::STAGE;
    static_assert(NS >= 4 && NS * STAGE <= MG_LDS && MG_ROWS * (BN + 4) * 4 <= NS * STAGE, "LDS ring");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = w.nk;   // >= NS (d = 768: 12 or 48 k-steps)

    if (wave >= NC) {
        // ================================ loader waves ================================
        const int lw = wave - NC;
        const int nA = PC::na(lw), nW = PC::nw(lw), per = nA + nW;   // pieces of this wave per stage
        unsigned voffA[PC::MAXA], voffW[PC::MAXW];
#pragma unroll
        for (int i = 0; i < PC::MAXA; ++i) {
            const int byte = (lw + NL * i) * 1024 + lane * 16;
            const int row = byte >> 7, t = ((byte >> 4) & 7) ^ ((row >> 1) & 7);
            const int gr = row < rows ? row : rows - 1;
            voffA[i] = (unsigned)(gr * lda + 8 * t) * 2u;
        }
        mg_w_offsets<BN, WKC>(w, lw, lane, voffW);
        const char* curA = reinterpret_cast<const char*>(A);
        const char* curW = reinterpret_cast<const char*>(w.W);
        const int64_t stepW = mg_w_step<WKC>(w);
        auto issue_a = [&](int buf, int st) {
#pragma unroll
            for (int i = 0; i < PC::MAXA; ++i)
                if (lw + NL * i < PC::PA)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(curA + 128 * st + voffA[i]),
                                                     (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (lw + NL * i) * 1024), 16, 0, 0);
        };
        // in flight, oldest first: W(0 .. NS-2) (issued inside the hand-off), now A(0 .. NS-2), then whole stages
#pragma unroll
        for (int st = 0; st < NS - 1; ++st) issue_a(st, st);
        int t = 0, buf = NS - 1;
        for (; t + NS - 1 < nk; ++t) {
            // barrier t: stages <= t + 1 have landed.  Still allowed to fly: the A pieces of the prologue stages t + 2 .. NS-2
            // and the t whole stages issued in this loop so far (at most NS-3 of them)
            const int pro = NS - 3 - t;
            mg_wait_vmcnt_n(pro > 0 ? pro * nA + t * per : (NS - 3) * per);
            mg_barrier();
            issue_a(buf, t + NS - 1);
            mg_issue_w<BN>(smem, buf, curW + stepW * (t + NS - 1), voffW, lw);
            buf = buf + 1 == NS ? 0 : buf + 1;
        }
        for (; t < nk; ++t) {   // tail: everything is issued; stages t + 2 .. nk - 1 may still fly
            const int fly = nk - 2 - t;
            mg_wait_vmcnt_n(fly > 0 ? (fly < NS - 3 ? fly : NS - 3) * per : 0);
            mg_barrier();
        }
    } else {
        mg_gemm_consume<BN, CN, NS, WKC>(smem, nk, wave, lane);
    }
    if (wave >= NC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // matches the consumers' "ring is free" barrier
    }
    __syncthreads();                    // the fp32 tile is complete
}

// ---- LayerNorm of the member's rows (one wave per row): x fp32 -> y bf16 (sc1), mean / rstd ------------------------
__device__ __forceinline__ void mg_layernorm(const MegaFwd& p, const float* x, const float* gamma, const float* beta,
                                             unsigned short* y, float* mean, float* rstd, int row0, int rows_valid, int member) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = 7 * member + wave;     // 16 members x 7 rows = 112
    if (wave < 7 && r < rows_valid) {
        const int d = p.d;               // d = 768: 3 float4 per lane
        const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)(row0 + r) * d);
        float4 v[3], gm[3], bt[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) v[i] = xr[lane + 64 * i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {    // every load of the row's pass is in flight before the first reduction
            gm[i] = reinterpret_cast<const float4*>(gamma)[lane + 64 * i];
            bt[i] = reinterpret_cast<const float4*>(beta)[lane + 64 * i];
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
        const float mu = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, dd = v[i].w - mu;
            q += a * a + b * b + cc * cc + dd * dd;
        }
        const float var = wave_sum(q) / (float)d;
        const float rs = rsqrtf(var + p.eps);
        if (lane == 0) {
            mean[row0 + r] = mu;
            rstd[row0 + r] = rs;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = lane + 64 * i;
            const float o0 = (v[i].x - mu) * rs * gm[i].x + bt[i].x, o1 = (v[i].y - mu) * rs * gm[i].y + bt[i].y;
            const float o2 = (v[i].z - mu) * rs * gm[i].z + bt[i].z, o3 = (v[i].w - mu) * rs * gm[i].w + bt[i].w;
            st_sc1_8(y + (int64_t)(row0 + r) * d + 4 * c, pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
        }
    }
}

// ---- epilogue passes over the fp32 tile cl[112][BN + 4]: a thread keeps ONE column chunk (bias loaded once) and walks
// the rows; every global load of the pass is issued before the first store (loads and stores share vmcnt, in order).
// BN = 48, fp32 out: 12 chunks of 4 columns x 42 rows per pass (504 threads), 3 passes: out = cl + bias + residual (sc1 store)
__device__ __forceinline__ void mg_epi_residual(const float* cl, const float* bias, const float* res, float* out, int64_t ld,
                                                int rows, int tid) {
    const int cc = tid % 12, r0 = tid / 12;
    if (r0 >= 42) return;
    const float4 b0 = *reinterpret_cast<const float4*>(bias + 4 * cc);
    float4 rr[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int row = r0 + 42 * it;
        if (row < rows) rr[it] = *reinterpret_cast<const float4*>(res + (int64_t)row * ld + 4 * cc);
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int row = r0 + 42 * it;
        if (row >= rows) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(cl + row * 52 + 4 * cc);
        u32x4_t v;
        v[0] = __float_as_uint(c0.x + b0.x + rr[it].x); v[1] = __float_as_uint(c0.y + b0.y + rr[it].y);
        v[2] = __float_as_uint(c0.z + b0.z + rr[it].z); v[3] = __float_as_uint(c0.w + b0.w + rr[it].w);
        st_sc1_16(out + (int64_t)row * ld + 4 * cc, v);
    }
}
// BN = 192: 24 chunks of 8 columns x 21 rows per pass (504 threads), 6 passes; -> the chunk with bias added, as 8 bf16
template <class F>
__device__ __forceinline__ void mg_epi_bf16(const float* cl, const float* b0p, const float* b1p, int tid, F&& emit) {
    const int cc = tid % 24, r0 = tid / 24;
    if (r0 >= 21) return;
    const float4 b0 = *reinterpret_cast<const float4*>(b0p), b1 = *reinterpret_cast<const float4*>(b1p);
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int row = r0 + 21 * it;
        if (row >= MG_ROWS) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(cl + row * 196 + 8 * cc);
        const float4 c1 = *reinterpret_cast<const float4*>(cl + row * 196 + 8 * cc + 4);
        u32x4_t pk;
        pk[0] = pack_bf16x2(c0.x + b0.x, c0.y + b0.y); pk[1] = pack_bf16x2(c0.z + b0.z, c0.w + b0.w);
        pk[2] = pack_bf16x2(c1.x + b1.x, c1.y + b1.y); pk[3] = pack_bf16x2(c1.z + b1.z, c1.w + b1.w);
        emit(row, pk);
    }
}

// Round 6: the same two passes with every global LOAD taken off the chain between a product's last MFMA and its hand-off.
//  * The residual a thread adds in the out_proj / c_proj epilogue is the value the SAME thread stored in the previous c_proj / out_proj
//    epilogue (a member owns its 48 columns of the residual stream for the whole launch, and both passes map threads to (row, chunk)
//    alike): it stays in 12 registers instead of being stored write-through and loaded back 3 hand-offs later (~1 us per epilogue).
//  * The bias chunks are loaded in front of the product's k-loop.
__device__ __forceinline__ void mg_epi_residual_keep(const float* cl, const float4 b0, float4 (&keep)[3], float* out, int64_t ld, int rows, int tid) {
    const int cc = tid % 12, r0 = tid / 12;
    if (r0 >= 42) return;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int row = r0 + 42 * it;
        if (row >= rows) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(cl + row * 52 + 4 * cc);
        const float4 o = make_float4(c0.x + b0.x + keep[it].x, c0.y + b0.y + keep[it].y, c0.z + b0.z + keep[it].z, c0.w + b0.w + keep[it].w);
        keep[it] = o;
        u32x4_t v;
        v[0] = __float_as_uint(o.x); v[1] = __float_as_uint(o.y); v[2] = __float_as_uint(o.z); v[3] = __float_as_uint(o.w);
        st_sc1_16(out + (int64_t)row * ld + 4 * cc, v);
    }
}
// the thread's slice of the residual stream at a panel's start (the stem wrote it): rows r0 + 42 it, columns 4 cc .. of the member's 48
__device__ __forceinline__ void mg_residual_load(const float* res, int64_t ld, int rows, int tid, float4 (&keep)[3]) {
    const int cc = tid % 12, r0 = tid / 12;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int row = r0 + 42 * it;
        keep[it] = (r0 < 42 && row < rows) ? *reinterpret_cast<const float4*>(res + (int64_t)row * ld + 4 * cc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <class F>
__device__ __forceinline__ void mg_epi_bf16_b(const float* cl, const float4 b0, const float4 b1, int tid, F&& emit) {
    const int cc = tid % 24, r0 = tid / 24;
    if (r0 >= 21) return;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int row = r0 + 21 * it;
        if (row >= MG_ROWS) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(cl + row * 196 + 8 * cc);
        const float4 c1 = *reinterpret_cast<const float4*>(cl + row * 196 + 8 * cc + 4);
        u32x4_t pk;
        pk[0] = pack_bf16x2(c0.x + b0.x, c0.y + b0.y); pk[1] = pack_bf16x2(c0.z + b0.z, c0.w + b0.w);
        pk[2] = pack_bf16x2(c1.x + b1.x, c1.y + b1.y); pk[3] = pack_bf16x2(c1.z + b1.z, c1.w + b1.w);
        emit(row, pk);
    }
}

// ---- attention of one (image, head) from LDS tiles [128][ATP] of q, k, v (rows = panel rows); wave w of the image's 4 waves
// takes query rows 16w .. 16w+15.  Same arithmetic and rounding points as attn_fwd_small_kernel (csrc/attention.hip).
__device__ __forceinline__ void mg_attention(const unsigned short* qt, const unsigned short* kt, const unsigned short* vt, int r0,
                                             int T, int w, int lane, float scale, unsigned short* out, int64_t ldo) {
    const int r16 = lane & 15, g = lane >> 4;
    const int q = w * 16 + r16, qq = q < T ? q : T - 1;
    abf16x8_t qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = ld_frag(qt + (r0 + qq) * ATP + 32 * ks + 8 * g);
    af32x4_t s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int key = 16 * j + r16, kk = key < T ? key : T - 1;
        s[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(kt + (r0 + kk) * ATP + 32 * ks + 8 * g), qf[ks], s[j], 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = 16 * j + 4 * g + e;
            s[j][e] = key < T ? s[j][e] * scale : -INFINITY;
            mx = fmaxf(mx, s[j][e]);
        }
    mx = xor_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = 16 * j + 4 * g + e;
            s[j][e] = key < T ? __expf(s[j][e] - mx) : 0.f;
            sum += s[j][e];
        }
    const float inv = 1.0f / xor_sum(sum);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[j][e] = bf2f(f2bf(s[j][e] * inv));
    const abf16x8_t pf[2] = {acc_frag(s, 0), acc_frag(s, 1)};
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) {
        af32x4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(vt + r0 * ATP, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * jc, r16), pf[ks], o, 0, 0, 0);
        if (q < T) st_sc1_8(out + (int64_t)q * ldo + 16 * jc + 4 * g, pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
    }
}

// team <-> XCD affinity (speed only: the hand-off protocol is placement-independent).  Blocks b, b + 8, ... share an XCD under
// round-robin placement.  xsplit = 1 (rounds 3-5): an XCD hosts whole teams (two of 16 members) — every member's weight slice is then
// pulled into that XCD's L2 by two workgroups, i.e. once per XCD from the Infinity Cache (8 copies of the tower's weights per pass).
// xsplit = 2 / 4 / 8: a team's members are spread over that many XCDs, member groups of 8 / 4 / 2 — an XCD then hosts the SAME member
// group of 4 / 8 / 16 teams and sees only a half / a quarter / an eighth of every weight matrix, read by 4 / 8 / 16 workgroups: a
// quarter of a block's weights (3.5 MB) fits the XCD's 4 MB L2, so seven of eight readers hit it.  The panels, written through to
// memory anyway, are pulled once per XCD they are read on instead of once: at 8 XCDs per team that costs more than the weights gain.
// Default 4 (round 6).
__device__ __forceinline__ void mg_place(int nteams, int xsplit, int b, int& team, int& member) {
    if (nteams % 8 == 0 && (xsplit == 2 || xsplit == 4 || xsplit == 8)) {
        const int x = b & 7, s = b >> 3;              // XCD, slot on it (0 .. 2 nteams / 8 * 16 / ... )
        const int per = MG_TEAM / xsplit;             // members of a team on one XCD
        const int tg = x / xsplit, mg = x % xsplit;   // XCD group (hosts nteams * xsplit / 8 teams), member group inside it
        team = tg * (nteams * xsplit / 8) + s / per;
        // (member groups are CONTIGUOUS: members 4 g .. 4 g + 3 on one XCD.  Interleaved groups — g, g + 4, g + 8, g + 12, which would
        // spread the four members without a head over four XCDs — measured slower: forward +7 us, backward +40 us, whose FC' / IN'
        // members 4 g .. 4 g + 3 share one K-quarter of the panel)
        member = mg * per + s % per;
    } else if (nteams % 8 == 0) {
        const int q = nteams / 8, s = b >> 3;
        team = (b & 7) * q + s / MG_TEAM;
        member = s % MG_TEAM;
    } else {
        team = b / MG_TEAM;
        member = b % MG_TEAM;
    }
}

#define MG_STAMP(i)                                                                                   \
    do {                                                                                              \
        if (p.trace && threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 64 + ((i) & 63)] = mg_clock();  \
    } while (0)

// MG_FWD_VGPR (a build-time knob): arch-VGPR budget of the persistent forward.  NOTE the toolchain's quirk on gfx90a+ (unified
// VGPR / AGPR file): the value `amdgpu_num_vgpr` takes is HALF the allocation it results in — 112 gives a 224-register kernel, 224 is
// out of range and silently ignored (rounds 3-5 concluded the attribute had no effect).
#ifndef MG_FWD_VGPR
#define MG_FWD_VGPR 128
#endif
__global__ __launch_bounds__(MG_THREADS) __attribute__((amdgpu_num_vgpr(MG_FWD_VGPR))) void vit_fwd_mega_kernel(const MegaFwd p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[MG_LDS];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    // team <-> XCD affinity (speed only): blocks b, b + 8, ... share an XCD under round-robin placement
    int team, member;
    mg_place(p.nteams, p.xsplit, b, team, member);
    const int d = p.d, T = p.T, mlp = p.mlp;
    const int64_t M = p.M, Mp = p.Mp;
    const float scale = 1.0f / sqrtf(64.0f);
    unsigned seq = 0;
    int stamp = 0;
    MG_STAMP(stamp++);
    // MG_SYNC(next_, BN_, NS_): hand-off; the loader waves meanwhile start the weight stream of the GEMM phase that follows it
#define MG_SYNC(next_, BN_, NS_, k_)                                                                                  \
    team_sync(p, team, member, ++seq * MG_TEAM, [&](int lw) { mg_w_prologue<BN_, NS_>(smem, next_, lw); },                     \
              (p.trace && l == 0 && panel == team) ? p.trace + (size_t)blockIdx.x * 64 + 8 + 4 * (k_) : nullptr);     \
    if (l == 0) MG_STAMP(stamp++)
#define MG_SYNC0(k_)                                                                                                  \
    team_sync(p, team, member, ++seq * MG_TEAM, [&](int) {},                                                                  \
              (p.trace && l == 0 && panel == team) ? p.trace + (size_t)blockIdx.x * 64 + 8 + 4 * (k_) : nullptr);     \
    if (l == 0) MG_STAMP(stamp++)
#define MG_GEMM_DONE(k_)                                                                                             \
    if (p.trace && l == 0 && panel == team && tid == 0) p.trace[(size_t)blockIdx.x * 64 + 32 + (k_)] = mg_clock()
    for (int panel = team; panel < p.npanels; panel += p.nteams) {
        const int img0 = panel * p.P;
        const int nimg = (p.B - img0) < p.P ? (p.B - img0) : p.P;
        const int row0 = img0 * T, rows = nimg * T;
        // this thread's slice of the residual stream (mg_epi_residual_keep): from the stem at the panel's start, then in registers
        float4 keep[3];
        mg_residual_load(p.xs + (int64_t)row0 * d + 48 * member, d, rows, tid, keep);
        for (int l = 0; l < p.L; ++l) {
            const int64_t po = (int64_t)l * p.layer_stride;
            const int64_t li = p.save ? l : 0;
            const float* xin = p.xs + (p.save ? (int64_t)l : (int64_t)(l & 1)) * M * d;
            float* xout = p.xs + (p.save ? (int64_t)(l + 1) : (int64_t)((l + 1) & 1)) * M * d;
            float* x1 = p.x1 + li * M * d;
            unsigned short* h1 = p.h1 + li * Mp * d;
            unsigned short* h2 = p.h2 + li * Mp * d;
            unsigned short* at = p.a + li * Mp * d;
            unsigned short* qkv = p.qkv + li * M * 3 * d;
            unsigned short* f = p.f + li * M * mlp;
            unsigned short* ga = p.gact + li * Mp * mlp;
            // weight slices of this member in the layer's four products
            const MegaW wQ{member < p.H ? p.shadow + po + p.lo[QKVW] : nullptr, d, d >> 6, 0, member, d};
            const MegaW wO{p.shadow + po + p.lo[OUTW], d, d >> 6, 48 * member, -1, d};
            const MegaW wF{p.shadow + po + p.lo[FCW], d, d >> 6, 192 * member, -1, d};
            const MegaW wP{p.shadow + po + p.lo[PRW], mlp, mlp >> 6, 48 * member, -1, d};
            // ---------------- LN1 ----------------
            mg_layernorm(p, xin, p.params + po + p.lo[LN1W], p.params + po + p.lo[LN1B], h1, p.mean1 + li * M, p.rstd1 + li * M,
                         row0, rows, member);
            MG_SYNC(wQ, 192, 4, 0);
            // ---------------- QKV + attention (member = head) ----------------
            if (member < p.H) {
                const int h = member;
                const int cc = tid % 24, part = cc >> 3, c64 = (cc & 7) * 8;
                const float* bp = p.params + po + p.lo[QKVB] + part * d + 64 * h + c64;
                const float4 bq0 = *reinterpret_cast<const float4*>(bp), bq1 = *reinterpret_cast<const float4*>(bp + 4);   // in front of the k-loop
                mg_gemm_nt<192, 4, 4>(smem, h1 + (int64_t)row0 * d, d, rows, wQ);
                const float* cl = reinterpret_cast<const float*>(smem);
                MG_GEMM_DONE(0);
                unsigned short* tiles = reinterpret_cast<unsigned short*>(smem + MG_ROWS * 196 * 4);  // q | k | v : [128][ATP] each
                {
                    unsigned short* gq = qkv + (int64_t)row0 * 3 * d + part * d + 64 * h + c64;
                    unsigned short* tq = tiles + part * 128 * ATP + c64;
                    mg_epi_bf16_b(cl, bq0, bq1, tid, [&](int row, u32x4_t pk) {
                        *reinterpret_cast<u32x4_t*>(tq + row * ATP) = pk;
                        if (row < rows) *reinterpret_cast<u32x4_t*>(gq + (int64_t)row * 3 * d) = pk;   // backward re-reads it (a later launch)
                    });
                }
                // v rows 112..127 are read (times P = 0) by the last image's key fragments: they must be finite
                for (int c = tid; c < 16 * ATP / 8; c += MG_THREADS)
                    *reinterpret_cast<u32x4_t*>(tiles + 2 * 128 * ATP + MG_ROWS * ATP + c * 8) = (u32x4_t){0u, 0u, 0u, 0u};
                __syncthreads();
                const int wave = tid >> 6, lane = tid & 63;
                for (int im = wave >> 2; im < nimg; im += 2)   // waves 0-3: images 0, 2, ..; waves 4-7: images 1, 3, ..
                    mg_attention(tiles, tiles + 128 * ATP, tiles + 2 * 128 * ATP, im * T, T, wave & 3, lane, scale,
                                 at + (int64_t)(row0 + im * T) * d + 64 * h, d);
            }
            MG_SYNC(wO, 48, 7, 1);
            // ---------------- out_proj + bias + residual -> x1 ----------------
            {
                const int n0 = 48 * member;
                const float4 bo = *reinterpret_cast<const float4*>(p.params + po + p.lo[OUTB] + n0 + 4 * (tid % 12));   // in front of the k-loop
                mg_gemm_nt<48, 3, 7>(smem, at + (int64_t)row0 * d, d, rows, wO);
                MG_GEMM_DONE(1);
                mg_epi_residual_keep(reinterpret_cast<const float*>(smem), bo, keep, x1 + (int64_t)row0 * d + n0, d, rows, tid);   // keep: x -> x1
            }
            MG_SYNC0(2);
            // ---------------- LN2 ----------------
            mg_layernorm(p, x1, p.params + po + p.lo[LN2W], p.params + po + p.lo[LN2B], h2, p.mean2 + li * M, p.rstd2 + li * M,
                         row0, rows, member);
            MG_SYNC(wF, 192, 4, 3);
            // ---------------- c_fc + bias, activation -> f, g ----------------
            {
                const int n0 = 192 * member;
                const int cc = tid % 24;
                const float* bp = p.params + po + p.lo[FCB] + n0 + 8 * cc;
                const float4 bf0 = *reinterpret_cast<const float4*>(bp), bf1 = *reinterpret_cast<const float4*>(bp + 4);   // in front of the k-loop
                mg_gemm_nt<192, 4, 4>(smem, h2 + (int64_t)row0 * d, d, rows, wF);
                MG_GEMM_DONE(2);
                unsigned short* fo = f + (int64_t)row0 * mlp + n0 + 8 * cc;
                unsigned short* go = ga + (int64_t)row0 * mlp + n0 + 8 * cc;
                auto epi = [&](auto ACT) {
                    constexpr int act = decltype(ACT)::value;
                    mg_epi_bf16_b(reinterpret_cast<const float*>(smem), bf0, bf1, tid, [&](int row, u32x4_t pk) {
                        if (row >= rows) return;
                        *reinterpret_cast<u32x4_t*>(fo + (int64_t)row * mlp) = pk;   // pre-activation: only the backward reads it
                        u32x4_t pk2;
#pragma unroll
                        for (int hh = 0; hh < 4; ++hh)
                            pk2[hh] = pack_bf16x2(act_fwd(__uint_as_float(pk[hh] << 16), act), act_fwd(__uint_as_float(pk[hh] & 0xffff0000u), act));
                        st_sc1_16(go + (int64_t)row * mlp, pk2);
                    });
                };
                if (p.act == 1) epi(std::integral_constant<int, 1>{});
                else if (p.act == 2) epi(std::integral_constant<int, 2>{});
                else epi(std::integral_constant<int, 0>{});
            }
            MG_SYNC(wP, 48, 7, 4);
            // ---------------- c_proj + bias + residual -> next block's input ----------------
            {
                const int n0 = 48 * member;
                const float4 bpj = *reinterpret_cast<const float4*>(p.params + po + p.lo[PRB] + n0 + 4 * (tid % 12));   // in front of the k-loop
                mg_gemm_nt<48, 3, 7>(smem, ga + (int64_t)row0 * mlp, mlp, rows, wP);
                MG_GEMM_DONE(3);
                mg_epi_residual_keep(reinterpret_cast<const float*>(smem), bpj, keep, xout + (int64_t)row0 * d + n0, d, rows, tid);   // keep: x1 -> next x
            }
            MG_SYNC0(5);
        }
    }
#undef MG_SYNC0
#undef MG_SYNC
#undef MG_GEMM_DONE
    MG_STAMP(stamp++);
}

// =====================================================================================================================
// The persistent BACKWARD of the blocks: data gradients, LayerNorm backward and attention backward of all L blocks in one launch
// (the weight gradients reduce over every row of the batch — a chip-wide dependency per block — and stay separate launches
// behind it, csrc/gemm_xl.hip).
//
// ref: loss.backward() through the 12 ResidualAttentionBlocks (utils/training.py:1486-1488); op-by-op form: swat_vit_backward
// (csrc/vit.hip), seven launches per block on the main chain, 6-7 us of fixed cost each on M = 1 600 rows.
//
// Same teams, panels and hand-off protocol as the forward above; per block (walked L-1 .. 0) a team alternates
//     PRJ' : member i: dF[:, 192i ..] = (dx W_proj[:, 192i ..]) * act'(F)                                  -> dfb   (bf16)
//     FC'  : member i: dh2[:, 48i ..] = dF W_fc[:, 48i ..]                                                -> dh2   (fp32)
//     LN2' : member i, rows 7i ..: dx += LN'(dh2; x1), bf16 copy, partial sums of d gamma / d beta / colsum -> dx, dxb2, lnp2
//     OUT' : member h < H: da[:, 64h ..] = dx W_out[:, 64h ..] kept in LDS, then the attention backward of head h
//            for the panel's images (S and P recomputed from the saved q, k as in attn_bwd_small_kernel)  -> dqkv  (bf16)
//     IN'  : member i: dh1[:, 48i ..] = dqkv W_in[:, 48i ..]                                              -> dh1   (fp32)
//     LN1' : as LN2' with the block's input                                                               -> dx, dxb of block l-1, lnp1
// The weights are read as stored ([out][in]: the reduction runs down their rows): stages [64 k][BN] in the stp_off image of
// gemm_device.hpp, fragments through ds_read_b64_tr_b16 (csrc/gemm_pc.hip's L_NN form).  Every buffer that crosses workgroups
// is per block (written once per launch: no acquire at the hand-offs, as in the forward with save = 1); the fp32 residual
// gradient dx is updated in place by the member that owns the row — only that workgroup ever touches it.
struct MegaBwd {
    int B, T, d, H, mlp, L, act, save, M, Mp;
    int P, npanels, nteams;
    int xsplit;
    const float* params;
    const unsigned short* shadow;
    int64_t lo[12];
    int64_t layer_stride;
    // saved by the forward
    const float *xs, *x1;
    const unsigned short *qkv, *f;
    const float *mean1, *rstd1, *mean2, *rstd2;
    // gradients
    float* dx;                 // [M, d] residual-stream gradient, in place (in: d(block L-1 output); out: d(block 0 input))
    unsigned short* dxb;       // [L][Mp, d]: slot l = bf16 copy of d(block l output); slot L-1 comes from the ln_post backward
    unsigned short *dfb, *dxb2, *dqkv;   // [L][Mp, mlp | d | 3d]: what the weight gradients reduce over
    float *dh2, *dh1;          // [L][4][M, d]: the four reduction quarters of FC' / IN', summed by the LayerNorm backward
    float *lnp1, *lnp2;        // [L][npanels * 16][3][d] partial sums of the LayerNorm backward (d gamma, d beta, colsum of bf16(dx))
    // hand-off state: as MegaFwd
    unsigned* ctr;
    unsigned* status;
    unsigned* host_flag;
    unsigned launch_no;
    unsigned long long spin_ticks;
    unsigned ctr_base;
    int fail_sync;
    unsigned long long* trace;
};

// Warm the L2 with lines a LATER part of the phase will load (saved activations of the forward: in HBM by now): the four consumer
// waves, which otherwise only wait for the first stages, pull one dword of every 128-B line through the LDS-DMA path into a
// 256-B scratch slot above the operand ring — no registers, nothing waits for it (the next compiler-placed vmcnt(0) of these
// waves is behind the product).  nrows rows of `lines` 128-B lines each.
__device__ __forceinline__ void mg_prefetch(unsigned char* smem, const void* base, int64_t row_stride_bytes, int nrows, int lines) {
    const int tid = threadIdx.x;
    if (tid >= 64 * MG_NC) return;
    unsigned char* slot = smem + MG_LDS - 1024 + (tid >> 6) * 256;
    const int total = nrows * lines;
    for (int i = tid; i < total; i += 64 * MG_NC) {
        const int r = i / lines, c = i - r * lines;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + r * row_stride_bytes + c * 128),
                                         (__attribute__((address_space(3))) void*)slot, 4, 0, 0);
    }
}

// fp32 tile cl[112][BN + 4] (BN = 48) -> out rows, write-through
__device__ __forceinline__ void mg_epi_f32(const float* cl, float* out, int64_t ld, int rows, int tid) {
    const int cc = tid % 12, r0 = tid / 12;
    if (r0 >= 42) return;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int row = r0 + 42 * it;
        if (row >= rows) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(cl + row * 52 + 4 * cc);
        u32x4_t v;
        v[0] = __float_as_uint(c0.x); v[1] = __float_as_uint(c0.y); v[2] = __float_as_uint(c0.z); v[3] = __float_as_uint(c0.w);
        st_sc1_16(out + (int64_t)row * ld + 4 * cc, v);
    }
}

// fp32 tile cl[112][196] (BN = 192) -> out rows, write-through: 48 chunks of 4 columns x 10 rows per pass (480 threads), 12 passes
__device__ __forceinline__ void mg_epi_f32_wide(const float* cl, float* out, int64_t ld, int rows, int tid) {
    const int cc = tid % 48, r0 = tid / 48;
    if (r0 >= 10) return;
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int row = r0 + 10 * it;
        if (row >= rows) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(cl + row * 196 + 4 * cc);
        u32x4_t v;
        v[0] = __float_as_uint(c0.x); v[1] = __float_as_uint(c0.y); v[2] = __float_as_uint(c0.z); v[3] = __float_as_uint(c0.w);
        st_sc1_16(out + (int64_t)row * ld + 4 * cc, v);
    }
}

// cl[112][196] * act'(F) -> dF (bf16, write-through); a thread keeps one 8-column chunk and walks 6 rows, the F loads first
template <int ACT>
__device__ __forceinline__ void mg_epi_actgrad(const float* cl, const unsigned short* f, unsigned short* out, int64_t ld, int rows, int tid) {
    const int cc = tid % 24, r0 = tid / 24;
    if (r0 >= 21) return;
    u32x4_t fv[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int row = r0 + 21 * it;
        if (row < rows) fv[it] = *reinterpret_cast<const u32x4_t*>(f + (int64_t)row * ld + 8 * cc);
    }
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int row = r0 + 21 * it;
        if (row >= rows) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(cl + row * 196 + 8 * cc);
        const float4 c1 = *reinterpret_cast<const float4*>(cl + row * 196 + 8 * cc + 4);
        float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            v[2 * h] *= act_bwd(bf2f((unsigned short)(fv[it][h] & 0xffffu)), ACT);
            v[2 * h + 1] *= act_bwd(bf2f((unsigned short)(fv[it][h] >> 16)), ACT);
        }
        u32x4_t pk;
#pragma unroll
        for (int h = 0; h < 4; ++h) pk[h] = pack_bf16x2(v[2 * h], v[2 * h + 1]);
        st_sc1_16(out + (int64_t)row * ld + 8 * cc, pk);
    }
}

// LayerNorm backward of the member's rows (one wave per row; arithmetic of ln_bwd_kernel, csrc/elementwise.hip):
//   dx[row] += rstd * (dy g - mean(dy g) - xhat mean(dy g xhat));  dxb = bf16(dx) (write-through);
//   part[0 | 1 | 2][d] = this member's sums over its rows of dy xhat | dy | bf16(dx)   (waves added in wave order)
__device__ __forceinline__ void mg_ln_bwd(unsigned char* smem, int d, const float* dy, int64_t plane, const float* x, const float* gamma, const float* mean,
                                          const float* rstd, float* dx, unsigned short* dxb, float* part, int row0, int rows_valid,
                                          int member) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = 7 * member + wave;
    float4 dg[3], db[3], dcs[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[i] = dg[i];
        dcs[i] = dg[i];
    }
    if (wave < 7 && r < rows_valid) {
        const int64_t row = row0 + r;
        const float4* dyr = reinterpret_cast<const float4*>(dy + row * d);
        const float4* xr = reinterpret_cast<const float4*>(x + row * d);
        float4* dxr = reinterpret_cast<float4*>(dx + row * d);
        float4 dyv[3], xh[3], add[3], gm[3], pl[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {   // d = 768: 3 float4 per lane; every load of the row in flight before the first use
            const int c = lane + 64 * i;
            dyv[i] = dyr[c];
#pragma unroll
            for (int sp = 1; sp < 4; ++sp) pl[sp - 1][i] = reinterpret_cast<const float4*>(dy + sp * plane + row * d)[c];
            xh[i] = xr[c];
            add[i] = dxr[c];
            gm[i] = reinterpret_cast<const float4*>(gamma)[c];
        }
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int sp = 1; sp < 4; ++sp) {   // the four reduction quarters of the producing product, fixed order
                const float4 t = pl[sp - 1][i];
                dyv[i].x += t.x; dyv[i].y += t.y; dyv[i].z += t.z; dyv[i].w += t.w;
            }
            const float4 xv = xh[i];
            xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            const float a0 = dyv[i].x * gm[i].x, a1 = dyv[i].y * gm[i].y, a2 = dyv[i].z * gm[i].z, a3 = dyv[i].w * gm[i].w;
            s1 += a0 + a1 + a2 + a3;
            s2 += a0 * xh[i].x + a1 * xh[i].y + a2 * xh[i].z + a3 * xh[i].w;
            dg[i].x += dyv[i].x * xh[i].x; dg[i].y += dyv[i].y * xh[i].y;
            dg[i].z += dyv[i].z * xh[i].z; dg[i].w += dyv[i].w * xh[i].w;
            db[i].x += dyv[i].x; db[i].y += dyv[i].y; db[i].z += dyv[i].z; db[i].w += dyv[i].w;
            dyv[i] = make_float4(a0, a1, a2, a3);
        }
        const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = lane + 64 * i;
            float4 o;
            o.x = rs * (dyv[i].x - m1 - xh[i].x * m2);
            o.y = rs * (dyv[i].y - m1 - xh[i].y * m2);
            o.z = rs * (dyv[i].z - m1 - xh[i].z * m2);
            o.w = rs * (dyv[i].w - m1 - xh[i].w * m2);
            o.x += add[i].x; o.y += add[i].y; o.z += add[i].z; o.w += add[i].w;
            dxr[c] = o;
            const unsigned p0 = pack_bf16x2(o.x, o.y), p1 = pack_bf16x2(o.z, o.w);
            st_sc1_8(dxb + row * d + 4 * c, p0, p1);
            dcs[i].x += __uint_as_float(p0 << 16); dcs[i].y += __uint_as_float(p0 & 0xffff0000u);
            dcs[i].z += __uint_as_float(p1 << 16); dcs[i].w += __uint_as_float(p1 & 0xffff0000u);
        }
    }
    float* sh = reinterpret_cast<float*>(smem);   // [3][7][d]: all three quantities in one pass (64.5 KB)
    if (wave < 7) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                reinterpret_cast<float4*>(sh + (q * 7 + wave) * d)[lane + 64 * i] = q == 0 ? dg[i] : (q == 1 ? db[i] : dcs[i]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * d; c += MG_THREADS) {
        const int q = c / d, cc = c - q * d;
        float v = sh[q * 7 * d + cc];
#pragma unroll
        for (int w = 1; w < 7; ++w) v += sh[(q * 7 + w) * d + cc];
        part[c] = v;
    }
    __syncthreads();
}

// attention backward of one (image, head): 4 waves (w = query / key rows 16w ..), tiles [64][ATP] of this image group in LDS:
// qt, kt, vt (staged from the saved qkv), dot_ (the head's columns of da, bf16), dst / pt (written here).  Arithmetic and rounding
// points of attn_bwd_small_kernel (csrc/attention.hip); `live` = false: a group without an image in this pass (it keeps the
// workgroup barriers, stores nothing).
__device__ __forceinline__ void mg_attn_bwd(int64_t ld, int d, int T, const unsigned short* qt, const unsigned short* kt, const unsigned short* vt,
                                            const unsigned short* dot_, unsigned short* dst, unsigned short* pt, unsigned short* dbase, int w,
                                            int lane, float scale, bool live) {
    const int r16 = lane & 15, g = lane >> 4;
    const int q = w * 16 + r16, qq = q < T ? q : T - 1;
    // S = Q K^T and the row softmax: scores_softmax (csrc/attention.hpp) with the q / k fragments from the tiles (rows >= T hold
    // row T-1: the same clamping)
    af32x4_t p[4];
    {
        abf16x8_t qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = ld_frag(qt + qq * ATP + 32 * ks + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = 16 * j + r16;
            p[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                p[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(kt + key * ATP + 32 * ks + 8 * g), qf[ks], p[j], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = 16 * j + 4 * g + e;
                p[j][e] = key < T ? p[j][e] * scale : -INFINITY;
                mx = fmaxf(mx, p[j][e]);
            }
        mx = xor_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = 16 * j + 4 * g + e;
                p[j][e] = key < T ? __expf(p[j][e] - mx) : 0.f;
                sum += p[j][e];
            }
        const float inv = 1.0f / xor_sum(sum);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) p[j][e] = bf2f(f2bf(p[j][e] * inv));
    }
    abf16x8_t dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) dof[ks] = ld_frag(dot_ + qq * ATP + 32 * ks + 8 * g);
    af32x4_t ds[4];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int key = 16 * j + r16;
        ds[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            ds[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(vt + key * ATP + 32 * ks + 8 * g), dof[ks], ds[j], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) dot += p[j][e] * ds[j][e];
    }
    dot = xor_sum(dot);
    const bool qv = q < T;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = 16 * j + 4 * g + e;
            ds[j][e] = (qv && key < T) ? scale * p[j][e] * (ds[j][e] - dot) : 0.f;
            if (!qv) p[j][e] = 0.f;
        }
        store4(dst + q * ATP + 16 * j + 4 * g, ds[j]);
        store4(pt + q * ATP + 16 * j + 4 * g, p[j]);
    }
    __syncthreads();   // dst / pt complete (both image groups pass here together)
    const abf16x8_t dsf[2] = {acc_frag(ds, 0), acc_frag(ds, 1)};
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) {
        af32x4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(kt, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * jc, r16), dsf[ks], o, 0, 0, 0);
        if (qv && live) st_sc1_8(dbase + (int64_t)q * ld + 16 * jc + 4 * g, pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
    }
    const int key = w * 16 + r16;
    abf16x8_t dst_f[2], pt_f[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        dst_f[ks] = tr_frag(dst, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * w, r16);
        pt_f[ks] = tr_frag(pt, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * w, r16);
    }
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) {
        af32x4_t dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(qt, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * jc, r16), dst_f[ks], dk, 0, 0, 0);
            dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(dot_, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * jc, r16), pt_f[ks], dv, 0, 0, 0);
        }
        if (key < T && live) {
            st_sc1_8(dbase + d + (int64_t)key * ld + 16 * jc + 4 * g, pack_bf16x2(dk[0], dk[1]), pack_bf16x2(dk[2], dk[3]));
            st_sc1_8(dbase + 2 * d + (int64_t)key * ld + 16 * jc + 4 * g, pack_bf16x2(dv[0], dv[1]), pack_bf16x2(dv[2], dv[3]));
        }
    }
}

__global__ __launch_bounds__(MG_THREADS) void vit_bwd_mega_kernel(const MegaBwd p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[MG_LDS];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    int team, member;
    mg_place(p.nteams, p.xsplit, b, team, member);
    const int d = p.d, T = p.T, mlp = p.mlp;
    const int64_t M = p.M, Mp = p.Mp;
    const float scale = 1.0f / sqrtf(64.0f);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    unsigned seq = 0;
    int stamp = 0;
    const int trl = p.L > 2 ? p.L / 2 : p.L - 1;   // the block whose phases the trace stamps
    MG_STAMP(stamp++);
    auto w_prj = [&](int l) {   // dF columns 192 member ..: W_proj [d][mlp], reduction over its d rows
        return MegaW{p.shadow + (int64_t)l * p.layer_stride + p.lo[PRW] + 192 * member, mlp, d >> 6, 0, -1, d};
    };
#define MB_TR(k_) ((p.trace && l == trl && panel == team) ? p.trace + (size_t)blockIdx.x * 64 + 8 + 4 * (k_) : nullptr)
#define MB_SYNC(next_, BN_, NS_, k_)                                                                                                  \
    team_sync(p, team, member, ++seq * MG_TEAM, [&](int lw) { mg_w_prologue<BN_, NS_, false>(smem, next_, lw); }, MB_TR(k_));         \
    if (l == trl) MG_STAMP(stamp++)
#define MB_SYNC0(k_)                                                              \
    team_sync(p, team, member, ++seq * MG_TEAM, [&](int) {}, MB_TR(k_));          \
    if (l == trl) MG_STAMP(stamp++)
#define MB_GEMM_DONE(k_)                                                                                             \
    if (p.trace && l == trl && panel == team && tid == 0) p.trace[(size_t)blockIdx.x * 64 + 32 + (k_)] = mg_clock()
    // the first product's weight stages: no hand-off in front of it
    if (team < p.npanels && wave >= MG_NC) mg_w_prologue<192, 4, false>(smem, w_prj(p.L - 1), wave - MG_NC);
    for (int panel = team; panel < p.npanels; panel += p.nteams) {
        const int img0 = panel * p.P;
        const int nimg = (p.B - img0) < p.P ? (p.B - img0) : p.P;
        const int row0 = img0 * T, rows = nimg * T;
        const int64_t pblk = (int64_t)panel * MG_TEAM + member;   // this member's block of LayerNorm partial sums
        const int64_t npart = (int64_t)p.npanels * MG_TEAM;
        for (int l = p.L - 1; l >= 0; --l) {
            const int64_t po = (int64_t)l * p.layer_stride;
            if (l == trl) MG_STAMP(stamp++);
            const float* xin = p.xs + (int64_t)l * M * d;
            const float* x1 = p.x1 + (int64_t)l * M * d;
            const unsigned short* qkv = p.qkv + (int64_t)l * M * 3 * d;
            const unsigned short* f = p.f + (int64_t)l * M * mlp;
            const unsigned short* dxb_in = p.dxb + (int64_t)l * Mp * d;
            unsigned short* dxb_out = p.dxb + (int64_t)(l > 0 ? l - 1 : p.L) * Mp * d;   // (slot L: a scratch slot nobody reads)
            unsigned short* dfb = p.dfb + (int64_t)l * Mp * mlp;
            unsigned short* dxb2 = p.dxb2 + (int64_t)l * Mp * d;
            unsigned short* dqkv = p.dqkv + (int64_t)l * Mp * 3 * d;
            float* dh2 = p.dh2 + (int64_t)l * 4 * M * d;   // four planes: the reduction quarters of FC' / IN'
            float* dh1 = p.dh1 + (int64_t)l * 4 * M * d;
            const int ng = member & 3, kg = member >> 2;   // FC' / IN': 192 columns x a quarter of the reduction per member
            const MegaW wP = w_prj(l);
            const MegaW wF{p.shadow + po + p.lo[FCW] + (int64_t)kg * (mlp >> 2) * d + 192 * ng, d, mlp >> 8, 0, -1, d};
            const MegaW wO{member < p.H ? p.shadow + po + p.lo[OUTW] + 64 * member : nullptr, d, d >> 6, 0, -1, d};
            const MegaW wI{p.shadow + po + p.lo[QKVW] + (int64_t)kg * ((3 * d) >> 2) * d + 192 * ng, d, (3 * d) >> 8, 0, -1, d};
            // ---------------- PRJ': dF = (dx W_proj) * act'(F) ----------------
            {
                const int n0 = 192 * member;
                mg_prefetch(smem, f + (int64_t)row0 * mlp + n0, (int64_t)mlp * 2, rows, 3);   // the epilogue's F tile
                mg_gemm_nt<192, 4, 4, false>(smem, dxb_in + (int64_t)row0 * d, d, rows, wP);
                MB_GEMM_DONE(0);
                const float* cl = reinterpret_cast<const float*>(smem);
                const unsigned short* fi = f + (int64_t)row0 * mlp + n0;
                unsigned short* fo = dfb + (int64_t)row0 * mlp + n0;
                if (p.act == 1) mg_epi_actgrad<1>(cl, fi, fo, mlp, rows, tid);
                else if (p.act == 2) mg_epi_actgrad<2>(cl, fi, fo, mlp, rows, tid);
                else mg_epi_actgrad<0>(cl, fi, fo, mlp, rows, tid);
            }
            MB_SYNC(wF, 192, 4, 0);
            // ---------------- FC': dh2[kg] = dF[:, quarter kg] W_fc[quarter kg, 192 ng ..] ----------------
            {
                const int r7 = 7 * member, n7 = rows - r7 < 7 ? rows - r7 : 7;
                if (n7 > 0) mg_prefetch(smem, x1 + (int64_t)(row0 + r7) * d, (int64_t)d * 4, n7, d / 32);   // LN2' reads them next
            }
            mg_gemm_nt<192, 4, 4, false>(smem, dfb + (int64_t)row0 * mlp + kg * (mlp >> 2), mlp, rows, wF);
            MB_GEMM_DONE(1);
            mg_epi_f32_wide(reinterpret_cast<const float*>(smem), dh2 + (int64_t)kg * M * d + (int64_t)row0 * d + 192 * ng, d, rows, tid);
            MB_SYNC0(1);
            // ---------------- LN2' ----------------
            mg_ln_bwd(smem, d, dh2, M * d, x1, p.params + po + p.lo[LN2W], p.mean2 + (int64_t)l * M, p.rstd2 + (int64_t)l * M, p.dx, dxb2,
                      p.lnp2 + ((int64_t)l * npart + pblk) * 3 * d, row0, rows, member);
            MB_SYNC(wO, 64, 6, 2);
            // ---------------- OUT' + attention backward (member = head) ----------------
            if (member < p.H) {
                const int h = member;
                mg_prefetch(smem, qkv + (int64_t)row0 * 3 * d + 64 * h, (int64_t)d * 2, 3 * rows, 1);   // q | k | v head slices: rows of 128 B at stride d
                mg_gemm_nt<64, 4, 6, false>(smem, dxb2 + (int64_t)row0 * d, d, rows, wO);
                MB_GEMM_DONE(3);
                const float* cl = reinterpret_cast<const float*>(smem);          // [112][68] fp32: da of this head
                unsigned short* tiles = reinterpret_cast<unsigned short*>(smem + MG_ROWS * 68 * 4);   // 2 groups x {qt, kt, vt, dot_, dst, pt}
                const int grp = tid >> 8, gt = tid & 255;
                unsigned short* qt = tiles + grp * 6 * 64 * ATP;
                unsigned short *kt = qt + 64 * ATP, *vt = kt + 64 * ATP, *dot_ = vt + 64 * ATP, *dst = dot_ + 64 * ATP, *pt = dst + 64 * ATP;
                for (int im0 = 0; im0 < nimg; im0 += 2) {
                    const bool live = im0 + grp < nimg;
                    const int im = live ? im0 + grp : nimg - 1;
                    const unsigned short* base = qkv + (int64_t)(row0 + im * T) * 3 * d + 64 * h;
                    if (im0 > 0) __syncthreads();   // the tiles of the previous pass have been read
                    for (int i = gt; i < 64 * 8; i += 256) {
                        const int row = i >> 3, ch = i & 7;
                        const int rr = row < T ? row : T - 1;
                        *reinterpret_cast<au32x4_t*>(qt + row * ATP + ch * 8) = *reinterpret_cast<const au32x4_t*>(base + (int64_t)rr * 3 * d + ch * 8);
                        *reinterpret_cast<au32x4_t*>(kt + row * ATP + ch * 8) = *reinterpret_cast<const au32x4_t*>(base + d + (int64_t)rr * 3 * d + ch * 8);
                        *reinterpret_cast<au32x4_t*>(vt + row * ATP + ch * 8) = *reinterpret_cast<const au32x4_t*>(base + 2 * d + (int64_t)rr * 3 * d + ch * 8);
                        const float* c = cl + (im * T + rr) * 68 + ch * 8;
                        const float4 c0 = *reinterpret_cast<const float4*>(c), c1 = *reinterpret_cast<const float4*>(c + 4);
                        au32x4_t pk;
                        pk[0] = pack_bf16x2(c0.x, c0.y); pk[1] = pack_bf16x2(c0.z, c0.w);
                        pk[2] = pack_bf16x2(c1.x, c1.y); pk[3] = pack_bf16x2(c1.z, c1.w);
                        *reinterpret_cast<au32x4_t*>(dot_ + row * ATP + ch * 8) = pk;
                    }
                    __syncthreads();
                    mg_attn_bwd(3 * (int64_t)d, d, T, qt, kt, vt, dot_, dst, pt, dqkv + (int64_t)(row0 + im * T) * 3 * d + 64 * h, wave & 3, lane,
                                scale, live);
                }
            }
            MB_SYNC(wI, 192, 4, 3);
            // ---------------- IN': dh1[kg] = dqkv[:, quarter kg] W_in[quarter kg, 192 ng ..] ----------------
            {
                const int r7 = 7 * member, n7 = rows - r7 < 7 ? rows - r7 : 7;
                if (n7 > 0) mg_prefetch(smem, xin + (int64_t)(row0 + r7) * d, (int64_t)d * 4, n7, d / 32);   // LN1' reads them next
            }
            mg_gemm_nt<192, 4, 4, false>(smem, dqkv + (int64_t)row0 * 3 * d + kg * ((3 * d) >> 2), 3 * d, rows, wI);
            MB_GEMM_DONE(4);
            mg_epi_f32_wide(reinterpret_cast<const float*>(smem), dh1 + (int64_t)kg * M * d + (int64_t)row0 * d + 192 * ng, d, rows, tid);
            MB_SYNC0(4);
            // ---------------- LN1' ----------------
            mg_ln_bwd(smem, d, dh1, M * d, xin, p.params + po + p.lo[LN1W], p.mean1 + (int64_t)l * M, p.rstd1 + (int64_t)l * M, p.dx, dxb_out,
                      p.lnp1 + ((int64_t)l * npart + pblk) * 3 * d, row0, rows, member);
            {
                // the next product of this workgroup: block l - 1's PRJ', or the next panel's first one
                const bool more = l > 0 || panel + p.nteams < p.npanels;
                MegaW wN = w_prj(l > 0 ? l - 1 : p.L - 1);
                if (!more) wN.W = nullptr;
                MB_SYNC(wN, 192, 4, 5);
            }
        }
    }
#undef MB_SYNC
#undef MB_SYNC0
#undef MB_GEMM_DONE
#undef MB_TR
    MG_STAMP(stamp++);
}

// =====================================================================================================================
// Round 6: the same two launches with the waves' ROLES fixed for the whole launch (vit_fwd_mega2_kernel, vit_bwd_mega2_kernel).
//
// What bounds the kernels above (profiles/r05_mega_*_trace.txt): a product's k-loop is fed from L2 / the
// Infinity Cache at what ~76 KB in flight per CU buy — the member's weight slice (0.9 MB per block and CU, first touch per XCD from
// the Infinity Cache at ~33 GB/s per CU) costs more than the panel it multiplies (1.2 MB, L2 hits at ~68 GB/s) — while between the
// products (epilogue, hand-off, LayerNorm, attention: 45-55 % of a block) the memory pipes idle.  The weights depend on NO hand-off,
// and the four loader waves own 4 x 64 x 256 registers they never use.  So here the loader waves do nothing but move operand bytes:
// behind a product's last stage they load the NEXT product's weight slice — up to 48 x 16 B per lane, 192 KB per CU — into their
// REGISTERS while the consumer waves run the epilogue, the hand-off and the LayerNorm in between, and inside the next product's
// k-loop they put those stages into the ring with ds_write_b128 instead of LDS-DMA; the ring's LDS-DMA then carries the panel
// (and the few weight stages that did not fit).  vmcnt is in order, so a wave with long-lived loads in flight must never wait for a
// younger one: the loader waves take no part in epilogues, LayerNorm or attention (the four consumer waves do all of it, 256
// threads), never drain, and pass the workgroup's barriers as raw s_barriers.  Arithmetic, rounding points and summation orders
// are those of the kernels above: outputs are bit-identical (tests/test_gpu_vit.py).
// the weight stages [S0, S1) of the product that follows -> registers (plain loads: they stay in flight across the hand-offs)
template <int BN, int NS, int NK, bool LO, bool WKC>
__device__ __forceinline__ void mg_w_prefetch(const MegaW& w, int lw, int lane, u32x4_t (&wreg)[MG_WREG]) {
    using RS = MgRegStages<BN, NS, NK>;
    constexpr int NW = MgLoad<BN, LO>::NW;
    lane = mg_opaque(lane);   // (w.W is not null: the caller's product exists)
    unsigned voff[NW];
    mg_w_offsets2<BN, LO, WKC>(w, lw, lane, voff);
    const char* base = reinterpret_cast<const char*>(w.W);
    const int64_t step = mg_w_step<WKC>(w);
    // (inline asm: a load the compiler counts makes it put s_waitcnt vmcnt(0) in front of the first ds_write of the next k-loop — it
    // cannot follow the count across the hand-offs — which drains the ring's stream once per product.  The k-loop's own counted wait
    // in front of its barrier 0 covers these loads: they are the oldest the wave has in flight.)
    mg_static_for<(RS::S1 - RS::S0) * NW>([&](auto Q) {
        constexpr int s = decltype(Q)::value / NW, i = decltype(Q)::value % NW;
        u32x4_t& r = wreg[s * RS::MAXW + i];
        const unsigned vo = voff[i];
        const char* bs = base + step * (RS::S0 + s);
        asm volatile("global_load_dwordx4 %0, %1, %2 ; mg-stage" : "=&v"(r) : "v"(vo), "s"(bs) : "memory");
    });
}

// ---- the loader waves' half of a GEMM phase (NK k-steps).  VMEM issue order of a wave, oldest first: the register stages
// (mg_w_prefetch, behind the previous product), W of stages 0 .. NS-2 by LDS-DMA (inside the hand-off), then here A of stages
// 0 .. NS-2 and, behind barrier t, stage t + NS - 1: its A pieces by LDS-DMA and its W pieces either from registers (ds_write) or by
// LDS-DMA.  Barrier t needs stages <= t + 1 landed: the wait in front of it allows everything younger to fly (an immediate: the
// piece counts of the wave's class are compile-time).
template <int BN, int NS, int NK, bool LO, bool WKC>
__device__ __forceinline__ void mg_gemm_load(unsigned char* smem, const unsigned short* A, int lda, int rows, const MegaW& w, int lw, int lane,
                                             u32x4_t (&wreg)[MG_WREG]) {
    using PC = MgPieces<BN>;
    using RS = MgRegStages<BN, NS, NK>;
    using ML = MgLoad<BN, LO>;
    constexpr int NL = MG_NL, STAGE = PC::STAGE, A_BYTES = PC::A_BYTES, NA = ML::NA, NW = ML::NW;
    static_assert(NK >= NS && NS >= 4, "ring");
    lane = mg_opaque(lane);
    unsigned voffA[NA], voffW[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int byte = (lw + NL * i) * 1024 + lane * 16;
        const int row = byte >> 7, t = ((byte >> 4) & 7) ^ ((row >> 1) & 7);
        const int gr = row < rows ? row : rows - 1;
        voffA[i] = (unsigned)(gr * lda + 8 * t) * 2u;
    }
    mg_w_offsets2<BN, LO, WKC>(w, lw, lane, voffW);
    const char* curA = reinterpret_cast<const char*>(A);
    const char* curW = reinterpret_cast<const char*>(w.W);
    const int64_t stepW = mg_w_step<WKC>(w);
    // ds_write_b128 takes a 16-bit immediate offset: one base register per 64 KB window of the ring
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem) + lw * 1024 + lane * 16;
    const unsigned ldsw[3] = {lds0, lds0 + 65536u, lds0 + 131072u};
    auto issue_a = [&](int buf, int st) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(curA + 128 * st + voffA[i]),
                                             (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (lw + NL * i) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue_a(st, st);
    mg_static_for<NK>([&](auto T) {
        constexpr int t = decltype(T)::value;
        // items younger than the last one of stage t + 1: `ca` whole A sets and `cw` LDS-DMA W sets
        constexpr int lo = (t + 1 <= NS - 2) ? NS - 1 : t + 2;
        constexpr int hi0 = t + NS - 2, hi = hi0 < NK - 1 ? hi0 : NK - 1;   // stages issued so far: up to t + NS - 2
        constexpr int pro = (t + 1 <= NS - 2) ? NS - 3 - t : 0;               // A sets of the prologue behind A(t + 1)
        constexpr int ca = pro + (hi >= lo ? hi - lo + 1 : 0);
        constexpr int nreg = (hi < RS::S1 - 1 ? hi : RS::S1 - 1) - (lo > RS::S0 ? lo : RS::S0) + 1;   // register stages among lo .. hi
        constexpr int cw = (hi >= lo ? hi - lo + 1 : 0) - (nreg > 0 ? nreg : 0);
        constexpr int fly = ca * NA + cw * NW;
        mg_wait_vmcnt<(fly < 63 ? fly : 63)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the register stages written behind the previous barrier
        mg_barrier();
        constexpr int s = t + NS - 1;
        if constexpr (s < NK) {
            constexpr int buf = s % NS;
            issue_a(buf, s);
            if constexpr (RS::reg(s)) {
                mg_static_for<NW>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    constexpr int off = buf * STAGE + A_BYTES + NL * i * 1024;
                    mg_lds_store16<(off & 0xffff)>(ldsw[off >> 16], wreg[(s - RS::S0) * RS::MAXW + i]);
                });
            } else {
                mg_issue_w2<BN, LO>(smem, buf, curW + stepW * s, voffW, lw);
            }
        }
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mg_barrier();   // the consumers' "ring is free" barrier
    mg_barrier();   // the fp32 tile is complete (the consumers' __syncthreads)
}

// ---- "the epilogue's stores are issued": a CU has ONE in-order vector-memory pipe, and whatever a consumer wave loads or stores
// behind 192 KB of weight loads waits until most of them have landed (first build of this kernel: every epilogue 3 us longer, the
// whole gain of the shorter k-loops gone).  So the loader waves fire the register stages only when each consumer wave has issued the
// last store of its epilogue — the burst then queues BEHIND them and runs under the drain, the hand-off and (attention / LayerNorm)
// whatever the consumers do without the memory pipe.  One LDS word counts the signals of the launch; it only grows.
__device__ __forceinline__ void mg_signal(unsigned* flag, int lane) {
    if (MG_TRIGGER && lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void mg_wait_flag(unsigned* flag, unsigned want) {
    if (!MG_TRIGGER) return;
    for (int i = 0; i < 8192; ++i) {   // (bounded: a missed signal costs the burst's placement, never the launch)
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((int)(__builtin_amdgcn_readfirstlane(v) - want) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

// consumer waves: hand-off.  The loader waves pass the same two barriers in team_sync_l
template <class Params>
__device__ __forceinline__ void team_sync_c(const Params& p, int team, int member, unsigned target, unsigned long long* tr) {
    if (tr && threadIdx.x == 0) tr[0] = mg_clock();
    mg_drain();
    __syncthreads();
    if (threadIdx.x == 0) team_arrive(p, team, member, target, tr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mg_barrier();
}
template <class Early>
__device__ __forceinline__ void team_sync_l(Early&& early) {
    mg_barrier();
    early();
    mg_barrier();
}

// LayerNorm of the member's seven rows on FOUR waves: wave w takes rows w and w + 4, every load of both in flight before the first
// reduction; per row the arithmetic of mg_layernorm
__device__ __forceinline__ void mg_layernorm4(const MegaFwd& p, const float* x, const float* gamma, const float* beta, unsigned short* y,
                                              float* mean, float* rstd, int row0, int rows_valid, int member, int wave, int lane) {
    const int d = p.d;   // 768: 3 float4 per lane
    float4 v[2][3], gm[3], bt[3];
    bool ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 7 * member + wave + 4 * j;
        ok[j] = wave + 4 * j < 7 && r < rows_valid;
        if (ok[j]) {
            const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)(row0 + r) * d);
#pragma unroll
            for (int i = 0; i < 3; ++i) v[j][i] = xr[lane + 64 * i];
        }
    }
    if (!ok[0]) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        gm[i] = reinterpret_cast<const float4*>(gamma)[lane + 64 * i];
        bt[i] = reinterpret_cast<const float4*>(beta)[lane + 64 * i];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!ok[j]) continue;
        const int r = 7 * member + wave + 4 * j;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) s += v[j][i].x + v[j][i].y + v[j][i].z + v[j][i].w;
        const float mu = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float a = v[j][i].x - mu, b = v[j][i].y - mu, cc = v[j][i].z - mu, dd = v[j][i].w - mu;
            q += a * a + b * b + cc * cc + dd * dd;
        }
        const float var = wave_sum(q) / (float)d;
        const float rs = rsqrtf(var + p.eps);
        if (lane == 0) {
            mean[row0 + r] = mu;
            rstd[row0 + r] = rs;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = lane + 64 * i;
            const float o0 = (v[j][i].x - mu) * rs * gm[i].x + bt[i].x, o1 = (v[j][i].y - mu) * rs * gm[i].y + bt[i].y;
            const float o2 = (v[j][i].z - mu) * rs * gm[i].z + bt[i].z, o3 = (v[j][i].w - mu) * rs * gm[i].w + bt[i].w;
            st_sc1_8(y + (int64_t)(row0 + r) * d + 4 * c, pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
        }
    }
}

// the epilogue passes of the kernels above on 256 threads: BN = 48: 12 chunks of 4 columns x 21 rows per pass, 6 passes
__device__ __forceinline__ void mg_epi_residual4(const float* cl, const float* bias, const float* res, float* out, int64_t ld, int rows, int tid) {
    const int cc = tid % 12, r0 = tid / 12;
    if (r0 >= 21) return;
    const float4 b0 = *reinterpret_cast<const float4*>(bias + 4 * cc);
    float4 rr[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int row = r0 + 21 * it;
        if (row < rows) rr[it] = *reinterpret_cast<const float4*>(res + (int64_t)row * ld + 4 * cc);
    }
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int row = r0 + 21 * it;
        if (row >= rows) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(cl + row * 52 + 4 * cc);
        u32x4_t v;
        v[0] = __float_as_uint(c0.x + b0.x + rr[it].x); v[1] = __float_as_uint(c0.y + b0.y + rr[it].y);
        v[2] = __float_as_uint(c0.z + b0.z + rr[it].z); v[3] = __float_as_uint(c0.w + b0.w + rr[it].w);
        st_sc1_16(out + (int64_t)row * ld + 4 * cc, v);
    }
}
// BN = 192: 24 chunks of 8 columns x 10 rows per pass (240 threads), 12 passes
template <class F>
__device__ __forceinline__ void mg_epi_bf16_4(const float* cl, const float* b0p, const float* b1p, int tid, F&& emit) {
    const int cc = tid % 24, r0 = tid / 24;
    if (r0 >= 10) return;
    const float4 b0 = *reinterpret_cast<const float4*>(b0p), b1 = *reinterpret_cast<const float4*>(b1p);
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int row = r0 + 10 * it;
        if (row >= MG_ROWS) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(cl + row * 196 + 8 * cc);
        const float4 c1 = *reinterpret_cast<const float4*>(cl + row * 196 + 8 * cc + 4);
        u32x4_t pk;
        pk[0] = pack_bf16x2(c0.x + b0.x, c0.y + b0.y); pk[1] = pack_bf16x2(c0.z + b0.z, c0.w + b0.w);
        pk[2] = pack_bf16x2(c1.x + b1.x, c1.y + b1.y); pk[3] = pack_bf16x2(c1.z + b1.z, c1.w + b1.w);
        emit(row, pk);
    }
}

// ---- the loader waves' program of the forward: the walk of vit_fwd_mega2_kernel's consumer program, barrier for barrier ---------
template <bool LO>
__device__ __forceinline__ void mg_fwd_loader(const MegaFwd& p, unsigned char* smem, unsigned* flag, int team, int member, int lw, int lane) {
    const int d = p.d, T = p.T, mlp = p.mlp;
    const int64_t M = p.M, Mp = p.Mp;
    (void)M;
    u32x4_t wreg[MG_WREG];
    auto slices = [&](int l, MegaW& wQ, MegaW& wO, MegaW& wF, MegaW& wP) {
        const int64_t po = (int64_t)l * p.layer_stride;
        wQ = MegaW{p.shadow + po + p.lo[QKVW], d, d >> 6, 0, member < p.H ? member : member - p.H, d};
        wO = MegaW{p.shadow + po + p.lo[OUTW], d, d >> 6, 48 * member, -1, d};
        wF = MegaW{p.shadow + po + p.lo[FCW], d, d >> 6, 192 * member, -1, d};
        wP = MegaW{p.shadow + po + p.lo[PRW], mlp, mlp >> 6, 48 * member, -1, d};
    };
    // ONE loop over (panel, block) with the QKV stages loaded at its top: every register stage has a single defining point in front
    // of its single use (a value that crossed the loop's back edge could be copied by the compiler while it is still in flight)
    bool first = true;
    unsigned want = 0;   // signals (mg_signal, one per consumer wave) the waits so far have covered
    for (int panel = team; panel < p.npanels; panel += p.nteams) {
        const int img0 = panel * p.P;
        const int nimg = (p.B - img0) < p.P ? (p.B - img0) : p.P;
        const int row0 = img0 * T, rows = nimg * T;
        for (int l = 0; l < p.L; ++l) {
            const int64_t li = p.save ? l : 0;
            const unsigned short* h1 = p.h1 + li * Mp * d;
            const unsigned short* h2 = p.h2 + li * Mp * d;
            const unsigned short* at = p.a + li * Mp * d;
            const unsigned short* ga = p.gact + li * Mp * mlp;
            MegaW wQ, wO, wF, wP;
            slices(l, wQ, wO, wF, wP);
            // QKV: members H .. 15 have no head of their own.  They run the product of head member - H all the same (operands that another
            // member of the team pulls through the same L2 at the same time; the consumer waves drop the result): the loader program then
            // has ONE path for every member — no arm on which a register stage would stay in flight past its product, for the compiler
            // or for tools/micro/check_inflight_regs.py
            if (!first) mg_wait_flag(flag, want += MG_NC);             // the previous block's c_proj: its epilogue's stores are issued
            mg_w_prefetch<192, 4, 12, LO, true>(wQ, lw, lane, wreg);   // (or the kernel's start)
            if (!first) team_sync_l([] {});                             // ... that c_proj's hand-off
            first = false;
            team_sync_l([&] { mg_w_prologue2<192, 4, LO, true>(smem, wQ, lw, lane); });   // LN1 | hand-off
            mg_gemm_load<192, 4, 12, LO, true>(smem, h1 + (int64_t)row0 * d, d, rows, wQ, lw, lane, wreg);
            mg_barrier();   // tiles complete (the consumers' __syncthreads in front of the attention)
            mg_wait_flag(flag, want += MG_NC);
            mg_w_prefetch<48, 7, 12, LO, true>(wO, lw, lane, wreg);
            team_sync_l([&] { mg_w_prologue2<48, 7, LO, true>(smem, wO, lw, lane); });
            // out_proj
            mg_gemm_load<48, 7, 12, LO, true>(smem, at + (int64_t)row0 * d, d, rows, wO, lw, lane, wreg);
            mg_wait_flag(flag, want += MG_NC);
            mg_w_prefetch<192, 4, 12, LO, true>(wF, lw, lane, wreg);
            team_sync_l([] {});
            // LN2 | hand-off
            team_sync_l([&] { mg_w_prologue2<192, 4, LO, true>(smem, wF, lw, lane); });
            // c_fc
            mg_gemm_load<192, 4, 12, LO, true>(smem, h2 + (int64_t)row0 * d, d, rows, wF, lw, lane, wreg);
            mg_wait_flag(flag, want += MG_NC);
            mg_w_prefetch<48, 7, 48, LO, true>(wP, lw, lane, wreg);
            team_sync_l([&] { mg_w_prologue2<48, 7, LO, true>(smem, wP, lw, lane); });
            // c_proj
            mg_gemm_load<48, 7, 48, LO, true>(smem, ga + (int64_t)row0 * mlp, mlp, rows, wP, lw, lane, wreg);
        }
    }
    if (!first) team_sync_l([] {});   // the last c_proj's hand-off
}

__global__ __launch_bounds__(MG_THREADS) void vit_fwd_mega2_kernel(const MegaFwd p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[MG_LDS];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    int team, member;
    mg_place(p.nteams, p.xsplit, b, team, member);
    __shared__ unsigned flag;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (tid == 0) flag = 0;
    if (wave >= MG_NC) {
        mg_barrier();
        if (wave < MG_NC + 2) mg_fwd_loader<true>(p, smem, &flag, team, member, wave - MG_NC, lane);
        else mg_fwd_loader<false>(p, smem, &flag, team, member, wave - MG_NC, lane);
        return;
    }
    __syncthreads();
    // ================================ consumer waves: everything but operand movement ================================
    const int d = p.d, T = p.T, mlp = p.mlp;
    const int64_t M = p.M, Mp = p.Mp;
    const float scale = 1.0f / sqrtf(64.0f);
    unsigned seq = 0;
    int stamp = 0;
    MG_STAMP(stamp++);
#define MG2_SYNC(k_)                                                                                                  \
    team_sync_c(p, team, member, ++seq * MG_TEAM,                                                                      \
                (p.trace && l == 0 && panel == team) ? p.trace + (size_t)blockIdx.x * 64 + 8 + 4 * (k_) : nullptr);   \
    if (l == 0) MG_STAMP(stamp++)
#define MG2_GEMM_DONE(k_)                                                                                            \
    if (p.trace && l == 0 && panel == team && tid == 0) p.trace[(size_t)blockIdx.x * 64 + 32 + (k_)] = mg_clock()
    for (int panel = team; panel < p.npanels; panel += p.nteams) {
        const int img0 = panel * p.P;
        const int nimg = (p.B - img0) < p.P ? (p.B - img0) : p.P;
        const int row0 = img0 * T, rows = nimg * T;
        for (int l = 0; l < p.L; ++l) {
            const int64_t po = (int64_t)l * p.layer_stride;
            const int64_t li = p.save ? l : 0;
            const float* xin = p.xs + (p.save ? (int64_t)l : (int64_t)(l & 1)) * M * d;
            float* xout = p.xs + (p.save ? (int64_t)(l + 1) : (int64_t)((l + 1) & 1)) * M * d;
            float* x1 = p.x1 + li * M * d;
            unsigned short* h1 = p.h1 + li * Mp * d;
            unsigned short* h2 = p.h2 + li * Mp * d;
            unsigned short* at = p.a + li * Mp * d;
            unsigned short* qkv = p.qkv + li * M * 3 * d;
            unsigned short* f = p.f + li * M * mlp;
            unsigned short* ga = p.gact + li * Mp * mlp;
            // ---------------- LN1 ----------------
            mg_layernorm4(p, xin, p.params + po + p.lo[LN1W], p.params + po + p.lo[LN1B], h1, p.mean1 + li * M, p.rstd1 + li * M, row0, rows,
                          member, wave, lane);
            MG2_SYNC(0);
            // ---------------- QKV + attention (member = head) ----------------
            mg_gemm_consume<192, 4, 4, true>(smem, d >> 6, wave, lane);   // (members H .. 15: see mg_fwd_loader)
            __syncthreads();
            MG2_GEMM_DONE(0);
            if (member < p.H) {
                const int h = member;
                const float* cl = reinterpret_cast<const float*>(smem);
                unsigned short* tiles = reinterpret_cast<unsigned short*>(smem + MG_ROWS * 196 * 4);  // q | k | v : [128][ATP] each
                {
                    const int cc = tid % 24, part = cc >> 3, c64 = (cc & 7) * 8;
                    const float* bp = p.params + po + p.lo[QKVB] + part * d + 64 * h + c64;
                    unsigned short* gq = qkv + (int64_t)row0 * 3 * d + part * d + 64 * h + c64;
                    unsigned short* tq = tiles + part * 128 * ATP + c64;
                    mg_epi_bf16_4(cl, bp, bp + 4, tid, [&](int row, u32x4_t pk) {
                        *reinterpret_cast<u32x4_t*>(tq + row * ATP) = pk;
                        if (row < rows) *reinterpret_cast<u32x4_t*>(gq + (int64_t)row * 3 * d) = pk;   // backward re-reads it (a later launch)
                    });
                }
                // v rows 112..127 are read (times P = 0) by the last image's key fragments: they must be finite
                for (int c = tid; c < 16 * ATP / 8; c += MG_CT)
                    *reinterpret_cast<u32x4_t*>(tiles + 2 * 128 * ATP + MG_ROWS * ATP + c * 8) = (u32x4_t){0u, 0u, 0u, 0u};
                __syncthreads();
                mg_signal(&flag, lane);   // qkv is stored: the out_proj stages may load under the attention
                for (int im = 0; im < nimg; ++im)
                    mg_attention(tiles, tiles + 128 * ATP, tiles + 2 * 128 * ATP, im * T, T, wave, lane, scale,
                                 at + (int64_t)(row0 + im * T) * d + 64 * h, d);
            } else {
                __syncthreads();
                mg_signal(&flag, lane);
            }
            MG2_SYNC(1);
            // ---------------- out_proj + bias + residual -> x1 ----------------
            {
                const int n0 = 48 * member;
                mg_gemm_consume<48, 3, 7, true>(smem, d >> 6, wave, lane);
                __syncthreads();
                MG2_GEMM_DONE(1);
                mg_epi_residual4(reinterpret_cast<const float*>(smem), p.params + po + p.lo[OUTB] + n0, xin + (int64_t)row0 * d + n0,
                                 x1 + (int64_t)row0 * d + n0, d, rows, tid);
                mg_signal(&flag, lane);
            }
            MG2_SYNC(2);
            // ---------------- LN2 ----------------
            mg_layernorm4(p, x1, p.params + po + p.lo[LN2W], p.params + po + p.lo[LN2B], h2, p.mean2 + li * M, p.rstd2 + li * M, row0, rows,
                          member, wave, lane);
            MG2_SYNC(3);
            // ---------------- c_fc + bias, activation -> f, g ----------------
            {
                const int n0 = 192 * member;
                mg_gemm_consume<192, 4, 4, true>(smem, d >> 6, wave, lane);
                __syncthreads();
                MG2_GEMM_DONE(2);
                const int cc = tid % 24;
                const float* bp = p.params + po + p.lo[FCB] + n0 + 8 * cc;
                unsigned short* fo = f + (int64_t)row0 * mlp + n0 + 8 * cc;
                unsigned short* go = ga + (int64_t)row0 * mlp + n0 + 8 * cc;
                auto epi = [&](auto ACT) {
                    constexpr int act = decltype(ACT)::value;
                    mg_epi_bf16_4(reinterpret_cast<const float*>(smem), bp, bp + 4, tid, [&](int row, u32x4_t pk) {
                        if (row >= rows) return;
                        *reinterpret_cast<u32x4_t*>(fo + (int64_t)row * mlp) = pk;   // pre-activation: only the backward reads it
                        u32x4_t pk2;
#pragma unroll
                        for (int hh = 0; hh < 4; ++hh)
                            pk2[hh] = pack_bf16x2(act_fwd(__uint_as_float(pk[hh] << 16), act), act_fwd(__uint_as_float(pk[hh] & 0xffff0000u), act));
                        st_sc1_16(go + (int64_t)row * mlp, pk2);
                    });
                };
                if (p.act == 1) epi(std::integral_constant<int, 1>{});
                else if (p.act == 2) epi(std::integral_constant<int, 2>{});
                else epi(std::integral_constant<int, 0>{});
                mg_signal(&flag, lane);
            }
            MG2_SYNC(4);
            // ---------------- c_proj + bias + residual -> next block's input ----------------
            {
                const int n0 = 48 * member;
                mg_gemm_consume<48, 3, 7, true>(smem, mlp >> 6, wave, lane);
                __syncthreads();
                MG2_GEMM_DONE(3);
                mg_epi_residual4(reinterpret_cast<const float*>(smem), p.params + po + p.lo[PRB] + n0, x1 + (int64_t)row0 * d + n0,
                                 xout + (int64_t)row0 * d + n0, d, rows, tid);
                mg_signal(&flag, lane);
            }
            MG2_SYNC(5);
        }
    }
#undef MG2_SYNC
#undef MG2_GEMM_DONE
    MG_STAMP(stamp++);
}

// ---- host side --------------------------------------------------------------------------------------------------------
// Per-device state of the persistent forward.  The arrival counters are one block per device: persistent launches of a device
// are SERIALISED (a launch on another stream first waits for the previous launch's completion event) — two of them in flight at
// once could each hold a part of the chip's CUs and wait for members of their own that the other's workgroups keep from
// becoming resident (both would give up after the spin bound).  Kernels that do not spin (everything else in the library) may
// run beside a persistent launch: they always finish and hand their CUs back.
struct MegaDev {
    unsigned* state = nullptr;      // [16 teams x 32 words: zeroed before every launch | status 32 words: sticky] = 2176 B
    unsigned* host_flag = nullptr;  // pinned, mapped: launch number of the first launch that gave up (0 = none)
    hipEvent_t done = nullptr;      // completion of the last persistent launch
    hipStream_t last = nullptr;
    bool any = false;
    int cus = 0, resident = -1;     // CUs of the device; workgroups of the kernel that fit one CU (occupancy query)
    int resident_bwd = -1;          // the same for the persistent backward
    unsigned launches = 0, launches_bwd = 0;
    unsigned ctr = 0;               // value of the team counters after the launches queued so far
    bool ctr_dirty = false;         // the counters will not be uniform after the launches queued so far (injected failure, ragged rounds)
    int ctr_teams = 0;              // teams of the last launch
};
static MegaDev g_mega_dev[16];
static int g_mega_mode = -1;               // swat_vit_set_mega / SWAT_VIT_MEGA: 0 = op-by-op forward, 1 = persistent forward when eligible
static unsigned long long* g_mega_trace = nullptr;
static int g_fail_launches = 0, g_fail_sync = 0;   // swat_vit_mega_inject_failure
static int g_mega_roles = -1;              // swat_vit_set_mega_roles / SWAT_VIT_MEGA_ROLES: 1 = every wave takes part in every phase
                                           // (rounds 3-5), 2 = loader waves move operand bytes only and carry the next product's
                                           // weight stages in registers (round 6; default)
static int g_mega_bwd = -1;                // swat_vit_set_mega_backward / SWAT_VIT_MEGA_BWD
static unsigned long long* g_mega_trace_bwd = nullptr;
static int g_fail_bwd_launches = 0, g_fail_bwd_sync = 0;

// The team counters only grow: a launch starts at md->ctr (what they hold when every earlier launch has completed all its
// hand-offs) and leaves them `handoffs` x 16 higher — no memset on the stream in front of each launch (6-8 us on the critical path,
// twice per step).  After a give-up (the pinned host word says so) or near the 32-bit wrap they are zeroed, stream-ordered.
// Every team must leave a launch with the same count: a launch on fewer teams than the last one, or with panels that do not fill
// whole rounds of teams (tests), is followed by a reset too.
static int mega_counter_base(MegaDev* md, int npanels, int nteams, int L, hipStream_t s, unsigned& base) {
    const unsigned handoffs = (unsigned)((npanels + nteams - 1) / nteams) * 6u * (unsigned)L;
    if (md->ctr > 0xE0000000u || (md->host_flag && __atomic_load_n(md->host_flag, __ATOMIC_RELAXED) != 0) || md->ctr_dirty ||
        nteams != md->ctr_teams) {
        if (hipMemsetAsync(md->state, 0, 16 * 128, s) != hipSuccess) return -1;
        md->ctr = 0;
        md->ctr_dirty = false;
    }
    base = md->ctr;
    md->ctr += handoffs * MG_TEAM;
    md->ctr_teams = nteams;
    if (npanels % nteams) md->ctr_dirty = true;   // teams end this launch at different counts
    return 0;
}

// hand-off counters + sticky status word (device), the pinned host words, the completion event: once per device
static int mega_state_ensure(MegaDev* md) {
    if (md->state) return 0;
    if (hipMalloc((void**)&md->state, 17 * 128) != hipSuccess || hipMemset(md->state, 0, 17 * 128) != hipSuccess ||
        hipHostMalloc((void**)&md->host_flag, 128, hipHostMallocMapped) != hipSuccess ||
        hipEventCreateWithFlags(&md->done, hipEventDisableTiming) != hipSuccess) {
        set_error("vit_mega: allocation of the hand-off state failed");
        md->state = nullptr;
        return -1;
    }
    md->host_flag[0] = 0;   // launch number of the first persistent launch of this device that gave up
    md->host_flag[1] = 0;   // data-parallel runs: number of the first step some rank voided (swat_vit_mega_void_absorb)
    return 0;
}

static int g_mega_xsplit = -1;             // swat_vit_set_mega_xsplit / SWAT_VIT_MEGA_XSPLIT: XCDs a team's 16 members are spread over (1, 2, 4)
static int mega_xsplit() {
    if (g_mega_xsplit < 0) {
        const char* e = getenv("SWAT_VIT_MEGA_XSPLIT");
        const int v = e ? atoi(e) : 4;   // round 6, alternating in one process: forward 0.857 / 0.840 / 0.835 / 0.89 ms and persistent backward
        g_mega_xsplit = (v == 1 || v == 2 || v == 8) ? v : 4;   // -0 / -38 / -46 / -10 us at 1 / 2 / 4 / 8 XCDs per team (tools/micro/xsplit_ab.py)
    }
    return g_mega_xsplit;
}

static int mega_roles() {
    if (g_mega_roles < 0) {
        const char* e = getenv("SWAT_VIT_MEGA_ROLES");
        g_mega_roles = e && atoi(e) == 2 ? 2 : 1;
    }
    return g_mega_roles;
}

static MegaDev* mega_dev() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    return &g_mega_dev[dev];
}

// device word the optimizer / loss-accumulation kernels test before they write (null: no persistent forward has run on this device)
const unsigned* mega_guard_word() {
    MegaDev* md = mega_dev();
    return md && md->state ? md->state + 16 * 32 : nullptr;
}

// -> 0: launched; 1: not eligible (the caller runs the blocks op by op); < 0: error
int vit_mega_forward(int B, int T, int d, int H, int mlp, int L, int act, int save, int M, int Mp, const float* params,
                     const unsigned short* shadow, const int64_t* layer_offs, float* xs, float* x1, unsigned short* h1,
                     unsigned short* h2, unsigned short* a, unsigned short* qkv, unsigned short* f, unsigned short* gact, float* mean1,
                     float* rstd1, float* mean2, float* rstd2, hipStream_t s) {
    // one panel = 2 images (attention reads 64 key rows from the second image's first row: 2 T + 14 <= 128 tile rows)
    if (g_mega_mode < 0) {
        const char* e = getenv("SWAT_VIT_MEGA");
        g_mega_mode = e ? (atoi(e) == 0 ? 0 : atoi(e) == 2 ? 2 : 1) : 1;
    }
    if (g_mega_mode == 0 || d != 768 || mlp != 3072 || H != 12 || d / H != 64 || T < 38 || T > 56 || B < 1 || L < 1) return 1;
    // 16 teams walk the panels in rounds of 16: a launch pays whole rounds (~0.8 ms each for 12 blocks), so batches that leave
    // the last round mostly empty — and big batches, where the op-by-op products fill the chip on their own (B = 256: M = 12 800
    // rows) — stay on the op-by-op path.  mode 2 (tests) takes every batch.
    {
        const int np = (B + 1) / 2, rounds = (np + 15) / 16;
        if (g_mega_mode != 2 && (rounds > 2 || np < rounds * 16 - 3)) return 1;
    }
    const int64_t stride = L > 1 ? layer_offs[12] - layer_offs[0] : 0;
    for (int l = 1; l < L; ++l)
        for (int k = 0; k < 12; ++k)
            if (layer_offs[12 * l + k] - layer_offs[k] != (int64_t)l * stride) return 1;  // blocks are not laid out at a constant stride
    for (int k = 0; k < 12; ++k)
        if (layer_offs[k] % 8) return 1;  // 16-B rows of the bf16 shadow
    if (stride % 8) return 1;
    MegaDev* md = mega_dev();
    if (!md) return 1;
    const int npanels = (B + 1) / 2, nteams = npanels < 16 ? npanels : 16;
    // every team member must be resident at once (the hand-offs wait for all of them): one workgroup per CU (156 KB of LDS each),
    // so the grid must not exceed the CUs this process sees (a CU-masked or partitioned device, a smaller part), and the
    // occupancy query must admit the workgroup at all.  Otherwise: op by op.
    const int roles = mega_roles();
    if (md->resident < 0) {
        int cus = 0, occ = 0, occ2 = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, (int)(md - g_mega_dev)) != hipSuccess) cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vit_fwd_mega_kernel, MG_THREADS, 0) != hipSuccess) occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ2, vit_fwd_mega2_kernel, MG_THREADS, 0) != hipSuccess) occ2 = 0;
        (void)hipGetLastError();
        md->cus = cus;
        md->resident = occ < occ2 ? occ : occ2;
    }
    if (md->resident < 1 || nteams * MG_TEAM > md->cus) return 1;   // (156 KB of LDS: one workgroup per CU whatever the query says)
    if (mega_state_ensure(md) != 0) return -1;
    MegaFwd m{};
    m.B = B; m.T = T; m.d = d; m.H = H; m.mlp = mlp; m.L = L; m.act = act; m.save = save; m.M = M; m.Mp = Mp;
    m.eps = 1e-5f;
    m.params = params; m.shadow = shadow;
    for (int k = 0; k < 12; ++k) m.lo[k] = layer_offs[k];
    m.layer_stride = stride;
    m.xs = xs; m.x1 = x1; m.h1 = h1; m.h2 = h2; m.a = a; m.qkv = qkv; m.f = f; m.gact = gact;
    m.mean1 = mean1; m.rstd1 = rstd1; m.mean2 = mean2; m.rstd2 = rstd2;
    m.P = 2;
    m.npanels = npanels;
    m.nteams = nteams;
    m.xsplit = mega_xsplit();
    m.ctr = md->state;
    m.status = md->state + 16 * 32;
    m.host_flag = md->host_flag;   // (mapped pinned memory: the host pointer is the device pointer under unified addressing)
    m.launch_no = ++md->launches;
    m.spin_ticks = 50000000ull;    // 0.5 s
    m.fail_sync = 0;
    if (g_fail_launches > 0) {
        --g_fail_launches;
        m.fail_sync = g_fail_sync;
        m.spin_ticks = 200000ull;  // 2 ms: the injected failure need not cost the test half a second
    }
    m.trace = g_mega_trace;
    if (md->any && md->last != s) SWAT_HIP(hipStreamWaitEvent(s, md->done, 0));   // persistent launches of a device never overlap
    // (the status word behind the counters is sticky: read-and-clear)
    SWAT_REQUIRE(mega_counter_base(md, npanels, nteams, L, s, m.ctr_base) == 0, "vit_mega: cannot reset the hand-off counters");
    if (m.fail_sync) md->ctr_dirty = true;
    // products + attention inside the launch (what bench.py's roofline divides by the launch's duration)
    const double flops = (double)L * B * T * (2.0 * d * 3 * d + 2.0 * d * d + 4.0 * d * mlp + 4.0 * T * d);
    hipEvent_t e0, e1;
    auto kern = roles == 2 ? vit_fwd_mega2_kernel : vit_fwd_mega_kernel;
    if (gemm_prof_take(9, flops, e0, e1)) hipExtLaunchKernelGGL(kern, dim3(m.nteams * MG_TEAM), dim3(MG_THREADS), 0, s, e0, e1, 0, m);
    else hipLaunchKernelGGL(kern, dim3(m.nteams * MG_TEAM), dim3(MG_THREADS), 0, s, m);
    const int rc = check_launch("vit_fwd_mega_kernel");
    if (rc == 0) {
        SWAT_HIP(hipEventRecord(md->done, s));
        md->last = s;
        md->any = true;
    }
    return rc;
}

// shapes the persistent backward can serve, whatever the modes say (csrc/vit.hip carves its per-block gradient buffers by this)
bool vit_mega_backward_fits(int B, int T, int d, int H, int mlp) {
    return d == 768 && mlp == 3072 && H == 12 && T >= 38 && T <= 56 && B >= 1 && (B + 1) / 2 <= 32;
}
// ... and does serve now: the forward's rule (whole rounds of 16 panels, at most two)
bool vit_mega_backward_shape(int B, int T, int d, int H, int mlp, int L) {
    if (g_mega_mode < 0) {
        const char* e = getenv("SWAT_VIT_MEGA");
        g_mega_mode = e ? (atoi(e) == 0 ? 0 : atoi(e) == 2 ? 2 : 1) : 1;
    }
    if (g_mega_bwd < 0) {
        const char* e = getenv("SWAT_VIT_MEGA_BWD");
        g_mega_bwd = e ? (atoi(e) != 0) : 1;
    }
    if (g_mega_mode == 0 || !g_mega_bwd || L < 1 || !vit_mega_backward_fits(B, T, d, H, mlp)) return false;
    const int np = (B + 1) / 2, rounds = (np + 15) / 16;
    if (g_mega_mode != 2 && np < rounds * 16 - 3) return false;
    return true;
}

// bytes per block of the LayerNorm partial sums of the persistent backward: [panels * 16][3][d] floats
int64_t vit_mega_backward_lnp_bytes(int B, int d) { return (int64_t)((B + 1) / 2) * MG_TEAM * 3 * d * 4; }

// -> 0: launched; 1: not eligible (the caller runs the blocks op by op); < 0: error
int vit_mega_backward(int B, int T, int d, int H, int mlp, int L, int act, int M, int Mp, const float* params, const unsigned short* shadow,
                      const int64_t* layer_offs, const float* xs, const float* x1, const unsigned short* qkv, const unsigned short* f,
                      const float* mean1, const float* rstd1, const float* mean2, const float* rstd2, float* dx, unsigned short* dxb,
                      unsigned short* dfb, unsigned short* dxb2, unsigned short* dqkv, float* dh2, float* dh1, float* lnp1, float* lnp2,
                      hipStream_t s) {
    if (!vit_mega_backward_shape(B, T, d, H, mlp, L)) return 1;
    const int64_t stride = L > 1 ? layer_offs[12] - layer_offs[0] : 0;
    for (int l = 1; l < L; ++l)
        for (int k = 0; k < 12; ++k)
            if (layer_offs[12 * l + k] - layer_offs[k] != (int64_t)l * stride) return 1;
    for (int k = 0; k < 12; ++k)
        if (layer_offs[k] % 8) return 1;
    if (stride % 8) return 1;
    MegaDev* md = mega_dev();
    if (!md || !md->state) return 1;   // (the state is created by the first persistent forward: a backward follows one)
    const int npanels = (B + 1) / 2, nteams = npanels < 16 ? npanels : 16;
    if (md->resident_bwd < 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vit_bwd_mega_kernel, MG_THREADS, 0) != hipSuccess) occ = 0;
        (void)hipGetLastError();
        md->resident_bwd = occ;
    }
    if (md->resident_bwd < 1 || md->cus < 1 || nteams * MG_TEAM > md->cus) return 1;
    MegaBwd m{};
    m.B = B; m.T = T; m.d = d; m.H = H; m.mlp = mlp; m.L = L; m.act = act; m.save = 1; m.M = M; m.Mp = Mp;
    m.P = 2; m.npanels = npanels; m.nteams = nteams;
    m.xsplit = mega_xsplit();
    m.params = params; m.shadow = shadow;
    for (int k = 0; k < 12; ++k) m.lo[k] = layer_offs[k];
    m.layer_stride = stride;
    m.xs = xs; m.x1 = x1; m.qkv = qkv; m.f = f; m.mean1 = mean1; m.rstd1 = rstd1; m.mean2 = mean2; m.rstd2 = rstd2;
    m.dx = dx; m.dxb = dxb; m.dfb = dfb; m.dxb2 = dxb2; m.dqkv = dqkv; m.dh2 = dh2; m.dh1 = dh1; m.lnp1 = lnp1; m.lnp2 = lnp2;
    m.ctr = md->state;
    m.status = md->state + 16 * 32;
    m.host_flag = md->host_flag;
    m.launch_no = md->launches;    // the forward it follows: a give-up voids that step (the host replays from there)
    ++md->launches_bwd;
    m.spin_ticks = 50000000ull;
    m.fail_sync = 0;
    if (g_fail_bwd_launches > 0) {
        --g_fail_bwd_launches;
        m.fail_sync = g_fail_bwd_sync;
        m.spin_ticks = 200000ull;
    }
    m.trace = g_mega_trace_bwd;
    if (md->any && md->last != s) SWAT_HIP(hipStreamWaitEvent(s, md->done, 0));
    SWAT_REQUIRE(mega_counter_base(md, npanels, nteams, L, s, m.ctr_base) == 0, "vit_mega: cannot reset the hand-off counters");
    if (m.fail_sync) md->ctr_dirty = true;
    // data-gradient products + attention backward inside the launch
    const double flops = (double)L * B * T * (2.0 * d * 3 * d + 2.0 * d * d + 4.0 * d * mlp + 10.0 * T * d);
    hipEvent_t e0, e1;
    if (gemm_prof_take(12, flops, e0, e1)) hipExtLaunchKernelGGL(vit_bwd_mega_kernel, dim3(m.nteams * MG_TEAM), dim3(MG_THREADS), 0, s, e0, e1, 0, m);
    else hipLaunchKernelGGL(vit_bwd_mega_kernel, dim3(m.nteams * MG_TEAM), dim3(MG_THREADS), 0, s, m);
    const int rc = check_launch("vit_bwd_mega_kernel");
    if (rc == 0) {
        SWAT_HIP(hipEventRecord(md->done, s));
        md->last = s;
        md->any = true;
    }
    return rc;
}

// ---- data-parallel runs: a give-up on ANY rank voids the step on EVERY rank ------------------------------------------------------
// The device guard (the sticky status word) keeps a rank's own optimizer launches from applying a void step — but its void gradients
// would still be summed into the other ranks' updates.  So every step carries one float through a collective of its own:
//   swat_vit_mega_void_flag   : dst[0] = 1 if this device's status word is set (a persistent launch gave up), else 0
//   (the caller all-reduces dst with MAX / SUM over the ranks, on the stream the gradient all-reduces run on, BEFORE the step's
//   first optimizer launch)
//   swat_vit_mega_void_absorb : src[0] != 0 -> raise the local guard as well (status word, if it is still clear) and publish
//   `step_seq` in the pinned host word swat_vit_mega_poll_void reads (first voided step wins).
// Every rank then skips the step's updates and the host replays from `step_seq` on, on every rank alike (swat_amd/step.py).
__global__ void mega_void_flag_kernel(const unsigned* status, float* dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) dst[0] = (status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1.0f : 0.0f;
}
__global__ void mega_void_absorb_kernel(const float* src, unsigned* status, unsigned* host_flag, unsigned step_seq) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || !(src[0] != 0.0f)) return;
    unsigned expect = 0;
    __hip_atomic_compare_exchange_strong(status, &expect, 0xC0000000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    expect = 0;
    __hip_atomic_compare_exchange_strong(host_flag + 1, &expect, step_seq, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace swat
SWAT_API int swat_vit_mega_void_flag(float* dst, swat_stream_t stream) {
    SWAT_REQUIRE(dst, "swat_vit_mega_void_flag: null pointer");
    swat::MegaDev* md = swat::mega_dev();
    hipLaunchKernelGGL(swat::mega_void_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, md && md->state ? md->state + 16 * 32 : nullptr, dst);
    return swat::check_launch("mega_void_flag_kernel");
}
SWAT_API int swat_vit_mega_void_absorb(const float* src, int step_seq, swat_stream_t stream) {
    SWAT_REQUIRE(src && step_seq > 0, "swat_vit_mega_void_absorb: null pointer or step number < 1");
    swat::MegaDev* md = swat::mega_dev();
    // (a rank whose own launches were never persistent still has to honour another rank's give-up: the guard exists from here on)
    SWAT_REQUIRE(md && swat::mega_state_ensure(md) == 0, "swat_vit_mega_void_absorb: no device state");
    hipLaunchKernelGGL(swat::mega_void_absorb_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, md->state + 16 * 32, md->host_flag,
                       (unsigned)step_seq);
    return swat::check_launch("mega_void_absorb_kernel");
}
// no synchronisation: 0, or the number (the caller's own count, swat_vit_mega_void_absorb) of the first step some rank voided
SWAT_API int swat_vit_mega_poll_void(void) {
    swat::MegaDev* md = swat::mega_dev();
    if (!md || !md->host_flag) return 0;
    return (int)__atomic_load_n(md->host_flag + 1, __ATOMIC_RELAXED);
}
namespace swat {

}  // namespace swat
using namespace swat;

// 0 = data gradients of the blocks op by op, 1 = as one persistent launch when eligible (default; SWAT_VIT_MEGA_BWD)
SWAT_API int swat_vit_set_mega_backward(int mode) {
    SWAT_REQUIRE(mode == 0 || mode == 1, "swat_vit_set_mega_backward: mode must be 0 or 1");
    g_mega_bwd = mode;
    return 0;
}

// tuning / tests: as swat_vit_mega_inject_failure, for the next persistent BACKWARD launches (6 hand-offs per block, top block first)
SWAT_API int swat_vit_mega_backward_inject_failure(int launches, int handoff) {
    SWAT_REQUIRE(launches >= 0 && handoff >= 0, "swat_vit_mega_backward_inject_failure: negative argument");
    g_fail_bwd_launches = launches;
    g_fail_bwd_sync = handoff;
    return 0;
}

SWAT_API int swat_vit_mega_backward_launches(void) {
    MegaDev* md = mega_dev();
    return md ? (int)md->launches_bwd : 0;
}

SWAT_API int swat_vit_mega_backward_trace(void* device_buffer) {
    g_mega_trace_bwd = (unsigned long long*)device_buffer;
    return 0;
}

// tuning: 1 = the persistent kernels of rounds 3-5 (every wave in every phase), 2 = fixed roles + register-carried weight stages
SWAT_API int swat_vit_set_mega_roles(int mode) {
    SWAT_REQUIRE(mode == 1 || mode == 2, "swat_vit_set_mega_roles: mode must be 1 or 2");
    g_mega_roles = mode;
    return 0;
}

// tuning: XCDs a team's members are spread over under round-robin placement (1, 2 or 4; speed only)
SWAT_API int swat_vit_set_mega_xsplit(int xcds) {
    SWAT_REQUIRE(xcds == 1 || xcds == 2 || xcds == 4 || xcds == 8, "swat_vit_set_mega_xsplit: 1, 2, 4 or 8");
    g_mega_xsplit = xcds;
    return 0;
}

SWAT_API int swat_vit_set_mega(int mode) {
    SWAT_REQUIRE(mode >= 0 && mode <= 2, "swat_vit_set_mega: mode must be 0, 1 or 2");
    g_mega_mode = mode;
    return 0;
}

// status word of the persistent forwards since the last call (synchronises the device): 0 = every hand-off completed
SWAT_API int swat_vit_mega_status(void) {
    MegaDev* md = mega_dev();
    if (!md || !md->state) return 0;
    unsigned v = 0;
    SWAT_HIP(hipDeviceSynchronize());
    SWAT_HIP(hipMemcpy(&v, md->state + 16 * 32, 4, hipMemcpyDeviceToHost));
    if (v) {
        SWAT_HIP(hipMemset(md->state + 16 * 32, 0, 128));
        md->ctr_dirty = true;   // a launch gave up: the team counters are short of what the host has counted
    }
    md->host_flag[0] = 0;
    md->host_flag[1] = 0;
    return (int)(v & 0x3fffffffu) | (v ? 0x40000000 : 0);
}

// no synchronisation: 0, or the 1-based number (swat_vit_mega_launches) of the first persistent launch that gave up on a hand-off
SWAT_API int swat_vit_mega_poll(void) {
    MegaDev* md = mega_dev();
    if (!md || !md->host_flag) return 0;
    return (int)__atomic_load_n(md->host_flag, __ATOMIC_RELAXED);
}

SWAT_API int swat_vit_mega_launches(void) {
    MegaDev* md = mega_dev();
    return md ? (int)md->launches : 0;
}

// tuning / tests: in each of the next `launches` persistent forwards one team member skips its arrival at hand-off number
// `handoff` (1-based, 6 per block) and the spin bound is 2 ms — the launch gives up exactly as if that member were not resident
SWAT_API int swat_vit_mega_inject_failure(int launches, int handoff) {
    SWAT_REQUIRE(launches >= 0 && handoff >= 0, "swat_vit_mega_inject_failure: negative argument");
    g_fail_launches = launches;
    g_fail_sync = handoff;
    return 0;
}

// tuning: per-workgroup phase stamps (64 x uint64 per workgroup, 100 MHz ticks) of the next persistent forwards; null = off
SWAT_API int swat_vit_mega_trace(void* device_buffer) {
    g_mega_trace = (unsigned long long*)device_buffer;
    return 0;
}
