// swat_amd/csrc/gemm_wgrad.hip — ALL weight gradients of a transformer block as ONE launch of 256 x 128 tiles.
//
// ref: loss.backward() — utils/training.py:1486-1488 (train_cutmix), :340-343 (train_ce): the dW = dY^T X products of the
// four nn.Linear layers of an open_clip residual block (in_proj, out_proj, c_fc, c_proj), reduction over the B*T rows.
//
// Why.  At B = 32 (M = 1 600 rows, padded to 1 664) the four products of a block are 22.6 GFLOP with a deep reduction
// (26 k-tiles of 64) and small outputs (768 x 768 ... 3072 x 768).  Launched one by one with tiles chosen to fill the chip in
// one round each (96 x 96, 64 x 64: csrc/gemm_pc.hip) they cost 12-18 us apiece = 65 us per block: a 96 x 96 tile needs
// 0.021 operand bytes per flop from L2 and the per-CU L2 -> LDS path (~45 B/clk) bounds it at a third of the MFMA rate, plus
// 6-7 us of fixed cost per launch.  The four outputs together are 7.08 M elements = 216 tiles of
// 256 x 128: ONE resident round of the chip, 0.012 bytes per flop (48 KB per 64-deep k-tile against 256 MFMAs per wave pair),
// one dispatch, one prologue.  The joint tile list is what makes the big tile possible: alone, no product of the block has
// enough tiles of that size to fill the chip.
//
// Structure: 8 waves, 4 (M) x 2 (N); a wave owns 64 x 64 = 4 x 4 accumulator fragments of v_mfma_f32_16x16x32_bf16.  Both
// operands are k-strided ([rows][features] row-major, the reduction runs over rows): LDS image [64 k][256] for dY and
// [64 k][128] for X in the power-of-two-part swizzle of gemm_device.hpp (stp_off / stp_inv), filled by
// global_load_lds_dwordx4 with the permutation on the per-lane SOURCE address, read with ds_read_b64_tr_b16.  Ring of three
// 48 KB stages: two k-tiles in flight behind a counted vmcnt, ONE raw s_barrier per k-tile; the fragments of a k-tile's second
// 32-deep half are read while the MFMAs of its first half issue, those of the next tile's first half behind the barrier while
// the second half's MFMAs issue.  The bias gradients that go with c_fc and in_proj (column sums of dY) come from one extra MFMA
// against a ones fragment per half step in the n-tile-0 workgroups, as in the lean TN kernels.
#include <hip/hip_ext.h>
#include "gemm_device.hpp"
#include "../../include/swat_hip.h"

namespace swat {

constexpr int WG_BM = 256, WG_BN = 128, WG_NS = 3;
constexpr int WG_A_BYTES = 64 * WG_BM * 2, WG_B_BYTES = 64 * WG_BN * 2, WG_STAGE = WG_A_BYTES + WG_B_BYTES;
constexpr int WG_PIECES = WG_STAGE / 1024, WG_LPS = WG_PIECES / 8;   // 48 LDS-DMA pieces per stage, 6 per wave
constexpr int WG_MAXP = 8;

struct WgProb {
    const unsigned short* A;   // dY [K rows][M features], ld = lda
    const unsigned short* B;   // X  [K rows][N features], ld = ldb
    float* C;                  // dW [M][N] fp32, ld = ldc
    float* colsum;             // db [M] = column sums of dY, or null
    int M, N, K, lda, ldb, ldc;
    int tiles_m, tile0;        // tiles along M; first tile id of this product in the joint list
};
struct WgArgs {
    WgProb p[WG_MAXP];
    int n, total;
    unsigned long long* trace;   // ablation build, exp bit 16: 8 x uint64 stamps per workgroup (100 MHz)
    int exp;   // ablation build only (swat_gemm_set_exp): 1 no DMA after the prologue, 2 no MFMA, 4 no epilogue, 256 no fragment reads
};

__device__ __forceinline__ bf16x8_t wg_lds_tr(unsigned lo_addr, unsigned hi_addr) {
    s16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(lo_addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(hi_addr));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void wg_pin(bf16x8_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void wg_bar() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wg_wait_reads() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <bool ABL>
__global__ __launch_bounds__(512) void wgrad_group_kernel(const WgArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WG_NS * WG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    auto stamp = [&](int i) {
        if constexpr (ABL) {
            if ((a.exp & 16) && a.trace && tid == 0) a.trace[(size_t)blockIdx.x * 8 + i] = wall_clock64();
        }
    };
    stamp(0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // joint tile list, one contiguous chunk per XCD (blocks b, b + 8, .. share an XCD: speed only); inside a product tiles walk M
    // first, so the tiles an XCD runs at the same time share the X panel of their n-tile and a few dY panels
    int id;
    {
        const int lb = blockIdx.x, xcd = lb & 7, q = a.total >> 3, r = a.total & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lb >> 3);
    }
    int pi = 0;
#pragma unroll
    for (int i = 1; i < WG_MAXP; ++i)
        if (i < a.n && id >= a.p[i].tile0) pi = i;
    const WgProb& P = a.p[pi];
    const int t = id - P.tile0;
    const int tm = t % P.tiles_m, tn = t / P.tiles_m;
    const int m0 = tm * WG_BM, n0 = tn * WG_BN;
    const int M = P.M, N = P.N, nk = P.K >> 6;
    const int r16 = lane & 15, g = lane >> 4;

    // ---- this wave's LDS-DMA pieces: pieces wave, wave + 8, .. of a stage; pieces 0-31 = dY image, 32-47 = X image ------------
    unsigned off[WG_LPS];   // source byte offsets relative to the operand's k-tile base
#pragma unroll
    for (int i = 0; i < WG_LPS; ++i) {
        const int piece = wave + 8 * i;
        int k, col;
        if (piece < WG_A_BYTES / 1024) {
            stp_inv<WG_BM>(piece * 1024 + lane * 16, k, col);
            int gc = m0 + col;
            gc = gc + 8 <= M ? gc : 0;   // columns beyond the matrix: any readable address (their products are never stored)
            off[i] = (unsigned)(k * P.lda + gc) * 2u;
        } else {
            stp_inv<WG_BN>(piece * 1024 - WG_A_BYTES + lane * 16, k, col);
            int gc = n0 + col;
            gc = gc + 8 <= N ? gc : 0;
            off[i] = (unsigned)(k * P.ldb + gc) * 2u;
        }
    }
    const char* gA = reinterpret_cast<const char*>(P.A);
    const char* gB = reinterpret_cast<const char*>(P.B);
    const int64_t stepA = (int64_t)128 * P.lda, stepB = (int64_t)128 * P.ldb;   // bytes per 64-row k-tile
    auto issue = [&](int slot, int kt) {
        unsigned char* dst = smem + slot * WG_STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < WG_LPS; ++i) {
            const char* src = (wave + 8 * i < WG_A_BYTES / 1024 ? gA + kt * stepA : gB + kt * stepB) + off[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, 0, 0);
        }
    };

    // ---- fragment read addresses inside stage 0 ---------------------------------------------------------------------------------
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
    unsigned ra[4][2][2], rb[4][2][2];   // [fragment][ks][lo / hi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k = 32 * ks + 8 * g + (r16 >> 2);
            const unsigned sub = 8 * (r16 & 3);
            ra[i][ks][0] = lds0 + sub + stp_off<WG_BM>(k, 64 * wr + 16 * i);
            ra[i][ks][1] = lds0 + sub + stp_off<WG_BM>(k + 4, 64 * wr + 16 * i);
            rb[i][ks][0] = lds0 + WG_A_BYTES + sub + stp_off<WG_BN>(k, 64 * wc + 16 * i);
            rb[i][ks][1] = lds0 + WG_A_BYTES + sub + stp_off<WG_BN>(k + 4, 64 * wc + 16 * i);
        }

    f32x4_t acc[4][4], accs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_cs = P.colsum != nullptr && tn == 0 && wc == 0;
    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    bf16x8_t fa0[4], fb0[4], fa1[4], fb1[4];   // fragments of the k-tile's first / second 32-deep half
    if constexpr (ABL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa0[i] = fb0[i] = fa1[i] = fb1[i] = ones;
    }
    auto read_half = [&](bf16x8_t (&fa)[4], bf16x8_t (&fb)[4], unsigned so, int ks) {
        if constexpr (ABL) {
            if (a.exp & 256) return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = wg_lds_tr(rb[i][ks][0] + so, rb[i][ks][1] + so);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = wg_lds_tr(ra[i][ks][0] + so, ra[i][ks][1] + so);
    };
    auto mfma_half = [&](bf16x8_t (&fa)[4], bf16x8_t (&fb)[4]) {
        if constexpr (ABL) {
            if (a.exp & 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { wg_pin(fb[j]); wg_pin(fa[j]); }
                return;
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j) wg_pin(fb[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wg_pin(fa[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (do_cs) {
#pragma unroll
            for (int i = 0; i < 4; ++i) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], accs[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: k-tiles 0 .. 2 on their way, tile 0 landed, its first half in registers ------------------------------------
#pragma unroll
    for (int s = 0; s < WG_NS; ++s)
        if (s < nk) issue(s, s);
    if (nk >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (nk == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(1);
    wg_bar();
    stamp(2);
    read_half(fa0, fb0, 0u, 0);

    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned so = (unsigned)slot * WG_STAGE;
        const int nslot = slot + 1 == WG_NS ? 0 : slot + 1;
        wg_wait_reads();                 // first half of tile kt (issued one MFMA block ago)
        read_half(fa1, fb1, so, 1);      // second half: lands while the first half's MFMAs issue
        mfma_half(fa0, fb0);
        // tile kt + 1 must have landed behind the barrier; tile kt + 2 (this wave's 6 youngest pieces) may still fly
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_wait_reads();                 // every wave is done reading stage `slot` before anyone refills it
        wg_bar();
        if (kt + WG_NS < nk && !(ABL && (a.exp & 1))) issue(slot, kt + WG_NS);
        if (kt + 1 < nk) read_half(fa0, fb0, (unsigned)nslot * WG_STAGE, 0);
        mfma_half(fa1, fb1);
        slot = nslot;
    }

    stamp(3);
    // ---- epilogue: fp32 tile, 16-byte stores -------------------------------------------------------------------------------------
    if constexpr (ABL) {
        if (a.exp & 4) return;
    }
    GemmArgs e{};
    e.M = M; e.N = N; e.ldc = P.ldc; e.C = P.C; e.out_f32 = 1; e.alpha = 1.0f;
    gemm_epilogue<4, 4>(e, acc, m0 + 64 * wr, n0 + 64 * wc, lane, 0, 0);
    if (do_cs && g == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 64 * wr + 16 * i + r16;
            if (m < M) P.colsum[m] = accs[i][0];
        }
    }
    if constexpr (ABL) {
        stamp(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(5);
    }
}

// n <= 8 products dW_i = dY_i^T X_i (TN, fp32 outputs) in one launch.  Every product: K % 64 == 0 (pad rows zero), M % 8 == 0,
// N % 8 == 0, 16-byte aligned rows.  -> 0 launched, 1 shapes not served by this kernel (the caller runs them one by one)
int launch_wgrad_group(const GemmArgs* probs, int n, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, int exp, unsigned long long* trace) {
    if (n < 1 || n > WG_MAXP) return 1;
    WgArgs a{};
    a.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const GemmArgs& g = probs[i];
        if (g.K % 64 || g.K < 64 || g.M % 8 || g.N % 8 || g.lda % 8 || g.ldb % 8 || g.ldc % 4 || !g.out_f32 || g.bias || g.residual || g.actin ||
            g.C2 || g.rpg || g.alpha != 1.0f || g.inner > 1)
            return 1;
        if ((int64_t)g.K * g.lda * 2 >= (1ll << 31) || (int64_t)g.K * g.ldb * 2 >= (1ll << 31)) return 1;   // 32-bit piece offsets
        WgProb& p = a.p[i];
        p.A = g.A; p.B = g.B; p.C = (float*)g.C; p.colsum = g.colsum;
        p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc;
        p.tiles_m = (g.M + WG_BM - 1) / WG_BM;
        p.tile0 = total;
        total += p.tiles_m * ((g.N + WG_BN - 1) / WG_BN);
    }
    a.total = total;
    a.exp = exp & (1 | 2 | 4 | 16 | 256);
    a.trace = trace;
    if (a.exp) hipLaunchKernelGGL(wgrad_group_kernel<true>, dim3(total), dim3(512), 0, s, a);
    else if (ev0) hipExtLaunchKernelGGL(wgrad_group_kernel<false>, dim3(total), dim3(512), 0, s, ev0, ev1, 0, a);
    else hipLaunchKernelGGL(wgrad_group_kernel<false>, dim3(total), dim3(512), 0, s, a);
    return check_launch("wgrad_group_kernel");
}

}  // namespace swat
