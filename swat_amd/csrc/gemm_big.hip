// swat_amd/csrc/gemm_big.hip — 256 x 256 bf16 MFMA tile for the many-row products (ViT-B/16: M = 6 304, ViT-L/14: M = 8 224
// rows at B = 32): the forward linears (NT) and their data gradients (NN).  Same contract and fused epilogues as the other
// GEMM kernels (gemm_device.hpp); replaces the cuBLAS calls under open_clip's nn.Linear forward / backward for BASELINE
// config 5 (call sites utils/training.py:332,1474,1486-1488).
//
// Why another tile.  On this chip a CU pulls ~45 B/clk from L2 into LDS whatever the kernel does; a 128 x 128 or 160 x 128
// tile needs 0.014-0.016 operand bytes per flop and is bound by that path (MFMA busy 0.13-0.20, profiles/r02_gemm_pmc.json),
// a 256 x 256 tile needs 0.0078 — 64 KB per 64-deep k-step against 512 MFMAs (2 048 SIMD cycles): matrix-core bound.
// Structure (MI355X guide, 256^2 8-phase template, re-derived for this codebase's LDS images):
//   * 8 waves, 2 (M) x 4 (N): a wave owns 128 x 64 of the tile = 8 x 4 accumulator fragments (128 VGPRs).
//   * LDS 128 KiB = 2 k-tile buffers x {A rows 0-127, A rows 128-255, B columns 0-127, B columns 128-255} x 16 KiB.
//     k-contiguous operands use the [rows][64] image with the 16-B-slot XOR swizzle (kc_off), k-strided ones (B of the NN
//     product: W stored [k][n]) the [64][128] image read with ds_read_b64_tr_b16 (stp_off / stp_inv) — both filled by
//     global_load_lds_dwordx4 with the permutation applied to the per-lane SOURCE address.
//   * a k-tile is four phases, one 64 x 32 quadrant of the wave's tile each (16 MFMAs): fragment reads [12, 4, 8, 4]
//     ds_read_b128 per phase (A sub-tile kept over two phases, B re-read).  Phase = {fragment reads, this phase's share of
//     the NEXT k-tile's LDS-DMA (all of it goes out in phases 0 and 1), wait for the reads, barrier, 16 MFMAs, barrier}.
//   * the two wave rows run half a phase apart (one extra barrier at the start for waves 4-7, one at the end for waves 0-3):
//     while one half issues MFMAs the other half reads fragments, so the two waves of a SIMD do not compete for it.
//   * hazards: reads of a phase are complete (lgkmcnt(0)) before the phase's first barrier, so the buffer a k-tile was read
//     from can be refilled from the next phase on; a k-tile's DMA is waited for (vmcnt(0), two phases after its last piece
//     was issued) in phase 3's read part, before the barrier that precedes the first read of it.
#include <hip/hip_ext.h>
#include "gemm_device.hpp"
#include "../../include/swat_hip.h"

namespace swat {

constexpr int BG = 256;            // tile edge
constexpr int BG_HALF = 16 * 1024;  // one operand half per k-tile
constexpr int BG_BUF = 4 * BG_HALF;

__device__ __forceinline__ bf16x8_t bg_lds_b128(unsigned addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ bf16x8_t bg_lds_tr(unsigned lo_addr, unsigned hi_addr) {
    s16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(lo_addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(hi_addr));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void bg_pin(bf16x8_t& v) { asm volatile("" : "+v"(v)); }

// one 64 x 32 quadrant (RH, CH) of the wave's 128 x 64 tile over a 64-deep k-tile: 16 MFMAs (indices are compile-time: the
// accumulators must stay in registers)
template <int CH>
__device__ __forceinline__ void bg_mfma16(f32x4_t (&acc)[4][4], bf16x8_t (&fa)[4][2], bf16x8_t (&fb)[2][2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bg_pin(fa[i][ks]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bg_pin(fb[j][ks]);
                acc[i][2 * CH + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][ks], fa[i][ks], acc[i][2 * CH + j], 0, 0, 0);
            }
        }
    }
}

template <int LAYOUT>
__global__ __launch_bounds__(512) void gemm_big_kernel(const GemmArgs p) {
    static_assert(LAYOUT == L_NT || LAYOUT == L_NN, "forward (NT) and data-gradient (NN) products");
    constexpr bool B_KC = LAYOUT == L_NT;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BG_BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int bx, by;
    tile_coords(p, bx, by);
    const int m0 = by * BG, n0 = bx * BG;
    const int M = p.M, N = p.N, nk = p.K >> 6;
    const int r16 = lane & 15, g = lane >> 4;

    // ---- this thread's LDS-DMA pieces: per half (16 pieces of 1 KiB) wave w moves pieces w and w + 8 ------------------
    // source byte offsets relative to the operand's k-tile base; destination = half base + piece * 1024 (wave uniform)
    unsigned offA[2][2], offB[2][2];   // [half][piece]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int byte = (wave + 8 * i) * 1024 + lane * 16;
            {   // A: k-contiguous rows
                const int row = byte >> 7, t = ((byte >> 4) & 7) ^ ((row >> 1) & 7);
                int gr = m0 + 128 * h + row;
                gr = gr < M ? gr : M - 1;
                offA[h][i] = (unsigned)(gr * p.lda + 8 * t) * 2u;
            }
            if constexpr (B_KC) {
                const int row = byte >> 7, t = ((byte >> 4) & 7) ^ ((row >> 1) & 7);
                int gr = n0 + 128 * h + row;
                gr = gr < N ? gr : N - 1;
                offB[h][i] = (unsigned)(gr * p.ldb + 8 * t) * 2u;
            } else {
                int k, col;
                stp_inv<128>(byte, k, col);
                int gc = n0 + 128 * h + col;
                gc = gc + 8 <= p.ldb ? gc : 0;   // columns beyond the matrix: any readable address (masked in the epilogue)
                offB[h][i] = (unsigned)(k * p.ldb + gc) * 2u;
            }
        }
    const char* gA = reinterpret_cast<const char*>(p.A);
    const char* gB = reinterpret_cast<const char*>(p.B);
    const int64_t stepB = B_KC ? 128 : (int64_t)128 * p.ldb;  // bytes per k-tile (A: 128)
    auto issue_half = [&](int buf, int which /*0,1 = A halves; 2,3 = B halves*/, int kt) {
        unsigned char* dst = smem + buf * BG_BUF + which * BG_HALF + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* src = which < 2 ? gA + (int64_t)kt * 128 + offA[which][i] : gB + (int64_t)kt * stepB + offB[which - 2][i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, 0, 0);
        }
    };

    // ---- fragment read addresses inside buffer 0 (buffer 1: + BG_BUF) -------------------------------------------------
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
    // A: half wr, sub-tile rh (64 rows): fragment i (16 rows), k-half ks
    unsigned ra[2];   // [ks]; + (64 rh + 16 i) * 128
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) ra[ks] = lds0 + wr * BG_HALF + kc_off(r16, 4 * ks + g);
    // B: half wc >> 1, columns (wc & 1) * 64 + 32 ch + 16 j
    const int bc0 = (wc & 1) * 64;
    unsigned rbk[2];            // k-contiguous B
    unsigned rbs[2][2][2][2];   // k-strided B: [ch][j][ks][lo/hi]
    if constexpr (B_KC) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) rbk[ks] = lds0 + (2 + (wc >> 1)) * BG_HALF + kc_off(bc0 + r16, 4 * ks + g);
    } else {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int k = 32 * ks + 8 * g + (r16 >> 2);
                    const unsigned base = lds0 + (2 + (wc >> 1)) * BG_HALF + 8 * (r16 & 3);
                    rbs[ch][j][ks][0] = base + stp_off<128>(k, bc0 + 32 * ch + 16 * j);
                    rbs[ch][j][ks][1] = base + stp_off<128>(k + 4, bc0 + 32 * ch + 16 * j);
                }
    }

    f32x4_t acc0[4][4], acc1[4][4];   // rows 0-63 / 64-127 of the wave's tile (two arrays: every index stays a compile-time constant)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc0[i][j] = acc1[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa[4][2], fb[2][2];   // A sub-tile: [fragment][ks]; B sub-tile: [fragment][ks]

    auto read_a = [&](unsigned bo, int rh) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[i][ks] = bg_lds_b128(ra[ks] + bo + (64 * rh + 16 * i) * 128);
    };
    auto read_b = [&](unsigned bo, int ch) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (B_KC) fb[j][ks] = bg_lds_b128(rbk[ks] + bo + (32 * ch + 16 * j) * 128);
                else fb[j][ks] = bg_lds_tr(rbs[ch][j][ks][0] + bo, rbs[ch][j][ks][1] + bo);
            }
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto wait_reads = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: k-tile 0 into buffer 0 ------------------------------------------------------------------------
#pragma unroll
    for (int w = 0; w < 4; ++w) issue_half(0, w, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    if (wr == 1) bar();   // the second wave row runs one barrier (half a phase) behind the first

    for (int kt = 0; kt < nk; ++kt) {
        const unsigned bo = (kt & 1) * BG_BUF;
        const int nb = (kt + 1) & 1;
        const bool more = kt + 1 < nk;
        // phase 0: quadrant (0, 0)
        read_b(bo, 0);
        read_a(bo, 0);
        if (more) { issue_half(nb, 0, kt + 1); issue_half(nb, 1, kt + 1); }
        wait_reads();
        bar();
        __builtin_amdgcn_s_setprio(1);
        bg_mfma16<0>(acc0, fa, fb);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // phase 1: quadrant (0, 1)
        read_b(bo, 1);
        if (more) { issue_half(nb, 2, kt + 1); issue_half(nb, 3, kt + 1); }
        wait_reads();
        bar();
        __builtin_amdgcn_s_setprio(1);
        bg_mfma16<1>(acc0, fa, fb);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // phase 2: quadrant (1, 1)
        read_a(bo, 1);
        wait_reads();
        bar();
        __builtin_amdgcn_s_setprio(1);
        bg_mfma16<1>(acc1, fa, fb);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // phase 3: quadrant (1, 0); the next k-tile's DMA (issued two phases ago) must have landed behind this phase's first barrier
        read_b(bo, 0);
        wait_reads();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
        __builtin_amdgcn_s_setprio(1);
        bg_mfma16<0>(acc1, fa, fb);
        __builtin_amdgcn_s_setprio(0);
        bar();
    }
    if (wr == 0) bar();   // barrier counts match again
    gemm_epilogue<4, 4>(p, acc0, m0 + 128 * wr, n0 + 64 * wc, lane, 0, 0);
    gemm_epilogue<4, 4>(p, acc1, m0 + 128 * wr + 64, n0 + 64 * wc, lane, 0, 0);
}

int launch_gemm_big(int layout, const GemmArgs& a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    GemmArgs b = a;
    dim3 grid = plan_grid(b, BG, BG, 1);
    if (layout == L_NT) {
        if (ev0) hipExtLaunchKernelGGL(gemm_big_kernel<L_NT>, grid, dim3(512), 0, s, ev0, ev1, 0, b);
        else hipLaunchKernelGGL(gemm_big_kernel<L_NT>, grid, dim3(512), 0, s, b);
    } else if (layout == L_NN) {
        if (ev0) hipExtLaunchKernelGGL(gemm_big_kernel<L_NN>, grid, dim3(512), 0, s, ev0, ev1, 0, b);
        else hipLaunchKernelGGL(gemm_big_kernel<L_NN>, grid, dim3(512), 0, s, b);
    } else {
        set_error("gemm_big: layout %d is not built (NT and NN only)", layout);
        return -1;
    }
    return check_launch("swat_gemm_bf16 (256 x 256)");
}

}  // namespace swat
