// swat_amd/csrc/gemm.hip — bf16 MFMA GEMM for gfx950 with fused epilogues; the workhorse of path A.
//
// Replaces the cuBLAS calls behind open_clip's nn.Linear / nn.MultiheadAttention / conv1 forward and
// their autograd backward (ref: model.encode_image call sites utils/training.py:332,1474).
//
//   C[M,N] = op(A) * op(B)      fp32 accumulate in v_mfma_f32_16x16x32_bf16
//   layout NT: A[M,K] (k contiguous)  B[N,K] (k contiguous)      forward   Y = X W^T
//   layout NN: A[M,K] (k contiguous)  B[K,N] (n contiguous)      dgrad     dX = dY W
//   layout TN: A[K,M] (m contiguous)  B[K,N] (n contiguous)      wgrad     dW = dY^T X
//
// 256 (or 512) threads = 4 (8) waves, block tile BM x BN, 64-deep k-steps.  Three kernels share ONE LDS image:
//   * gemm_dma_kernel   (K % 64 == 0): tiles go global -> LDS by global_load_lds_dwordx4 (no VGPR staging, no
//     ds_write), NSTAGE buffers, NSTAGE-1 stages in flight behind a counted s_waitcnt vmcnt(N) + raw s_barrier;
//     every address loop-invariant.  The finetuning GEMMs are small (M = B*T = 1600, ~1-3 workgroups per CU):
//     latency has to be hidden by prefetch depth inside the wave, not by occupancy.
//   * gemm_ring2_kernel (K % 64 == 0): the same loop on a register ring (global_load_dwordx4 -> ds_write_b128).
//   * gemm_ring_kernel  (any K): register ring with zero-filled K tails and batch strides (attention's T-deep
//     products for T > 64, ViT-L/14's 588-wide patch rows, ragged shapes).
// LDS image (LINEAR, because an LDS-DMA instruction writes wave-uniform base + lane*16; the bank-conflict swizzle
// is applied to the per-lane SOURCE address and again on the read):
//   k-contiguous tile [rows][64]: 16-B slot t of row r lives at slot t ^ ((r >> 1) & 7); fragments are ONE
//       ds_read_b128 per lane (k = 32 ks + 8 g + j, g = lane >> 4) — conflict-free for b128's 16-lane groups.
//   k-strided tile [64][R]: 32-B column group c of k-row k lives at group c ^ f(k); fragments are two
//       ds_read_b64_tr_b16 (hardware transpose) — f() makes the 8 k-rows a half-wave touches hit distinct banks.
//   (Two ds_read_b64 per fragment would be fused by hipcc into ds_read2st64_b64, whose 32-bank / 4x16-lane
//   grouping conflicts heavily on these images: measured 15x slower LDS phase.  Hence b128 + tr_b64 only.)
// The MFMA is issued as mfma(Bfrag, Afrag) so each lane ends up with 4 CONSECUTIVE n of one row m: epilogue
// loads/stores are 8-16 B per lane.
#include <hip/hip_ext.h>
#include "kernels.hpp"
#include <stdio.h>
#include <stdlib.h>
#include "gemm_device.hpp"
#include "../../include/swat_hip.h"
#include "../../include/swat_hip_tuning.h"

namespace swat {

template <int FM, int FN, bool A_KC, bool B_KC, int BM, int BN, bool NOMFMA = false>
__device__ __forceinline__ void mma_stage(const unsigned char* sa, const unsigned char* sb, int wm0, int wn0, int lane,
                                          f32x4_t (&acc)[FM][FN]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = read_frag<A_KC, BM>(sa, wm0 + 16 * i, ks, lane);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = read_frag<B_KC, BN>(sb, wn0 + 16 * j, ks, lane);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                if constexpr (NOMFMA) acc[i][j][0] += (float)af[i][0] + (float)bfr[j][0];
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
}

// --------------------------------------------------------------------------------------------
// register-ring kernel (any K, zero-filled K tail): D stages of global loads are kept in flight in a
// register ring (16 VGPRs per stage for a 64x64 tile), so memory latency is covered by prefetch DEPTH while the
// per-step issue cost stays at global_load_dwordx4 + ds_write_b128 (~25 cycles per KiB vs ~100 measured for an
// LDS-DMA piece issued by the computing wave itself: rocprofv3 SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY, profiles/).
// Plain loads stay in flight across __syncthreads() (it only drains LDS ops when no LDS-DMA is pending).
// --------------------------------------------------------------------------------------------
template <int LAYOUT, int BM, int BN, int D>
__global__ __launch_bounds__(256) void gemm_ring_kernel(const GemmArgs p) {
    constexpr bool A_KC = (LAYOUT != L_TN);
    constexpr bool B_KC = (LAYOUT == L_NT);
    constexpr int BK = 64;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int A_CH = A_BYTES / 16 / 256, B_CH = B_BYTES / 16 / 256;

    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM;
    const int wn0 = (wave & 1) * WN;
    int bx, by;
    tile_coords(p, bx, by);
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int z = blockIdx.z;
    const int zo = z / p.inner, zi = z % p.inner;
    const unsigned short* __restrict__ A = p.A + zo * p.sA0 + zi * p.sA1;
    const unsigned short* __restrict__ B = p.B + zo * p.sB0 + zi * p.sB1;
    const int M = p.M, N = p.N, K = p.K;

    // per-thread chunk geometry is loop-invariant: precompute source pointers / LDS offsets once
    const unsigned short* srcA[A_CH];
    const unsigned short* srcB[B_CH];
    int ldsA[A_CH], ldsB[B_CH], kA[A_CH], kB[B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int c = tid + 256 * i;
        if constexpr (A_KC) {
            const int row = c >> 3, t = c & 7;
            int gr = m0 + row; gr = gr < M ? gr : M - 1;
            srcA[i] = A + (int64_t)gr * p.lda + 8 * t; kA[i] = 8 * t; ldsA[i] = kc_off(row, t);
        } else {
            const int krow = c / (BM / 8), c16 = c % (BM / 8);
            int gr = m0 + 8 * c16; gr = gr + 8 <= p.lda ? gr : 0;
            srcA[i] = A + (int64_t)krow * p.lda + gr; kA[i] = krow; ldsA[i] = st_off<BM>(krow, c16);
        }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int c = tid + 256 * i;
        if constexpr (B_KC) {
            const int row = c >> 3, t = c & 7;
            int gr = n0 + row; gr = gr < N ? gr : N - 1;
            srcB[i] = B + (int64_t)gr * p.ldb + 8 * t; kB[i] = 8 * t; ldsB[i] = kc_off(row, t);
        } else {
            const int krow = c / (BN / 8), c16 = c % (BN / 8);
            int gr = n0 + 8 * c16; gr = gr + 8 <= p.ldb ? gr : 0;
            srcB[i] = B + (int64_t)krow * p.ldb + gr; kB[i] = krow; ldsB[i] = st_off<BN>(krow, c16);
        }
    }
    const int64_t stepA = A_KC ? 64 : (int64_t)64 * p.lda;
    const int64_t stepB = B_KC ? 64 : (int64_t)64 * p.ldb;

    auto load_chunk = [&](const unsigned short* src, bool kcontig, int kk, int k0) -> s16x8_t {
        s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        const int gk = k0 + kk;
        if (gk < K) {
            v = *reinterpret_cast<const s16x8_t*>(src);
            if (kcontig && gk + 8 > K) {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (gk + e >= K) v[e] = 0;
            }
        }
        return v;
    };
    s16x8_t ring[D][A_CH + B_CH];
    auto load_stage = [&](s16x8_t (&r)[A_CH + B_CH], int step) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) r[i] = load_chunk(srcA[i] + step * stepA, A_KC, kA[i], step * BK);
#pragma unroll
        for (int i = 0; i < B_CH; ++i) r[A_CH + i] = load_chunk(srcB[i] + step * stepB, B_KC, kB[i], step * BK);
    };
    auto store_stage = [&](const s16x8_t (&r)[A_CH + B_CH], int buf) {
        unsigned char* sa = smem + buf * (A_BYTES + B_BYTES);
        unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *reinterpret_cast<s16x8_t*>(sa + ldsA[i]) = r[i];
#pragma unroll
        for (int i = 0; i < B_CH; ++i) *reinterpret_cast<s16x8_t*>(sb + ldsB[i]) = r[A_CH + i];
    };

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (K + BK - 1) / BK;
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nk) load_stage(ring[s], s);
    store_stage(ring[0], 0);
    if (D < nk) load_stage(ring[0], D);
    __syncthreads();
    for (int t0 = 0; t0 < nk; t0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int t = t0 + u;
            if (t < nk) {
                constexpr int dummy = 0; (void)dummy;
                const int nu = (u + 1) % D;  // static after unrolling: ring slot of step t+1
                if (t + 1 < nk) {
                    store_stage(ring[nu], (t + 1) & 1);  // the other buffer: its readers passed the last barrier
                    if (t + 1 + D < nk) load_stage(ring[nu], t + 1 + D);
                }
                const unsigned char* sa = smem + (t & 1) * (A_BYTES + B_BYTES);
                mma_stage<FM, FN, A_KC, B_KC, BM, BN>(sa, sa + A_BYTES, wm0, wn0, lane, acc);
                __syncthreads();
            }
        }
    }
    gemm_epilogue<FM, FN>(p, acc, m0 + wm0, n0 + wn0, lane, zo, zi);
}

// --------------------------------------------------------------------------------------------
// LDS-DMA pipelined kernel (K % 64 == 0)
// --------------------------------------------------------------------------------------------
template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// --------------------------------------------------------------------------------------------
// LDS-DMA kernel, lean main loop (K % 64 == 0): everything that does not
// change from step to step is hoisted: the per-lane source offsets are 32-bit and added to a wave-uniform base
// pointer that advances in SGPRs (saddr + voffset addressing), the main loop is unrolled by NSTAGE so every LDS
// stage address is an immediate, and the fragment read addresses are precomputed per lane.  rocprofv3 showed the
// first version issuing ~95 VALU + 60 SALU per 8 MFMAs (SQ_ACTIVE_INST_ANY 38 %, MFMA busy 17 %): the loop was
// bound by its own address arithmetic, not by memory, LDS or the matrix pipe.
// --------------------------------------------------------------------------------------------
// tuning only (exp bit 16): per-workgroup timeline {start, main loop done, end} in 100 MHz ticks + HW_ID / XCC_ID
__device__ unsigned long long* g_gemm_trace = nullptr;  // also used by gemm_pc.hip
__device__ __forceinline__ void trace_block(const GemmArgs& p, unsigned long long t0, unsigned long long t1) {
    if (!(p.exp & 16) || threadIdx.x != 0 || !g_gemm_trace) return;
    const unsigned long long t2 = wall_clock64();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    g_gemm_trace[4 * b + 0] = t0; g_gemm_trace[4 * b + 1] = t1; g_gemm_trace[4 * b + 2] = t2;
    g_gemm_trace[4 * b + 3] = ((unsigned long long)xcc << 32) | hw;
}

template <int LAYOUT, int BM, int BN, int NSTAGE, int WGM = 2, int WGN = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_dma_kernel(const GemmArgs p) {
    const unsigned long long tr0 = (p.exp & 16) ? wall_clock64() : 0ull;
    constexpr int NW = WGM * WGN;
    constexpr bool A_KC = (LAYOUT != L_TN);
    constexpr bool B_KC = (LAYOUT == L_NT);
    constexpr int BK = 64;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_INS = A_BYTES / 1024 / NW, B_INS = B_BYTES / 1024 / NW;
    static_assert(A_INS >= 1 && B_INS >= 1, "tile too small for this many waves");
    constexpr int LPS = A_INS + B_INS;
    constexpr int NA = A_KC ? 1 : 2, NB = B_KC ? 1 : 2;  // LDS reads per fragment

    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WGN) * WM;
    const int wn0 = (wave % WGN) * WN;
    int bx, by;
    tile_coords(p, bx, by);
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int z = blockIdx.z;
    const int zo = z / p.inner, zi = z % p.inner;
    const int M = p.M, N = p.N, K = p.K;

    // ---- loop-invariant per-lane SOURCE byte offsets (32-bit) relative to wave-uniform bases ----
    auto src_off = [&](bool kcontig, int ins, int row0, int R, int ld, int tile_rows) -> unsigned {
        const int byte = ins * 1024 + lane * 16;
        if (kcontig) {
            const int row = byte >> 7, ps = (byte >> 4) & 7;
            const int t = ps ^ ((row >> 1) & 7);
            int gr = row0 + row; gr = gr < R ? gr : R - 1;
            return (unsigned)((gr - row0) * ld + 8 * t) * 2u;   // relative to the tile's first row
        } else {
            const int rs = tile_rows * 2;
            const int k = byte / rs, pc16 = (byte % rs) >> 4;
            const int f = tile_rows == 128 ? st_f<128>(k) : st_f<64>(k);
            const int c16 = (((pc16 >> 1) ^ f) << 1) | (pc16 & 1);
            int gc = row0 + 8 * c16; gc = gc + 8 <= ld ? gc : 0;
            return (unsigned)(k * ld + gc) * 2u;                 // relative to row k0 of the matrix
        }
    };
    unsigned offA[A_INS], offB[B_INS];
#pragma unroll
    for (int i = 0; i < A_INS; ++i) offA[i] = src_off(A_KC, wave + NW * i, m0, M, p.lda, BM);
#pragma unroll
    for (int i = 0; i < B_INS; ++i) offB[i] = src_off(B_KC, wave + NW * i, n0, N, p.ldb, BN);
    // wave-uniform base pointers of step 0 (k-contiguous: first row of the tile; strided: row 0, column 0)
    const char* gA = reinterpret_cast<const char*>(p.A + zo * p.sA0 + zi * p.sA1 + (A_KC ? (int64_t)m0 * p.lda : 0));
    const char* gB = reinterpret_cast<const char*>(p.B + zo * p.sB0 + zi * p.sB1 + (B_KC ? (int64_t)n0 * p.ldb : 0));
    const int64_t stepA = A_KC ? 128 : (int64_t)128 * p.lda;  // bytes per 64-deep k-step
    const int64_t stepB = B_KC ? 128 : (int64_t)128 * p.ldb;

    // ---- loop-invariant per-lane fragment READ offsets inside a stage ----
    const int r16 = lane & 15, g = lane >> 4;
    unsigned ra[2][FM][NA], rb[2][FN][NB];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (A_KC) ra[ks][i][0] = kc_off(wm0 + 16 * i + r16, 4 * ks + g);
            else {
                const int k = 32 * ks + 8 * g + (r16 >> 2), c32 = (wm0 + 16 * i) >> 4;
                ra[ks][i][0] = k * (BM * 2) + ((c32 ^ st_f<BM>(k)) << 5) + 8 * (r16 & 3);
                ra[ks][i][NA - 1] = (k + 4) * (BM * 2) + ((c32 ^ st_f<BM>(k + 4)) << 5) + 8 * (r16 & 3);
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if constexpr (B_KC) rb[ks][j][0] = A_BYTES + kc_off(wn0 + 16 * j + r16, 4 * ks + g);
            else {
                const int k = 32 * ks + 8 * g + (r16 >> 2), c32 = (wn0 + 16 * j) >> 4;
                rb[ks][j][0] = A_BYTES + k * (BN * 2) + ((c32 ^ st_f<BN>(k)) << 5) + 8 * (r16 & 3);
                rb[ks][j][NB - 1] = A_BYTES + (k + 4) * (BN * 2) + ((c32 ^ st_f<BN>(k + 4)) << 5) + 8 * (r16 & 3);
            }
        }
    }
    auto frag = [&](const unsigned char* st, const unsigned* off, bool kcontig) -> bf16x8_t {
        if (kcontig) return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const s16x8_t*>(st + off[0]));
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(st + off[0]));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(st + off[1]));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };

    auto issue = [&](int stage, const char* a, const char* b) {
        unsigned char* sa = smem + stage * STAGE;
        unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_INS; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + offA[i]),
                                             (__attribute__((address_space(3))) void*)(sa + (wave + NW * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_INS; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + offB[i]),
                                             (__attribute__((address_space(3))) void*)(sb + (wave + NW * i) * 1024), 16, 0, 0);
    };

    // bias gradient beside a weight gradient: column sums of A = A^T 1, one MFMA against a ones fragment per k-step
    const bool do_cs = LAYOUT == L_TN && p.colsum && n0 == 0 && wn0 == 0;
    const s16x8_t ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_bits);
    f32x4_t accs[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) accs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    // k-order skew: blocks that share an operand panel start at different k-steps (and wrap), so at any instant they
    // pull different 128-B columns of the panel — with row strides of 1.5-6 KiB a common k-step lands every CU on
    // the same few L2 channels.  The sum order per block changes (still deterministic run to run).
    int kk = (p.exp & 8) ? 0 : (bx * 5 + by * 3) % nk;
    gA += kk * stepA; gB += kk * stepB;
    auto advance = [&]() {
        gA += stepA; gB += stepB;
        if (++kk == nk) { kk = 0; gA -= nk * stepA; gB -= nk * stepB; }
    };
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
        if (s < nk) issue(s, gA, gB);
        advance();
    }
    // gA / gB now point at the next stage to issue

    for (int t0 = 0; t0 < nk; t0 += NSTAGE) {
#pragma unroll
        for (int u = 0; u < NSTAGE; ++u) {
            const int t = t0 + u;
            if (t < nk) {
                const int younger = nk - 1 - t;  // issued stages younger than t: min(younger, NSTAGE-2)
                if (younger >= NSTAGE - 2) wait_vmcnt<(NSTAGE - 2) * LPS>();  // steady state
                else switch (younger) {                                        // drain at the tail
                    case 6: wait_vmcnt<(NSTAGE > 8 ? 6 : 0) * LPS>(); break;
                    case 5: wait_vmcnt<(NSTAGE > 7 ? 5 : 0) * LPS>(); break;
                    case 4: wait_vmcnt<(NSTAGE > 6 ? 4 : 0) * LPS>(); break;
                    case 3: wait_vmcnt<(NSTAGE > 5 ? 3 : 0) * LPS>(); break;
                    case 2: wait_vmcnt<(NSTAGE > 4 ? 2 : 0) * LPS>(); break;
                    case 1: wait_vmcnt<(NSTAGE > 3 ? 1 : 0) * LPS>(); break;
                    default: wait_vmcnt<0>(); break;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (t + NSTAGE - 1 < nk && !(p.exp & 1)) issue((u + NSTAGE - 1) % NSTAGE, gA, gB);
                advance();
                const unsigned char* st = smem + u * STAGE;  // immediate after unrolling
                if (p.exp & 2) continue;  // tuning only: DMA streaming without LDS reads / MFMA
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8_t af[FM], bfr[FN];
#pragma unroll
                    for (int i = 0; i < FM; ++i) af[i] = frag(st, ra[ks][i], A_KC);
#pragma unroll
                    for (int j = 0; j < FN; ++j) bfr[j] = frag(st, rb[ks][j], B_KC);
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                    if (do_cs) {
#pragma unroll
                        for (int i = 0; i < FM; ++i) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accs[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    const unsigned long long tr1 = (p.exp & 16) ? wall_clock64() : 0ull;
    if (!(p.exp & 4)) gemm_epilogue<FM, FN>(p, acc, m0 + wm0, n0 + wn0, lane, zo, zi);
    if (do_cs && (lane >> 4) == 0) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm0 + 16 * i + (lane & 15);
            if (m < p.M) p.colsum[m] = accs[i][0];
        }
    }
    if (p.exp & 16) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trace_block(p, tr0, tr1);
    }
}

// --------------------------------------------------------------------------------------------
// lean register-ring kernel (K % 64 == 0, K >= 64 * D): plain global_load_dwordx4 (cheap to issue: measured LDS-DMA
// pieces cost the issuing wave ~100 cycles each, ~25 GB/s per wave) into a D-deep register ring, ds_write_b128 into
// two LDS buffers; all addresses loop-invariant (32-bit lane offsets + SGPR base, immediates for the LDS side); the
// steady-state loop is branch-free so hipcc keeps counted vmcnt waits.
// --------------------------------------------------------------------------------------------
template <int LAYOUT, int BM, int BN, int D>
__global__ __launch_bounds__(256) void gemm_ring2_kernel(const GemmArgs p) {
    static_assert(D % 2 == 0, "even ring depth keeps the LDS buffer index static");
    constexpr bool A_KC = (LAYOUT != L_TN);
    constexpr bool B_KC = (LAYOUT == L_NT);
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_CH = A_BYTES / 16 / 256, B_CH = B_BYTES / 16 / 256;
    constexpr int NA = A_KC ? 1 : 2, NB = B_KC ? 1 : 2;

    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM;
    const int wn0 = (wave & 1) * WN;
    int bx, by;
    tile_coords(p, bx, by);
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int z = blockIdx.z;
    const int zo = z / p.inner, zi = z % p.inner;
    const int M = p.M, N = p.N, K = p.K;

    unsigned offA[A_CH], offB[B_CH], ldsA[A_CH], ldsB[B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int c = tid + 256 * i;
        if constexpr (A_KC) {
            const int row = c >> 3, t = c & 7;
            int gr = m0 + row; gr = gr < M ? gr : M - 1;
            offA[i] = (unsigned)((gr - m0) * p.lda + 8 * t) * 2u; ldsA[i] = kc_off(row, t);
        } else {
            const int krow = c / (BM / 8), c16 = c % (BM / 8);
            int gc = m0 + 8 * c16; gc = gc + 8 <= p.lda ? gc : 0;
            offA[i] = (unsigned)(krow * p.lda + gc) * 2u; ldsA[i] = st_off<BM>(krow, c16);
        }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int c = tid + 256 * i;
        if constexpr (B_KC) {
            const int row = c >> 3, t = c & 7;
            int gr = n0 + row; gr = gr < N ? gr : N - 1;
            offB[i] = (unsigned)((gr - n0) * p.ldb + 8 * t) * 2u; ldsB[i] = A_BYTES + kc_off(row, t);
        } else {
            const int krow = c / (BN / 8), c16 = c % (BN / 8);
            int gc = n0 + 8 * c16; gc = gc + 8 <= p.ldb ? gc : 0;
            offB[i] = (unsigned)(krow * p.ldb + gc) * 2u; ldsB[i] = A_BYTES + st_off<BN>(krow, c16);
        }
    }
    const char* gA = reinterpret_cast<const char*>(p.A + zo * p.sA0 + zi * p.sA1 + (A_KC ? (int64_t)m0 * p.lda : 0));
    const char* gB = reinterpret_cast<const char*>(p.B + zo * p.sB0 + zi * p.sB1 + (B_KC ? (int64_t)n0 * p.ldb : 0));
    const int64_t stepA = A_KC ? 128 : (int64_t)128 * p.lda;
    const int64_t stepB = B_KC ? 128 : (int64_t)128 * p.ldb;

    const int r16 = lane & 15, g = lane >> 4;
    unsigned ra[2][FM][NA], rb[2][FN][NB];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (A_KC) ra[ks][i][0] = kc_off(wm0 + 16 * i + r16, 4 * ks + g);
            else {
                const int k = 32 * ks + 8 * g + (r16 >> 2), c32 = (wm0 + 16 * i) >> 4;
                ra[ks][i][0] = k * (BM * 2) + ((c32 ^ st_f<BM>(k)) << 5) + 8 * (r16 & 3);
                ra[ks][i][NA - 1] = (k + 4) * (BM * 2) + ((c32 ^ st_f<BM>(k + 4)) << 5) + 8 * (r16 & 3);
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if constexpr (B_KC) rb[ks][j][0] = A_BYTES + kc_off(wn0 + 16 * j + r16, 4 * ks + g);
            else {
                const int k = 32 * ks + 8 * g + (r16 >> 2), c32 = (wn0 + 16 * j) >> 4;
                rb[ks][j][0] = A_BYTES + k * (BN * 2) + ((c32 ^ st_f<BN>(k)) << 5) + 8 * (r16 & 3);
                rb[ks][j][NB - 1] = A_BYTES + (k + 4) * (BN * 2) + ((c32 ^ st_f<BN>(k + 4)) << 5) + 8 * (r16 & 3);
            }
        }
    }
    auto frag = [&](const unsigned char* st, const unsigned* off, bool kcontig) -> bf16x8_t {
        if (kcontig) return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const s16x8_t*>(st + off[0]));
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(st + off[0]));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(st + off[1]));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };

    // bias gradient beside a weight gradient: column sums of A = A^T 1, one MFMA against a ones fragment per k-step
    const bool do_cs = LAYOUT == L_TN && p.colsum && n0 == 0 && wn0 == 0;
    const s16x8_t ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_bits);
    f32x4_t accs[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) accs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    s16x8_t ring[D][A_CH + B_CH];
    auto load_slot = [&](s16x8_t (&r)[A_CH + B_CH]) {  // loads the step gA/gB point at, then advances them
#pragma unroll
        for (int i = 0; i < A_CH; ++i) r[i] = *reinterpret_cast<const s16x8_t*>(gA + offA[i]);
#pragma unroll
        for (int i = 0; i < B_CH; ++i) r[A_CH + i] = *reinterpret_cast<const s16x8_t*>(gB + offB[i]);
        gA += stepA; gB += stepB;
    };
    auto store_slot = [&](const s16x8_t (&r)[A_CH + B_CH], int buf) {
        unsigned char* st = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *reinterpret_cast<s16x8_t*>(st + ldsA[i]) = r[i];
#pragma unroll
        for (int i = 0; i < B_CH; ++i) *reinterpret_cast<s16x8_t*>(st + ldsB[i]) = r[A_CH + i];
    };
    auto compute = [&](int buf) {
        const unsigned char* st = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = frag(st, ra[ks][i], A_KC);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = frag(st, rb[ks][j], B_KC);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < FM; ++i) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accs[i], 0, 0, 0);
            }
        }
    };

    const int nk = K >> 6;  // >= D (checked by the launcher)
#pragma unroll
    for (int s = 0; s < D; ++s) load_slot(ring[s]);  // steps 0 .. D-1
    store_slot(ring[0], 0);
    int next = D;  // next step to load
    if (next < nk) { load_slot(ring[0]); ++next; }
    __syncthreads();

    int t = 0;
    // steady state: every step stores step t+1 and loads a new one; no branches
    for (; t + D <= nk - D - 1 + 1 && next + D - 1 < nk; t += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            store_slot(ring[(u + 1) % D], (u + 1) & 1);
            load_slot(ring[(u + 1) % D]);
            compute(u & 1);
            __syncthreads();
        }
        next += D;
    }
    // tail: same schedule with guards (t is a multiple of D here, so slots / buffers stay static per u)
    for (; t < nk; t += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            if (t + u < nk) {
                if (t + u + 1 < nk) {
                    store_slot(ring[(u + 1) % D], (u + 1) & 1);
                    if (next < nk) { load_slot(ring[(u + 1) % D]); ++next; }
                }
                compute(u & 1);
                __syncthreads();
            }
        }
    }
    if (!(p.exp & 4)) gemm_epilogue<FM, FN>(p, acc, m0 + wm0, n0 + wn0, lane, zo, zi);
    if (do_cs && (lane >> 4) == 0) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm0 + 16 * i + (lane & 15);
            if (m < p.M) p.colsum[m] = accs[i][0];
        }
    }
}

static int g_exp = getenv("SWAT_GEMM_EXP") ? atoi(getenv("SWAT_GEMM_EXP")) : 0;  // tuning bits (swat_gemm_set_exp; the variable: whole-job A/B runs)
static unsigned long long* g_trace_host = nullptr;
static int g_swz_mode = -1;  // -1 automatic, 0 off (swat_gemm_set_swizzle)

// grid + XCD-aware tile order: per-XCD working set = (#distinct row tiles) * BM * K + (#distinct col tiles) * BN * K
dim3 plan_grid(GemmArgs& b, int bm, int bn, int batch) {
    const int tm = (b.M + bm - 1) / bm, tn = (b.N + bn - 1) / bn;
    b.tiles_m = tm; b.tiles_n = tn; b.swz = 0; b.exp = g_exp; b.trace = g_trace_host;
    if (batch != 1 || tm * tn < 16 || g_swz_mode == 0) return dim3(tn, tm, batch);
    const int per = (tm * tn + 7) / 8;
    auto ws = [&](int fast, int slow, int fsz, int ssz) {  // `per` consecutive tiles, `fast` dimension first
        const int nf = per < fast ? per : fast;
        const int ns = (per + fast - 1) / fast + (per % fast ? 1 : 0);
        return (int64_t)nf * fsz + (int64_t)(ns < slow ? ns : slow) * ssz;
    };
    const int64_t m_first = ws(tm, tn, bm, bn), n_first = ws(tn, tm, bn, bm);
    b.swz = m_first <= n_first ? 1 : 2;
    return dim3(tm * tn, 1, 1);
}

// tile configurations (swat_gemm_set_tile pins one; pick_cfg chooses): 14-19 / 25-28 / 43 LDS-DMA ring, 30-34 register
// ring (whole k-steps), 20-24 register ring with K tails
template <int LAYOUT>
static int launch_gemm(const GemmArgs& a, int batch, int cfg, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
#define SWAT_RING(BM_, BN_, D_)                                                                           \
    {                                                                                                     \
        GemmArgs b = a;                                                                                   \
        dim3 grid = plan_grid(b, BM_, BN_, batch);                                                        \
        if (ev0) hipExtLaunchKernelGGL((gemm_ring_kernel<LAYOUT, BM_, BN_, D_>), grid, dim3(256), 0, s, ev0, ev1, 0, b); \
        else hipLaunchKernelGGL((gemm_ring_kernel<LAYOUT, BM_, BN_, D_>), grid, dim3(256), 0, s, b);           \
    }
#define SWAT_DMA(BM_, BN_, NS_)                                                                           \
    {                                                                                                     \
        GemmArgs b = a;                                                                                   \
        dim3 grid = plan_grid(b, BM_, BN_, batch);                                                        \
        if (ev0) hipExtLaunchKernelGGL((gemm_dma_kernel<LAYOUT, BM_, BN_, NS_>), grid, dim3(256), 0, s, ev0, ev1, 0, b); \
        else hipLaunchKernelGGL((gemm_dma_kernel<LAYOUT, BM_, BN_, NS_>), grid, dim3(256), 0, s, b);           \
    }
#define SWAT_RING2(BM_, BN_, D_)                                                                          \
    {                                                                                                     \
        GemmArgs b = a;                                                                                   \
        dim3 grid = plan_grid(b, BM_, BN_, batch);                                                        \
        if (ev0) hipExtLaunchKernelGGL((gemm_ring2_kernel<LAYOUT, BM_, BN_, D_>), grid, dim3(256), 0, s, ev0, ev1, 0, b); \
        else hipLaunchKernelGGL((gemm_ring2_kernel<LAYOUT, BM_, BN_, D_>), grid, dim3(256), 0, s, b);          \
    }
#define SWAT_DMA8(BM_, BN_, NS_, WGM_, WGN_)                                                              \
    {                                                                                                     \
        GemmArgs b = a;                                                                                   \
        dim3 grid = plan_grid(b, BM_, BN_, batch);                                                        \
        if (ev0) hipExtLaunchKernelGGL((gemm_dma_kernel<LAYOUT, BM_, BN_, NS_, WGM_, WGN_>), grid, dim3(512), 0, s, ev0, ev1, 0, b); \
        else hipLaunchKernelGGL((gemm_dma_kernel<LAYOUT, BM_, BN_, NS_, WGM_, WGN_>), grid, dim3(512), 0, s, b); \
    }
    if (cfg == 43 && a.K % 64 != 0) cfg = 20;
    if (cfg >= 25 && cfg <= 28 && a.K % 64 != 0) cfg = 20;
    if (cfg >= 30 && (a.K % 64 != 0 || a.K < 64 * 6)) cfg = 20;  // lean ring needs whole, >= D k-steps
    if (cfg >= 14 && cfg < 20 && a.K % 64 != 0) cfg = 20;  // the LDS-DMA kernel has no K tail
    switch (cfg) {
        case 43: SWAT_DMA(128, 64, 2) break;
        case 14: SWAT_DMA(64, 64, 3) break;
        case 15: SWAT_DMA(128, 128, 3) break;
        case 16: SWAT_DMA(128, 64, 3) break;
        case 17: SWAT_DMA(64, 64, 4) break;
        case 18: SWAT_DMA(64, 64, 6) break;
        case 19: SWAT_DMA(64, 64, 8) break;
        case 20: SWAT_RING(64, 64, 4) break;
        case 25: SWAT_DMA8(128, 128, 3, 2, 4) break;
        case 26: SWAT_DMA8(128, 128, 4, 2, 4) break;
        case 27: SWAT_DMA8(128, 64, 4, 4, 2) break;
        case 28: SWAT_DMA8(256, 128, 2, 4, 2) break;
        case 30: SWAT_RING2(64, 64, 4) break;
        case 31: SWAT_RING2(64, 64, 2) break;
        case 32: SWAT_RING2(128, 64, 4) break;
        case 33: SWAT_RING2(128, 128, 2) break;
        case 34: SWAT_RING2(64, 64, 6) break;
        case 21: SWAT_RING(128, 128, 3) break;
        case 22: SWAT_RING(128, 64, 4) break;
        case 23: SWAT_RING(64, 64, 6) break;
        case 24: SWAT_RING(64, 64, 2) break;
        default: SWAT_RING(64, 64, 4) break;
    }
#undef SWAT_RING
#undef SWAT_RING2
#undef SWAT_DMA
#undef SWAT_DMA8
    return check_launch("swat_gemm_bf16");
}

static int g_force_cfg = -1;  // swat_gemm_set_tile(): experiments / tests pin one configuration

static int pick_cfg_base(int layout, const GemmArgs& a, int batch);
static int pick_cfg(int layout, const GemmArgs& a, int batch) {
    // tuning (SWAT_PC_MAP="53:60,54:61"): replace a chosen configuration by another one inside the step
    static int map[128], init = 0;
    if (!init) {
        for (int i = 0; i < 128; ++i) map[i] = i;
        if (const char* e = getenv("SWAT_PC_MAP")) {
            int from, to, n = 0;
            while (sscanf(e, "%d:%d%n", &from, &to, &n) == 2) {
                if (from >= 0 && from < 128) map[from] = to;
                e += n;
                if (*e == ',') ++e;
            }
        }
        init = 1;
    }
    const int c = pick_cfg_base(layout, a, batch);
    return c >= 0 && c < 128 ? map[c] : c;
}
static int pick_cfg_base(int layout, const GemmArgs& a, int batch) {
    if (g_force_cfg >= 0) return g_force_cfg;
    // Measured on MI355X (rocprofv3 kernel durations, tools/exp_trace.py; M = B*T = 1600).  The launches are short
    // (10-30 us) and run as lock-step rounds of resident workgroups (tools/block_timeline.py), each round paying its
    // own epilogue burst: wide outputs (N >= 2048, K <= 1024) want ONE round, which 128x64 tiles with a 2-stage ring
    // (48 KB: three workgroups per CU, 468-624 tiles resident at once) give (43); elsewhere 64x64 tiles (more
    // workgroups in flight) win: LDS-DMA (14) for the k-contiguous NT products and the large wgrads, the lean
    // register ring (30) for the deep-K dgrad / small wgrad products; everything ragged goes to the staged ring (20).
    const bool whole = a.K % 64 == 0;
    // split-K planes of a deep product with a narrow output (c_proj forward, the c_fc / in_proj dgrads: N = 768, K = 2304 /
    // 3072): 128 x 128 tiles over <= 4 planes fill the chip with half the operand bytes per flop of the 96 x 64 tiling
    if (whole && batch >= 2 && batch <= 4 && layout != L_TN && !(g_exp & 512) && a.K >= 256) {
        const int t128 = ((a.M + 127) / 128) * ((a.N + 127) / 128) * batch;
        if (t128 > 160 && t128 <= 256) return 50;
    }
    if (!whole || batch > 1) return a.K % 64 == 0 && batch > 1 ? 14 : 20;
    // One-round grids go to the producer / consumer kernel (gemm_pc.hip: one large tile per CU, loader waves + consumer
    // waves, LDS-staged epilogue).  Measured per launch with the step's own epilogues (tools/perf_gemm.py, M = 1600):
    // c_fc forward 17.1 vs 18.8 us, c_proj dgrad 17.0 vs 19.0 (160 x 128: 240 tiles); the N = 768 products 9.4 / 19.9 /
    // 17.9 / 7.3 / 14.5 vs 12.6 / 24.7 / 23.8 / 8.4 / 19.3 us (96 x 64: 204 tiles); c_fc / c_proj weight gradients 15.8 /
    // 16.6 vs 21.3 us (96 x 96: 256 tiles); out_proj / in_proj weight gradients 9.7 / 14.0 vs 10.6 / 17.2 us (64 x 64).
    if (!(g_exp & 512) && a.K >= 256) {
        auto tiles = [&](int bm, int bn) { return ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
        if (layout != L_TN) {
            if (a.N >= 2048 && a.K <= 1024) {
                if (tiles(160, 128) > 192 && tiles(160, 128) <= 256) return 51;
            } else if (tiles(96, 64) > 128 && tiles(96, 64) <= 256) return 53;
        } else {
            if (tiles(96, 96) > 192 && tiles(96, 96) <= 256) return 54;
            if (tiles(64, 64) >= 128 && tiles(64, 64) <= 512) return 56;
        }
    }
    // thousands of rows (ViT-B/16: 6 304, ViT-L/14: 8 224 at B = 32) fill several rounds with any tile: there the
    // 128-wide tiles' lower operand traffic wins the deep-K products (register ring 128x128 / 128x64) and the wgrads
    const bool many_rows = (layout == L_TN ? a.K : a.M) >= 4096;
    // forward / data-gradient products with thousands of rows: the 256 x 256 tile (gemm_big.hip) runs its main loop matrix-core
    // bound (~50 % of the MFMA peak per CU, the guide template's own figure), but a ViT-L/14 product is only 1.5-3 rounds of such
    // tiles, so tile quantisation and the 64 K-element epilogue per workgroup eat it.  Measured per launch alone
    // (tools/perf_gemm_big.py, M = 8 224, d = 1 024): in_proj 73 vs 77 us, c_proj 110 vs 114, c_fc dgrad 99 vs 102, in_proj dgrad
    // 77 vs 81 — but out_proj 45 vs 37, c_fc 173 vs 135, c_proj dgrad 177 vs 150; at M = 6 304 it loses everywhere.  Inside
    // the ViT-L/14 step even the four winning shapes cost 1.5 % (968 vs 982 img/s: a 512-thread, 128-KiB workgroup leaves the
    // weight-gradient lane nothing to overlap with).  So it is OPT-IN (swat_gemm_set_exp bit 2048 / SWAT_GEMM_EXP=2048).
    // tens of thousands of rows (mined-feature extraction encodes 1024 images / captions per batch: M = 51 200 / 78 848): the
    // 256 x 256 tile's quantisation loss is gone (1 800+ tiles) and it wins every product but out_proj.  Measured per launch
    // (tools/perf_gemm.py, M = 51 200, d = 768; the previous choice in brackets): in_proj 234 us = 775 TF (303), c_fc 387 (426),
    // c_proj 292 = 827 TF (332); M = 78 848, d = 512: in_proj 187 (284), c_proj 223 (280).
    // 256 x 128 tiles with loader / consumer waves, persistent (gemm_xl.hip, configuration 90): whole rounds of the chip.  Measured per
    // launch with the step's epilogues (tools/perf_gemm.py, round 4; the previous choice in brackets) — M = 8 224, d = 1 024 (32 tile
    // rows + a 32-row remainder launch): in_proj 61 us = 843 TF (78), out_proj 34 (39), c_fc 101 (116), c_proj 83 = 836 TF (96),
    // c_proj dgrad 98 (127), c_fc dgrad 76 = 910 TF (86), out_proj dgrad 28 (29), in_proj dgrad 60 = 868 TF (69): 0.93 of what the
    // vendor library needs for the eight plain products.  It LOSES where the tile grid leaves a round mostly empty (M = 6 304, d = 768:
    // 144 / 432 / 576 tiles), so it is taken only when the last round is at least 90 % full.
    if (many_rows && layout != L_TN && !(g_exp & (512 | 8192)) && a.inner == 1 && a.N % 128 == 0 && a.K >= 512 && a.M >= 2048) {
        const int64_t t = (int64_t)(a.M / 256) * (a.N / 128), rounds = (t + 255) / 256;
        const bool big80 = a.M >= 32768 && layout == L_NT && (a.N >= 1024 || a.K >= 2048) && a.N % 256 == 0 && !a.actin;   // (the 256 x 256 tile below)
        GemmArgs m = a;
        m.M = a.M / 256 * 256;
        if (t >= 256 && t * 10 >= rounds * 256 * 9 && !big80 && !a.rpg && !a.rowbias && gemm_xl_serves(layout, m)) return 90;
    }
    if (a.M >= 32768 && layout == L_NT && !(g_exp & 512) && (a.N >= 1024 || a.K >= 2048) && a.N % 256 == 0 && a.inner == 1 &&
        a.lda % 8 == 0 && a.ldb % 8 == 0 && !a.actin && a.K >= 512)
        return 80;
    if ((g_exp & 2048) && many_rows && layout != L_TN && a.M >= 8192 && a.N >= 1024 && a.N % 256 == 0 && a.inner == 1 && a.lda % 8 == 0 &&
        a.ldb % 8 == 0 && !a.C2 && !a.actin && (a.K >= 2048 || (a.K >= 1024 && a.N == 3 * a.K)))
        return 80;
    // many rows onto a NARROW output (N <= 1 536: out_proj / c_proj forward, every data gradient that lands on the residual width):
    // the producer / consumer 160 x 128 tile over several rounds.  Measured per launch (tools/perf_gemm_big.py, round 3; the
    // previous choices 50 / 30 / 14 in brackets) — M = 8 224, d = 1 024: c_proj 92 us = 749 TF (113), c_fc dgrad 84 = 821 TF (104),
    // out_proj dgrad 28 (36), in_proj dgrad 67 = 772 TF (81); M = 6 304, d = 768: 42 (59), 35 = 843 TF (53), 14 (19), 28 = 809 TF (41).
    // The forward out_proj (fp32 residual epilogue, K = d) and every wide product (N >= 2 048) keep their tiles: 51 loses there.
    if (many_rows && layout != L_TN && !(g_exp & 512) && a.N <= 1536 && a.K >= 256 && (a.K >= 2048 || layout == L_NN)) return 51;
    if (layout == L_NT) {
        if (a.N >= 2048 && a.K <= 1024 && a.M >= 128) return 43;
        return many_rows && a.K >= 2048 ? 50 : 14;  // (50 = producer / consumer 128 x 128, several rounds: ViT-L/14 step 34.3 ->
    }                                               //  32.3 ms, ViT-B/16 9.39 -> 9.05 ms against the register rings 33 / 32)
    if (layout == L_NN) {
        if (a.N >= 2048 && a.K <= 1024 && a.M >= 128) return 43;
        if (many_rows && a.K >= 2048) return 50;
        return a.K >= 384 ? 30 : 14;  // (re-checked inside the overlapped step: 43 / 30 stay the best pair)
    }
    // weight gradients run on the backward's second stream beside the data-gradient chain: alone the LDS-DMA 64x64
    // kernel is the fastest for the large ones (21 vs 22 us), but in the step the 128x128 register ring disturbs the
    // main chain least (3.65 vs 3.79 ms/step; 15 / 16 / 30 / 32 in between) — half the operand traffic through L2
    // (the same holds at 6 304 / 8 224 rows: ViT-L/14 40.1 ms/step with 33 vs 43.3 with the 128x64 LDS-DMA tile)
    if ((int64_t)a.M * a.N >= (1 << 21)) return a.K >= 384 ? 50 : 14;
    return a.K >= 384 ? 30 : 14;
}

// ---- optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream around every GEMM ----
struct GemmProf {
    bool on = false;
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;  // 2 per launch
    int* kind = nullptr;       // layout * 3 + (0: 64-row tile, 1: 128-row tile, 2: producer / consumer kernel)
    double* flops = nullptr;
};
static GemmProf g_prof;

// other kernel files time their launches with the same facility: -> true and the start / stop events of the next slot
bool gemm_prof_take(int kind, double flops, hipEvent_t& ev0, hipEvent_t& ev1) {
    if (!(g_prof.on && g_prof.n < g_prof.cap)) return false;
    ev0 = g_prof.ev[2 * g_prof.n];
    ev1 = g_prof.ev[2 * g_prof.n + 1];
    g_prof.kind[g_prof.n] = kind;
    g_prof.flops[g_prof.n] = flops;
    ++g_prof.n;
    return true;
}

int gemm_dispatch(int layout, const GemmArgs& a, int batch, hipStream_t s) {
    int tile = pick_cfg(layout, a, batch);
    if (tile >= 50 && (batch > 4 || a.K % 64 != 0)) tile = (a.K % 64 == 0) ? 14 : 20;  // the producer / consumer kernel: whole k-steps, <= 4 planes
    if (a.colsum) {
        const bool lean = tile >= 50 || tile == 14 || (tile >= 15 && tile <= 19) || (tile >= 25 && tile <= 28) || (tile >= 30 && tile <= 34) || tile == 43;
        SWAT_REQUIRE(layout == L_TN && batch == 1 && lean && a.K % 64 == 0 && a.K >= 384,
                     "gemm: fused column sums need a TN product on one of the lean kernels (tile %d)", tile);
    }
    const bool prof = g_prof.on && g_prof.n < g_prof.cap;
    // profiling: the events ride on the kernel's own dispatch packet (hipExtLaunchKernel), so their difference is the
    // kernel's execution time as rocprofv3 reports it, without the ~2-3 us each separately recorded event adds
    hipEvent_t ev0 = prof ? g_prof.ev[2 * g_prof.n] : nullptr, ev1 = prof ? g_prof.ev[2 * g_prof.n + 1] : nullptr;
    int rc;
    if (tile == 90) {
        // 256 x 128 tiles, loader / consumer waves, persistent (gemm_xl.hip): whole tiles only — the rows beyond the last multiple of
        // 256 (every fused epilogue is row-local) go to a second, small launch
        const int Mmain = layout == L_TN ? a.M : a.M / 256 * 256;
        GemmArgs m = a;
        m.M = Mmain;
        const bool ok = batch == 1 && a.inner == 1 && Mmain >= 256 && !a.rpg && !a.rowbias && gemm_xl_serves(layout, m) &&
                        (a.M == Mmain || ((int64_t)Mmain * a.ldc) % 8 == 0);
        if (ok) {
            // up to 32 remainder rows ride in the launch's tail phase (gemm_xl.hip); more (or shapes the scheme does not take) go to a
            // second launch below
            int tail = 0;
            if (a.M > Mmain && a.M - Mmain <= 32 && !(g_exp & 32768)) {
                GemmArgs r = a;
                r.M = a.M - Mmain;
                r.A = a.A + (int64_t)Mmain * a.lda;
                if (gemm_skinny_serves(layout, r)) tail = a.M - Mmain;
            }
            rc = launch_gemm_xl(layout, &m, 1, s, ev0, ev1, g_exp, g_trace_host, tail ? &tail : nullptr);
            if (prof) {
                g_prof.kind[g_prof.n] = layout == L_TN ? 10 : 11;
                g_prof.flops[g_prof.n] = 2.0 * Mmain * a.N * a.K;
                ++g_prof.n;
            }
            if (rc == 0 && a.M > Mmain && !tail) {
                GemmArgs r = a;
                r.M = a.M - Mmain;
                r.A = a.A + (int64_t)Mmain * a.lda;
                const int64_t co = (int64_t)Mmain * a.ldc;
                if (a.C) r.C = a.out_f32 ? (void*)((float*)a.C + co) : (void*)((unsigned short*)a.C + co);
                if (a.C2) r.C2 = (unsigned short*)a.C2 + co;
                if (a.residual) r.residual = a.residual + co;
                if (a.actin) r.actin = a.actin + (int64_t)Mmain * a.ldact;
                rc = (g_exp & 16384) ? 1 : launch_gemm_skinny(layout, r, s);   // one wave per 16 columns (gemm_skinny.hip)
                if (rc == 1) {
                    const int keep = g_force_cfg;
                    g_force_cfg = -1;
                    rc = gemm_dispatch(layout, r, 1, s);
                    g_force_cfg = keep;
                }
            }
            return rc;
        }
        tile = layout == L_TN ? 50 : (a.K % 64 == 0 ? 14 : 20);
    }
    // (the 256 x 256 kernel forms its per-thread operand offsets in 32 bits: the whole operands must stay below 4 GiB)
    if (tile == 80 && (layout == L_TN || batch != 1 || a.inner != 1 || a.K % 64 != 0 || a.K < 128 || a.lda % 8 || a.ldb % 8 ||
                       (int64_t)a.M * a.lda * 2 >= (1ll << 32) || (int64_t)(layout == L_NT ? a.N : a.K) * a.ldb * 2 >= (1ll << 32)))
        tile = layout == L_TN ? 50 : 14;
    if (tile == 80) rc = launch_gemm_big(layout, a, s, ev0, ev1);  // 256 x 256, matrix-core bound (gemm_big.hip)
    else if (tile >= 70 && tile <= 72 && batch == 1 && a.inner == 1) rc = launch_gemm_pc_group(layout, &a, 1, s, ev0, ev1, tile);  // persistent tile loop
    else if (tile >= 70) { set_error("gemm: persistent configuration %d needs one unbatched product", tile); rc = -1; }
    else if (tile >= 50) rc = launch_gemm_pc(layout, a, batch, tile, s, ev0, ev1);
    else if (layout == L_NT) rc = launch_gemm<L_NT>(a, batch, tile, s, ev0, ev1);
    else if (layout == L_NN) rc = launch_gemm<L_NN>(a, batch, tile, s, ev0, ev1);
    else rc = launch_gemm<L_TN>(a, batch, tile, s, ev0, ev1);
    if (prof) {
        g_prof.kind[g_prof.n] = layout * 3 + (tile >= 50 && tile != 80 ? 2 : 0) + (tile == 80) + (tile == 15 || tile == 16 || tile == 21 || tile == 22 || tile == 32 || tile == 33 || (tile >= 25 && tile <= 28) || tile == 43);
        g_prof.flops[g_prof.n] = 2.0 * a.M * a.N * a.K * batch;
        ++g_prof.n;
    }
    return rc;
}

// n products of one layout (the weight gradients of a transformer block): ONE persistent launch over the joint tile list
// when every product qualifies for the producer / consumer kernel's grouped form, else one launch each.
static int g_wgrad_all = getenv("SWAT_WGRAD_ALL") ? atoi(getenv("SWAT_WGRAD_ALL")) : 1;   // all blocks' weight gradients behind the persistent backward: 1 = one launch, 0 = one per block
static int g_wgrad_big = getenv("SWAT_WGRAD_BIG") ? atoi(getenv("SWAT_WGRAD_BIG")) : 1;   // swat_gemm_set_wgrad_big: 0 = never, 1 = when the joint list is one round of 256 x 128 tiles, 2 = always

// the weight gradients of a block as ONE round of 256 x 128 tiles (gemm_wgrad.hip): taken when the joint tile list fills most of
// the chip once — more tiles than CUs would run a second, mostly empty round at a whole tile's latency each.
// -> 0 launched, 1 not taken (the caller goes on with its own route), < 0 error
// all weight gradients of a tower's blocks (up to 64 products) as one persistent launch over the joint list of 256 x 128 tiles
int wgrad_all_try(const GemmArgs* probs, int n, hipStream_t s, const XlAdam* adam) {
    if (!g_wgrad_big || g_wgrad_all == 0 || n < 1 || n > 64 || g_force_cfg >= 0 || (g_exp & 4096)) return 1;
    const bool prof = g_prof.on && g_prof.n < g_prof.cap;
    hipEvent_t ev0 = prof ? g_prof.ev[2 * g_prof.n] : nullptr, ev1 = prof ? g_prof.ev[2 * g_prof.n + 1] : nullptr;
    const int rc = launch_gemm_xl_table(probs, n, s, ev0, ev1, adam);
    if (prof && rc == 0) {
        g_prof.kind[g_prof.n] = 10;
        double fl = 0;
        for (int i = 0; i < n; ++i) fl += 2.0 * probs[i].M * probs[i].N * probs[i].K;
        g_prof.flops[g_prof.n] = fl;
        ++g_prof.n;
    }
    return rc;
}

int wgrad_big_try(const GemmArgs* probs, int n, hipStream_t s, bool dry_run) {
    if (!g_wgrad_big || n < 1 || n > 8 || g_force_cfg >= 0) return 1;
    int tiles = 0;
    bool deep = true;
    for (int i = 0; i < n; ++i) {
        tiles += ((probs[i].M + 255) / 256) * ((probs[i].N + 127) / 128);
        deep = deep && probs[i].K >= 512 && probs[i].M >= 256 && probs[i].N >= 128;
    }
    // (more tiles than one round: the persistent form walks them; taken while the last round is at least half full — ViT-L/14's
    // block is 384 tiles = 1.5 rounds and still 4 % faster in the step than four launches of 128 x 128 tiles)
    const int rounds = (tiles + 255) / 256;
    if (g_wgrad_big != 2 && !(deep && tiles >= 144 && tiles * 4 >= rounds * 256 * 3)) return 1;
    if (dry_run) {   // would launch_wgrad_group take these shapes?
        for (int i = 0; i < n; ++i) {
            const GemmArgs& g = probs[i];
            if (g.K % 64 || g.K < 64 || g.M % 8 || g.N % 8 || g.lda % 8 || g.ldb % 8 || g.ldc % 4 || !g.out_f32 ||
                (int64_t)g.K * g.lda * 2 >= (1ll << 31) || (int64_t)g.K * g.ldb * 2 >= (1ll << 31))
                return 1;
        }
        return 0;
    }
    const bool prof = g_prof.on && g_prof.n < g_prof.cap;
    hipEvent_t ev0 = prof ? g_prof.ev[2 * g_prof.n] : nullptr, ev1 = prof ? g_prof.ev[2 * g_prof.n + 1] : nullptr;
    int rc = 1;
    if (!(g_exp & 4096)) rc = launch_gemm_xl(L_TN, probs, n, s, ev0, ev1, g_exp, g_trace_host);   // split-role form (gemm_xl.hip)
    if (rc == 1) rc = launch_wgrad_group(probs, n, s, ev0, ev1, g_exp, g_trace_host);             // shapes it does not serve / exp 4096: gemm_wgrad.hip
    if (prof && rc == 0) {
        g_prof.kind[g_prof.n] = 10;
        double fl = 0;
        for (int i = 0; i < n; ++i) fl += 2.0 * probs[i].M * probs[i].N * probs[i].K;
        g_prof.flops[g_prof.n] = fl;
        ++g_prof.n;
    }
    return rc;
}

int gemm_group_dispatch(int layout, const GemmArgs* probs, int n, hipStream_t s) {
    if (layout == L_TN) {
        const int rc = wgrad_big_try(probs, n, s);
        if (rc <= 0) return rc;
    }
    bool ok = layout == L_TN && n >= 2 && n <= 64 && !(g_exp & 512) && g_force_cfg < 0;
    for (int i = 0; ok && i < n; ++i) {
        const GemmArgs& a = probs[i];
        ok = a.K % 64 == 0 && a.K >= 256 && a.out_f32 && a.N % 4 == 0 && a.ldc % 4 == 0 && !a.actin && !a.C2 && a.inner == 1 &&
             (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0;
    }
    if (!ok) {
        for (int i = 0; i < n; ++i)
            if (int rc = gemm_dispatch(layout, probs[i], 1, s)) return rc;
        return 0;
    }
    const bool prof = g_prof.on && g_prof.n < g_prof.cap;
    hipEvent_t ev0 = prof ? g_prof.ev[2 * g_prof.n] : nullptr, ev1 = prof ? g_prof.ev[2 * g_prof.n + 1] : nullptr;
    const int rc = launch_gemm_pc_group(layout, probs, n, s, ev0, ev1);
    if (prof) {
        g_prof.kind[g_prof.n] = layout * 3 + 2;
        double fl = 0;
        for (int i = 0; i < n; ++i) fl += 2.0 * probs[i].M * probs[i].N * probs[i].K;
        g_prof.flops[g_prof.n] = fl;
        ++g_prof.n;
    }
    return rc;
}

}  // namespace swat

SWAT_API int swat_gemm_set_trace(void* dev_buf) {  /* tuning only: 4 x uint64 per workgroup, see trace_block() */
    unsigned long long* ptr = (unsigned long long*)dev_buf;
    swat::g_trace_host = ptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(swat::g_gemm_trace), &ptr, sizeof(ptr)) == hipSuccess ? 0 : -1;
}

SWAT_API int swat_gemm_set_wgrad_big(int mode) {
    swat::g_wgrad_big = mode < 0 ? 0 : mode > 2 ? 2 : mode;
    return 0;
}

SWAT_API int swat_gemm_set_exp(int e) {
    swat::g_exp = e;
    return 0;
}

SWAT_API int swat_gemm_set_swizzle(int mode) {
    swat::g_swz_mode = mode;
    return 0;
}

SWAT_API int swat_gemm_set_tile(int cfg) {
    swat::g_force_cfg = cfg;
    return 0;
}

// Start timing every GEMM launch (up to max_launches) with hipEvents on its own stream; not graph-capturable.
SWAT_API int swat_gemm_prof_begin(int max_launches) {
    using swat::g_prof;
    if (g_prof.ev) {
        for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
        delete[] g_prof.ev; delete[] g_prof.kind; delete[] g_prof.flops;
    }
    g_prof.cap = max_launches; g_prof.n = 0;
    g_prof.ev = new hipEvent_t[2 * max_launches];
    g_prof.kind = new int[max_launches];
    g_prof.flops = new double[max_launches];
    for (int i = 0; i < 2 * max_launches; ++i) SWAT_HIP(hipEventCreate(&g_prof.ev[i]));
    g_prof.on = true;
    return 0;
}

// Stop, synchronise the events and report per kind — 16 entries: layout * 3 + {64-row tile, 128-row / 256-row tile, producer /
// consumer} for NT / NN / TN, then 9 = the persistent forward (csrc/vit_mega.hip: all blocks' products, LayerNorms and
// attention in one launch), 10 = a block's weight gradients as one launch of 256 x 128 tiles, 11 = the 256 x 128 loader /
// consumer kernel on NT / NN products, 12 = the persistent backward (data gradients of all blocks), 13-15 unused: total ms,
// launches, flops.
SWAT_API int swat_gemm_prof_end(double* ms_by_kind, int64_t* count_by_kind, double* flops_by_kind) {
    using swat::g_prof;
    g_prof.on = false;
    for (int k = 0; k < 16; ++k) { ms_by_kind[k] = 0; count_by_kind[k] = 0; flops_by_kind[k] = 0; }
    for (int i = 0; i < g_prof.n; ++i) {
        SWAT_HIP(hipEventSynchronize(g_prof.ev[2 * i + 1]));
        float ms = 0.f;
        SWAT_HIP(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        ms_by_kind[g_prof.kind[i]] += ms;
        count_by_kind[g_prof.kind[i]] += 1;
        flops_by_kind[g_prof.kind[i]] += g_prof.flops[i];
    }
    const int n = g_prof.n;
    g_prof.n = 0;
    return n;
}

// Generic entry (tests, attention as batched GEMMs).  See include/swat_hip.h.
SWAT_API int swat_gemm_bf16(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb,
                            int ldc, int batch, int inner, int64_t sA0, int64_t sA1, int64_t sB0, int64_t sB1,
                            int64_t sC0, int64_t sC1, const float* bias, const float* residual, const void* actin,
                            int ldact, int act, void* C2, int out_f32, float alpha, swat_stream_t stream) {
    SWAT_REQUIRE(layout >= 0 && layout <= 2, "swat_gemm_bf16: bad layout %d", layout);
    SWAT_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && inner > 0, "swat_gemm_bf16: bad shape");
    SWAT_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "swat_gemm_bf16: lda/ldb must be multiples of 8, ldc of 4");
    SWAT_REQUIRE(ldc >= ((N + 3) & ~3), "swat_gemm_bf16: ldc must cover N rounded up to 4");
    swat::GemmArgs a{};
    a.A = (const unsigned short*)A; a.B = (const unsigned short*)B; a.C = C; a.C2 = C2;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.inner = inner; a.sA0 = sA0; a.sA1 = sA1; a.sB0 = sB0; a.sB1 = sB1; a.sC0 = sC0; a.sC1 = sC1;
    a.bias = bias; a.residual = residual; a.actin = (const unsigned short*)actin; a.ldact = ldact; a.act = act;
    a.out_f32 = out_f32; a.alpha = alpha; a.rpg = 0; a.gstride = 0; a.roff = 0; a.rowbias = nullptr;
    SWAT_REQUIRE(C || (C2 && !out_f32), "swat_gemm_bf16: C may be null only with a bf16 C2 (activated output only)");
    if (!C) { a.C = C2; a.c2_only = 1; }
    return swat::gemm_dispatch(layout, a, batch, (hipStream_t)stream);
}
