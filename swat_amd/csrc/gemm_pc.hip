// swat_amd/csrc/gemm_pc.hip — producer / consumer bf16 MFMA GEMM for gfx950: ONE large tile per CU.
//
// Same contract as gemm.hip (C = op(A) op(B), layouts NT / NN / TN, the fused epilogues of gemm_device.hpp); replaces the
// cuBLAS calls under open_clip's nn.Linear forward / backward (call sites utils/training.py:332,1474,1486-1488).
//
// Why a second structure.  The finetuning products have M = B*T = 1600 rows: at 64 x 64 tiles the operand traffic
// L2 -> LDS (M N K 2 (1/BM + 1/BN) bytes) is what bounds them (~100 GB/s per CU measured), and a larger tile run by the
// gemm_dma structure — every wave issues LDS-DMA pieces AND computes — serialises DMA issue (~23 cycles per 1 KiB piece,
// the wave is stuck in the issue), fragment reads and MFMAs inside each wave.  Here the roles are split:
//   * NL loader waves only issue global_load_lds_dwordx4 pieces into an NS-deep LDS ring and wait for them with counted
//     s_waitcnt vmcnt(N): their stalls cost nothing.
//   * CM x CN consumer waves (one per SIMD at 2 x 2) only read fragments and issue MFMAs, with the fragments of the next
//     32-deep half step in flight behind the MFMAs of the current one (register double buffer, also across stages).
//   * one raw s_barrier per 64-deep k-step: at barrier t the loaders have seen stages t and t+1 land and the consumers
//     have finished reading stage t-1, whose buffer the loaders refill with stage t+NS-1 right after the barrier.
// One workgroup per CU (up to 160 KiB of LDS); tile shapes are chosen per product so that the grid is ~240-256 tiles.
//
// LDS image: k-contiguous tiles as in gemm_device.hpp (kc_off).  k-strided tiles [64][R] of ANY width R (multiple of
// 16) are stored as power-of-two parts (128 / 64 / 32 / 16 columns, largest first), each [64][w] with a 32-B-group XOR
// that makes both ds_read_b64_tr_b16 of a fragment conflict-free: a half wave reads k-rows {0..3, 8..11} (+4 for the
// second read) of one 16-column group, and f(k) puts those 8 x 32 B on 8 distinct bank groups of a 256-B line.
#include <hip/hip_ext.h>
#include <type_traits>
#include <utility>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "gemm_device.hpp"
#include "../../include/swat_hip.h"
#include "../../include/swat_hip_tuning.h"

namespace swat {

// tuning only (exp bit 16): 8 x uint64 per workgroup {entry, set-up done, barrier 0 passed, main loop done, epilogue
// issued, stores acknowledged, HW_ID | XCC_ID << 32, 0} in 100 MHz ticks, written by consumer wave 0
#define PC_STAMP(i)                                                         \
    do {                                                                    \
        if (tracing) tr[i] = wall_clock64();                                \
    } while (0)

template <int N_>
__device__ __forceinline__ void pc_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
__device__ __forceinline__ void pc_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- inline-asm LDS reads: the compiler's own wait insertion drains ALL outstanding LDS reads (lgkmcnt(0)) in front of
// the first MFMA of a half step, prefetched fragments included; with asm reads it inserts nothing and the waits below
// are counted by hand.  pin() ties a fragment to the wait in front of it (every later use reads the asm's output).
template <int IMM>
__device__ __forceinline__ bf16x8_t pc_lds_b128(unsigned addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
template <int N_, typename F, int... Is>
__device__ __forceinline__ void pc_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N_, typename F>
__device__ __forceinline__ void pc_static_for(F&& f) {
    pc_static_for_impl<N_>(f, std::make_integer_sequence<int, N_>{});
}
__device__ __forceinline__ bf16x8_t pc_lds_tr(unsigned lo_addr, unsigned hi_addr) {
    s16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(lo_addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(hi_addr));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
template <int N_>
__device__ __forceinline__ void pc_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_ > 15 ? 15 : N_) : "memory");
}
__device__ __forceinline__ void pc_pin(bf16x8_t& v) { asm volatile("" : "+v"(v)); }

// ---- epilogue through LDS: the consumers park their fp32 accumulators in a [BM][BN + 4] tile (the operand ring is
// free by then), and ALL waves of the workgroup — loaders included — apply the fused epilogue to 8 (bf16 output) or 4
// (fp32 output) consecutive columns per thread and store whole 16-B chunks of contiguous row segments.  Same arithmetic
// and rounding as gemm_epilogue (gemm_device.hpp).
template <int BM, int BN, int NT_>
__device__ __forceinline__ void pc_store_tile(const GemmArgs& p, const float* cl, int m0, int n0, int tid, int zo, int zi) {
    constexpr int LDT = BN + 4;
    const int M = p.M, N = p.N;
    const int64_t coff = zo * p.sC0 + zi * p.sC1;
    const bool full_epi = !(p.epi0 && zi != 0);
    const bool has_bias = p.bias && full_epi, has_rb = p.rowbias && full_epi, has_res = p.residual && full_epi;
    if (!p.out_f32) {
        constexpr int CPR = BN / 8, NCH = BM * CPR, IT = (NCH + NT_ - 1) / NT_;
#pragma unroll 2
        for (int it = 0; it < IT; ++it) {
            const int c = tid + it * NT_;
            const int row = c / CPR, col = (c % CPR) * 8;
            const int m = m0 + row, n = n0 + col;
            if (c >= NCH || m >= M || n >= N) continue;
            int orow = m, brow = 0;
            if (p.rpg > 0) {
                brow = (m % p.rpg) + p.roff;
                orow = (m / p.rpg) * p.gstride + brow;
            }
            const float4 c0 = *reinterpret_cast<const float4*>(cl + row * LDT + col);
            const float4 c1 = *reinterpret_cast<const float4*>(cl + row * LDT + col + 4);
            float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
            if (has_bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            if (has_rb) {
                const float* rp = p.rowbias + (int64_t)brow * N + n;
                const float4 b0 = *reinterpret_cast<const float4*>(rp), b1 = *reinterpret_cast<const float4*>(rp + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            const int64_t o = coff + (int64_t)orow * p.ldc + n;
            if (has_res) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.residual + o), b1 = *reinterpret_cast<const float4*>(p.residual + o + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            if (p.actin) {  // dgrad through the activation: dF = dG * act'(F)
                const u32x4_t f = *reinterpret_cast<const u32x4_t*>(p.actin + (int64_t)m * p.ldact + n);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    v[2 * h] *= act_bwd(bf2f((unsigned short)(f[h] & 0xffffu)), p.act);
                    v[2 * h + 1] *= act_bwd(bf2f((unsigned short)(f[h] >> 16)), p.act);
                }
            }
            u32x4_t pk;
#pragma unroll
            for (int h = 0; h < 4; ++h) pk[h] = pack_bf16x2(v[2 * h], v[2 * h + 1]);
            if (!p.c2_only) *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned short*>(p.C) + o) = pk;
            if (p.C2) {  // second output: activation of the ROUNDED pre-activation (what backward re-reads)
                u32x4_t pk2;
#pragma unroll
                for (int h = 0; h < 4; ++h)
                    pk2[h] = pack_bf16x2(act_fwd(__uint_as_float(pk[h] << 16), p.act), act_fwd(__uint_as_float(pk[h] & 0xffff0000u), p.act));
                *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned short*>(p.C2) + o) = pk2;
            }
        }
    } else {
        constexpr int CPR = BN / 4, NCH = BM * CPR, IT = (NCH + NT_ - 1) / NT_;
#pragma unroll 4
        for (int it = 0; it < IT; ++it) {
            const int c = tid + it * NT_;
            const int row = c / CPR, col = (c % CPR) * 4;
            const int m = m0 + row, n = n0 + col;
            if (c >= NCH || m >= M || n >= N) continue;
            int orow = m, brow = 0;
            if (p.rpg > 0) {
                brow = (m % p.rpg) + p.roff;
                orow = (m / p.rpg) * p.gstride + brow;
            }
            const float4 c0 = *reinterpret_cast<const float4*>(cl + row * LDT + col);
            float v[4] = {c0.x * p.alpha, c0.y * p.alpha, c0.z * p.alpha, c0.w * p.alpha};
            if (has_bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            }
            if (has_rb) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.rowbias + (int64_t)brow * N + n);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            }
            const int64_t o = coff + (int64_t)orow * p.ldc + n;
            if (has_res) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.residual + o);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            }
            if (p.actin) {
                const ushort4 f4 = *reinterpret_cast<const ushort4*>(p.actin + (int64_t)m * p.ldact + n);
                v[0] *= act_bwd(bf2f(f4.x), p.act); v[1] *= act_bwd(bf2f(f4.y), p.act);
                v[2] *= act_bwd(bf2f(f4.z), p.act); v[3] *= act_bwd(bf2f(f4.w), p.act);
            }
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <int LAYOUT, int BM, int BN, int CM, int CN, int NL, int NS>
__global__ __launch_bounds__(64 * (CM * CN + NL)) void gemm_pc_kernel(const GemmArgs p) {
    constexpr bool A_KC = (LAYOUT != L_TN);
    constexpr bool B_KC = (LAYOUT == L_NT);
    constexpr int NC = CM * CN;
    constexpr int NTHR = 64 * (NC + NL);
    constexpr int WM = BM / CM, WN = BN / CN;
    constexpr int FM = WM / 16, FN = WN / 16;
    static_assert(BM % (16 * CM) == 0 && BN % (16 * CN) == 0, "wave tiles are whole 16 x 16 fragments");
    constexpr int PA = BM / 8, PB = BN / 8, P = PA + PB;  // 1 KiB pieces per stage
    constexpr int A_BYTES = PA * 1024;
    constexpr int STAGE = P * 1024;
    constexpr int LPS = (P + NL - 1) / NL;  // pieces per loader wave and stage (the last may be a harmless repeat)
    static_assert(NS >= 4, "consumers read stages t and t+1 while t+2, t+3 fly");
    static_assert(NS * STAGE <= 160 * 1024, "LDS ring");
    static_assert(BM * (BN + 4) * 4 <= NS * STAGE, "the fp32 output tile reuses the ring");
    static_assert(P > NL, "more pieces than loader waves");
    constexpr int NA = A_KC ? 1 : 2, NB = B_KC ? 1 : 2;
    constexpr int NR = FM * NA + FN * NB;  // LDS read instructions per 32-deep half step

    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];
    const bool tracing = (p.exp & 16) != 0;
    unsigned long long tr[6] = {0, 0, 0, 0, 0, 0}, clkd = 0;
    PC_STAMP(0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    tile_coords(p, bx, by);
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int z = blockIdx.z;
    const int zo = z / p.inner, zi = z % p.inner;
    const int M = p.M, N = p.N;
    const int nk = p.K >> 6;
    // the LDS-staged epilogue needs 16-B rows everywhere it touches memory; anything else takes the per-wave epilogue
    const bool vec_epi = (p.out_f32 ? (N % 4 == 0) : (N % 8 == 0 && p.ldc % 8 == 0 && ((p.sC0 | p.sC1) % 8) == 0)) &&
                         (!p.actin || p.ldact % 8 == 0) && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(p.C2) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.actin) & 15) == 0 &&
                         !(p.exp & 32);

    if (wave >= NC) {
        // ================================ loader waves ================================
        // wave-uniform base pointers of k-step 0 (k-contiguous: first row of the tile; strided: row 0, column 0)
        const char* gA = reinterpret_cast<const char*>(p.A + zo * p.sA0 + zi * p.sA1 + (A_KC ? (int64_t)m0 * p.lda : 0));
        const char* gB = reinterpret_cast<const char*>(p.B + zo * p.sB0 + zi * p.sB1 + (B_KC ? (int64_t)n0 * p.ldb : 0));
        const int64_t stepA = A_KC ? 128 : (int64_t)128 * p.lda;  // bytes per 64-deep k-step
        const int64_t stepB = B_KC ? 128 : (int64_t)128 * p.ldb;
        const int lw = wave - NC;
        unsigned voff[LPS];
        unsigned amask = 0, dmask = 0;  // bit i: piece i of this wave belongs to A / repeats piece i - 1
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            int q = lw + NL * i;
            if (q >= P) { q -= NL; dmask |= 1u << i; }  // P % NL != 0: repeat the previous piece (same bytes, same place)
            const bool isA = q < PA;
            if (isA) amask |= 1u << i;
            const int byte = (isA ? q : q - PA) * 1024 + lane * 16;
            unsigned off;
            if (isA) {
                if constexpr (A_KC) {
                    const int row = byte >> 7, t = ((byte >> 4) & 7) ^ ((row >> 1) & 7);
                    int gr = m0 + row; gr = gr < M ? gr : M - 1;
                    off = (unsigned)((gr - m0) * p.lda + 8 * t) * 2u;
                } else {
                    int k, col;
                    stp_inv<BM>(byte, k, col);
                    int gc = m0 + col; gc = gc + 8 <= p.lda ? gc : 0;
                    off = (unsigned)(k * p.lda + gc) * 2u;
                }
            } else {
                if constexpr (B_KC) {
                    const int row = byte >> 7, t = ((byte >> 4) & 7) ^ ((row >> 1) & 7);
                    int gr = n0 + row; gr = gr < N ? gr : N - 1;
                    off = (unsigned)((gr - n0) * p.ldb + 8 * t) * 2u;
                } else {
                    int k, col;
                    stp_inv<BN>(byte, k, col);
                    int gc = n0 + col; gc = gc + 8 <= p.ldb ? gc : 0;
                    off = (unsigned)(k * p.ldb + gc) * 2u;
                }
            }
            voff[i] = off;
        }
        // k-order skew: tiles that share an operand panel start at different k-steps (and wrap), so at any instant they
        // pull different 128-B columns of the panel.  The per-tile summation order changes, deterministically.
        int kk = (p.exp & 8) ? 0 : (bx * 5 + by * 3) % nk;
        const char* curA = gA + kk * stepA;
        const char* curB = gB + kk * stepB;
        auto issue = [&](int buf) {
            unsigned char* st = smem + buf * STAGE + lw * 1024;
#pragma unroll
            for (int i = 0; i < LPS; ++i) {
                const char* src = ((amask >> i) & 1) ? curA : curB;
                unsigned char* dst = st + (NL * i - (((dmask >> i) & 1) ? NL : 0)) * 1024;
                if (p.exp & 64) {  // tuning only: the same traffic into registers (no LDS writes)
                    u32x4_t v;
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff[i]), "s"(src));
                    continue;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + voff[i]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
            curA += stepA; curB += stepB;
            if (++kk == nk) {
                kk = 0;
                curA -= nk * stepA; curB -= nk * stepB;
            }
        };
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nk) issue(s);
        // barrier t: stage min(t + 1, nk - 1) has landed; then stage t + NS - 1 goes into the buffer stage t - 1 left
        int t = 0, buf = NS - 1;
        for (; t + NS - 1 < nk; ++t) {  // steady state: stages up to t + NS - 2 issued, NS - 3 of them may still fly
            pc_wait_vmcnt<(NS - 3) * LPS>();
            pc_barrier();
            if (!(p.exp & 1)) issue(buf);
            buf = buf + 1 == NS ? 0 : buf + 1;
        }
        for (; t < nk; ++t) {  // tail: everything is issued; stages t + 2 .. nk - 1 may still fly
            switch (nk - 2 - t) {
                case 1: pc_wait_vmcnt<LPS>(); break;
                case 2: pc_wait_vmcnt<(NS >= 5 ? 2 : 0) * LPS>(); break;
                case 3: pc_wait_vmcnt<(NS >= 6 ? 3 : 0) * LPS>(); break;
                default: pc_wait_vmcnt<0>(); break;
            }
            pc_barrier();
        }
        if (!vec_epi) return;
    }

    // ================================ consumer waves ================================
    const int wm0 = (wave / CN) * WM;
    const int wn0 = (wave % CN) * WN;
    const int r16 = lane & 15, g = lane >> 4;
    f32x4_t acc[FM][FN];
    f32x4_t accs[FM];
    const bool do_cs = LAYOUT == L_TN && p.colsum && n0 == 0 && wn0 == 0 && wave < NC;
    if (wave < NC) {
        // loop-invariant per-lane fragment read addresses inside stage 0 (k-contiguous: one per half step, the 16-row
        // fragments follow at immediate offsets 2048 i; strided: every read has its own, the XOR is per k-row and part)
        const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
        unsigned ra[2][A_KC ? 1 : FM][NA], rb[2][B_KC ? 1 : FN][NB];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (A_KC) ra[ks][0][0] = lds0 + kc_off(wm0 + r16, 4 * ks + g);
            else {
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int k = 32 * ks + 8 * g + (r16 >> 2);
                    ra[ks][i][0] = lds0 + stp_off<BM>(k, wm0 + 16 * i) + 8 * (r16 & 3);
                    ra[ks][i][NA - 1] = lds0 + stp_off<BM>(k + 4, wm0 + 16 * i) + 8 * (r16 & 3);
                }
            }
            if constexpr (B_KC) rb[ks][0][0] = lds0 + A_BYTES + kc_off(wn0 + r16, 4 * ks + g);
            else {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int k = 32 * ks + 8 * g + (r16 >> 2);
                    rb[ks][j][0] = lds0 + A_BYTES + stp_off<BN>(k, wn0 + 16 * j) + 8 * (r16 & 3);
                    rb[ks][j][NB - 1] = lds0 + A_BYTES + stp_off<BN>(k + 4, wn0 + 16 * j) + 8 * (r16 & 3);
                }
            }
        }
        // Fragment registers: ONE set of A fragments, refreshed in place — right after the MFMAs of accumulator row i
        // have been issued, the read of row i's fragment for the NEXT half step goes out, so every A fragment has a
        // whole half step (FM x FN MFMAs) to arrive; TWO sets of B fragments (all rows use them), the next half's
        // going out behind the first rows.  Reads and MFMAs alternate row by row: a burst of reads in front of a half
        // step would hold the in-order wave's MFMAs back for as long as the LDS needs for all the waves' bursts.
        bf16x8_t fa[FM], fb[2][FN];
        constexpr int RB = FM >= 2 ? FM / 2 : 1;  // the next half's B fragments go out behind rows 0 .. RB-1
        auto read_a = [&](auto I, unsigned so, int ks) {
            constexpr int i = decltype(I)::value;
            if constexpr (A_KC) fa[i] = pc_lds_b128<2048 * i>(ra[ks][0][0] + so);
            else fa[i] = pc_lds_tr(ra[ks][i][0] + so, ra[ks][i][NA - 1] + so);
        };
        auto read_b = [&](auto J, unsigned so, int ks, bf16x8_t (&bfr)[FN]) {
            constexpr int j = decltype(J)::value;
            if constexpr (B_KC) bfr[j] = pc_lds_b128<2048 * j>(rb[ks][0][0] + so);
            else bfr[j] = pc_lds_tr(rb[ks][j][0] + so, rb[ks][j][NB - 1] + so);
        };
        // bias gradient beside a weight gradient: column sums of A = A^T 1, one MFMA against a ones fragment per half step
        const s16x8_t ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
        const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_bits);
#pragma unroll
        for (int i = 0; i < FM; ++i) accs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        constexpr int NRI = FM * NA + FN * NB;  // LDS read instructions per half step
        // one 32-deep half step on (fa, bcur); WITH_READS: the fragments of the next half step (stage offset sn, half
        // nks) replace fa in place and fill bnext.  Counted waits: in front of row i everything issued after a_i's own
        // read may still fly (NRI - NA instructions); in front of row 0 all of bcur must be there as well, i.e. only
        // the A reads of rows RB .. FM-1 of the previous half may fly.
        auto half = [&](auto WITH_READS, bf16x8_t (&bcur)[FN], bf16x8_t (&bnext)[FN], unsigned sn, int nks) {
            pc_static_for<FM>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (i == 0) {
                    pc_wait_lgkm<(FM - RB) * NA>();
#pragma unroll
                    for (int j = 0; j < FN; ++j) pc_pin(bcur[j]);
                } else if constexpr (decltype(WITH_READS)::value) {
                    pc_wait_lgkm<NRI - NA>();
                } else {  // last half step: nothing new was issued behind rows 0 .. i-1, only rows i+1 .. of the previous half
                    constexpr int later_b = i + 1 < RB ? FN - (i + 1) * FN / RB : 0;
                    pc_wait_lgkm<(FM - 1 - i) * NA + later_b * NB>();
                }
                pc_pin(fa[i]);
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bcur[j], fa[i], acc[i][j], 0, 0, 0);
                if constexpr (LAYOUT == L_TN) {
                    if (do_cs) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], accs[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (decltype(WITH_READS)::value) {
                    read_a(I, sn, nks);
                    if constexpr (i < RB) {
                        constexpr int j0 = i * FN / RB, j1 = (i + 1) * FN / RB;
                        pc_static_for<j1 - j0>([&](auto Jd) {
                            read_b(std::integral_constant<int, j0 + decltype(Jd)::value>{}, sn, nks, bnext);
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        };
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;

        PC_STAMP(1);
        pc_barrier();  // barrier 0: stages 0 and 1 have landed
        PC_STAMP(2);
        unsigned long long clk0 = 0;
        if (tracing) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0));
        if (!(p.exp & 2)) {
            pc_static_for<FM>([&](auto I) { read_a(I, 0u, 0); });
            pc_static_for<FN>([&](auto J) { read_b(J, 0u, 0, fb[0]); });
            pc_wait_lgkm<0>();
            unsigned so = 0;  // byte offset of stage t in the ring
            for (int t = 0; t + 1 < nk; ++t) {
                const unsigned sn = so + STAGE == NS * STAGE ? 0u : so + STAGE;
                half(T1{}, fb[0], fb[1], so, 1);  // half 0 of k-step t, reads of its half 1
                half(T1{}, fb[1], fb[0], sn, 0);  // half 1, reads of half 0 of k-step t + 1 (visible since barrier t)
                pc_barrier();                     // barrier t + 1: stage t is free, stage t + 2 has landed
                so = sn;
            }
            half(T1{}, fb[0], fb[1], so, 1);
            half(T0{}, fb[1], fb[0], 0u, 0);
        } else {  // tuning only: DMA streaming without LDS reads / MFMA
            for (int t = 0; t + 1 < nk; ++t) pc_barrier();
        }
        PC_STAMP(3);
        if (tracing) {
            unsigned long long clk1;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1));
            clkd = clk1 - clk0;
        }
        if (!vec_epi) {
            if (!(p.exp & 4)) gemm_epilogue<FM, FN>(p, acc, m0 + wm0, n0 + wn0, lane, zo, zi);
        }
        if (do_cs && (lane >> 4) == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm0 + 16 * i + (lane & 15);
                if (m < p.M) p.colsum[m] = accs[i][0];
            }
        }
    }
    if (vec_epi) {
        float* cl = reinterpret_cast<float*>(smem);
        pc_barrier();  // every consumer has finished reading the ring (and every LDS-DMA has landed)
        if (wave < NC && !(p.exp & 4)) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    *reinterpret_cast<f32x4_t*>(cl + (wm0 + 16 * i + r16) * (BN + 4) + wn0 + 16 * j + 4 * g) = acc[i][j];
        }
        __syncthreads();
        if (!(p.exp & 4)) pc_store_tile<BM, BN, NTHR>(p, cl, m0, n0, tid, zo, zi);
    }
    if (tracing && wave == 0) {
        PC_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PC_STAMP(5);
        if (tid == 0 && p.trace) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o = p.trace + 8 * (size_t)blockIdx.x;
            for (int i = 0; i < 6; ++i) o[i] = tr[i];
            o[6] = ((unsigned long long)xcc << 32) | hw;
            o[7] = clkd;
        }
    }
}

// ---- persistent, grouped form: up to 4 products of one layout in ONE launch, 256 workgroups walking the joint tile list
// (the weight gradients of a transformer block: four launches of 64-256 tiles each pay four times for dispatch, set-up
// and the first operand bytes).  The stage stream is continuous across tiles: the loaders run NS-1 stages ahead straight
// into the next tile's operands while the consumers finish a tile, park its accumulators in an LDS tile of its own
// (behind the ring) and store it with the fused epilogue; one extra barrier per tile.  Tiles of a workgroup are
// b, b + G, b + 2G, ... of the joint list after the XCD-contiguous remap, so an XCD's L2 keeps serving the same panels.
constexpr int PCG_MAX = 64;     // products per launch (a 12-block chunk of a tower = 48)
constexpr int PCG_MAX_WG = 512;  // persistent workgroups
struct GemmGroup {               // lives in device memory (built once per distinct product list, launch_gemm_pc_group)
    int n, total;
    int tile_end[PCG_MAX];       // running tile count behind product i
    int wg_stages[PCG_MAX_WG];   // all k-steps of all tiles of workgroup b (the loaders' stream length)
    GemmArgs p[PCG_MAX];
};

// joint tile list -> (problem, tile row, tile column), after the XCD-contiguous remap of the virtual tile number
__host__ __device__ inline void pcg_decode(const int* tile_end, int n, int total, int vt, int& pi, int& id_in) {
    const int xcd = vt & 7, q = total >> 3, r = total & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vt >> 3);
    int lo = 0, hi = n - 1;  // first product whose tile_end exceeds id
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (id >= tile_end[mid]) lo = mid + 1; else hi = mid;
    }
    pi = lo;
    id_in = id - (lo ? tile_end[lo - 1] : 0);
}

template <int LAYOUT, int BM, int BN, int CM, int CN, int NL, int NS>
__global__ __launch_bounds__(64 * (CM * CN + NL)) void gemm_pcg_kernel(const GemmGroup* __restrict__ gp) {
    const GemmGroup& g = *gp;
    constexpr bool A_KC = (LAYOUT != L_TN);
    constexpr bool B_KC = (LAYOUT == L_NT);
    constexpr int NC = CM * CN;
    constexpr int WM = BM / CM, WN = BN / CN;
    constexpr int FM = WM / 16, FN = WN / 16;
    static_assert(BM % (16 * CM) == 0 && BN % (16 * CN) == 0, "wave tiles are whole 16 x 16 fragments");
    constexpr int PA = BM / 8, PB = BN / 8, P = PA + PB;
    constexpr int A_BYTES = PA * 1024;
    constexpr int STAGE = P * 1024;
    constexpr int LPS = (P + NL - 1) / NL;
    constexpr int CT_BYTES = BM * (BN + 4) * 4;
    static_assert(NS >= 4, "consumers read stages t and t+1 while t+2, t+3 fly");
    static_assert(NS * STAGE + CT_BYTES <= 160 * 1024, "LDS ring + output tile");
    static_assert(P > NL, "more pieces than loader waves");
    constexpr int NA = A_KC ? 1 : 2, NB = B_KC ? 1 : 2;

    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE + CT_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, total = g.total;
    const int ntiles = (total - (int)blockIdx.x + G - 1) / G;  // tiles of this workgroup (>= 1: G <= total)

    auto decode = [&](int vt, int& pi, int& by, int& bx) {
        int id;
        pcg_decode(g.tile_end, g.n, total, vt, pi, id);
        const int tm = g.p[pi].tiles_m, tn = g.p[pi].tiles_n;
        if (g.p[pi].swz == 2) { bx = id % tn; by = id / tn; }
        else { by = id % tm; bx = id / tm; }
    };

    if (wave >= NC) {
        // ================================ loader waves ================================
        const int lw = wave - NC;
        unsigned voff[LPS];
        unsigned amask = 0, dmask = 0;
        const char *curA = nullptr, *curB = nullptr;
        int64_t stepA = 0, stepB = 0;
        int nk = 0, kk = 0;  // k-steps of the tile being issued, next k-step to issue
        auto open_tile = [&](int ti) {  // per-lane source offsets and wave-uniform bases of tile number ti of this workgroup
            int pi, by, bx;
            decode((int)blockIdx.x + ti * G, pi, by, bx);
            const GemmArgs& p = g.p[pi];
            const int m0 = by * BM, n0 = bx * BN, M = p.M, N = p.N;
            amask = 0; dmask = 0;
#pragma unroll
            for (int i = 0; i < LPS; ++i) {
                int q = lw + NL * i;
                if (q >= P) { q -= NL; dmask |= 1u << i; }
                const bool isA = q < PA;
                if (isA) amask |= 1u << i;
                const int byte = (isA ? q : q - PA) * 1024 + lane * 16;
                unsigned off;
                if (isA) {
                    if constexpr (A_KC) {
                        const int row = byte >> 7, t = ((byte >> 4) & 7) ^ ((row >> 1) & 7);
                        int gr = m0 + row; gr = gr < M ? gr : M - 1;
                        off = (unsigned)((gr - m0) * p.lda + 8 * t) * 2u;
                    } else {
                        int k, col;
                        stp_inv<BM>(byte, k, col);
                        int gc = m0 + col; gc = gc + 8 <= p.lda ? gc : 0;
                        off = (unsigned)(k * p.lda + gc) * 2u;
                    }
                } else {
                    if constexpr (B_KC) {
                        const int row = byte >> 7, t = ((byte >> 4) & 7) ^ ((row >> 1) & 7);
                        int gr = n0 + row; gr = gr < N ? gr : N - 1;
                        off = (unsigned)((gr - n0) * p.ldb + 8 * t) * 2u;
                    } else {
                        int k, col;
                        stp_inv<BN>(byte, k, col);
                        int gc = n0 + col; gc = gc + 8 <= p.ldb ? gc : 0;
                        off = (unsigned)(k * p.ldb + gc) * 2u;
                    }
                }
                voff[i] = off;
            }
            curA = reinterpret_cast<const char*>(p.A + (A_KC ? (int64_t)m0 * p.lda : 0));
            curB = reinterpret_cast<const char*>(p.B + (B_KC ? (int64_t)n0 * p.ldb : 0));
            stepA = A_KC ? 128 : (int64_t)128 * p.lda;
            stepB = B_KC ? 128 : (int64_t)128 * p.ldb;
            nk = p.K >> 6;
            kk = 0;
        };
        int ti_issue = 0;  // tile the next issue belongs to
        open_tile(0);
        const int S = g.wg_stages[blockIdx.x];  // all k-steps of all tiles of this workgroup (host-computed)
        int issued = 0, buf = 0;
        auto issue = [&]() {  // next stage of the stream into ring buffer `buf`
            unsigned char* st = smem + buf * STAGE + lw * 1024;
#pragma unroll
            for (int i = 0; i < LPS; ++i) {
                const char* src = ((amask >> i) & 1) ? curA : curB;
                unsigned char* dst = st + (NL * i - (((dmask >> i) & 1) ? NL : 0)) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + voff[i]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
            curA += stepA; curB += stepB;
            ++issued;
            buf = buf + 1 == NS ? 0 : buf + 1;
            if (++kk == nk && ti_issue + 1 < ntiles) open_tile(++ti_issue);
        };
#pragma unroll 1
        for (int s0 = 0; s0 < NS - 1 && s0 < S; ++s0) issue();
        // consumer side bookkeeping mirrored here: the extra barrier at the end of every tile
        int ct = 0, ck = 0, cnk;  // tile / k-step the consumers are at
        {
            int pi, by, bx;
            decode((int)blockIdx.x, pi, by, bx);
            cnk = g.p[pi].K >> 6;
        }
        for (int s = 0; s < S; ++s) {
            // stages issued: `issued`; stage min(s + 1, S - 1) must have landed
            const int need = s + 1 < S - 1 ? s + 1 : S - 1;
            switch (issued - 1 - need) {
                case 1: pc_wait_vmcnt<LPS>(); break;
                case 2: pc_wait_vmcnt<(NS >= 5 ? 2 : 0) * LPS>(); break;
                case 3: pc_wait_vmcnt<(NS >= 6 ? 3 : 0) * LPS>(); break;
                default: pc_wait_vmcnt<0>(); break;
            }
            pc_barrier();
            if (issued < S) issue();
            if (++ck == cnk) {  // the consumers finish a tile behind this stage: their accumulators go through LDS
                pc_barrier();
                ck = 0;
                if (++ct < ntiles) {
                    int pi, by, bx;
                    decode((int)blockIdx.x + ct * G, pi, by, bx);
                    cnk = g.p[pi].K >> 6;
                }
            }
        }
        return;
    }

    // ================================ consumer waves ================================
    const int wm0 = (wave / CN) * WM;
    const int wn0 = (wave % CN) * WN;
    const int r16 = lane & 15, gq = lane >> 4;
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
    float* cl = reinterpret_cast<float*>(smem + NS * STAGE);
    unsigned ra[2][A_KC ? 1 : FM][NA], rb[2][B_KC ? 1 : FN][NB];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if constexpr (A_KC) ra[ks][0][0] = lds0 + kc_off(wm0 + r16, 4 * ks + gq);
        else {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int k = 32 * ks + 8 * gq + (r16 >> 2);
                ra[ks][i][0] = lds0 + stp_off<BM>(k, wm0 + 16 * i) + 8 * (r16 & 3);
                ra[ks][i][NA - 1] = lds0 + stp_off<BM>(k + 4, wm0 + 16 * i) + 8 * (r16 & 3);
            }
        }
        if constexpr (B_KC) rb[ks][0][0] = lds0 + A_BYTES + kc_off(wn0 + r16, 4 * ks + gq);
        else {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int k = 32 * ks + 8 * gq + (r16 >> 2);
                rb[ks][j][0] = lds0 + A_BYTES + stp_off<BN>(k, wn0 + 16 * j) + 8 * (r16 & 3);
                rb[ks][j][NB - 1] = lds0 + A_BYTES + stp_off<BN>(k + 4, wn0 + 16 * j) + 8 * (r16 & 3);
            }
        }
    }
    bf16x8_t fa[FM], fb[2][FN];
    constexpr int RB = FM >= 2 ? FM / 2 : 1;
    auto read_a = [&](auto I, unsigned so, int ks) {
        constexpr int i = decltype(I)::value;
        if constexpr (A_KC) fa[i] = pc_lds_b128<2048 * i>(ra[ks][0][0] + so);
        else fa[i] = pc_lds_tr(ra[ks][i][0] + so, ra[ks][i][NA - 1] + so);
    };
    auto read_b = [&](auto J, unsigned so, int ks, bf16x8_t (&bfr)[FN]) {
        constexpr int j = decltype(J)::value;
        if constexpr (B_KC) bfr[j] = pc_lds_b128<2048 * j>(rb[ks][0][0] + so);
        else bfr[j] = pc_lds_tr(rb[ks][j][0] + so, rb[ks][j][NB - 1] + so);
    };
    const s16x8_t ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_bits);
    f32x4_t acc[FM][FN], accs[FM];
    bool do_cs = false;
    constexpr int NRI = FM * NA + FN * NB;
    auto half = [&](auto WITH_READS, bf16x8_t (&bcur)[FN], bf16x8_t (&bnext)[FN], unsigned sn, int nks) {
        pc_static_for<FM>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (i == 0) {
                pc_wait_lgkm<(FM - RB) * NA>();
#pragma unroll
                for (int j = 0; j < FN; ++j) pc_pin(bcur[j]);
            } else if constexpr (decltype(WITH_READS)::value) {
                pc_wait_lgkm<NRI - NA>();
            } else {
                constexpr int later_b = i + 1 < RB ? FN - (i + 1) * FN / RB : 0;
                pc_wait_lgkm<(FM - 1 - i) * NA + later_b * NB>();
            }
            pc_pin(fa[i]);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bcur[j], fa[i], acc[i][j], 0, 0, 0);
            if constexpr (LAYOUT == L_TN) {
                if (do_cs) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], accs[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (decltype(WITH_READS)::value) {
                read_a(I, sn, nks);
                if constexpr (i < RB) {
                    constexpr int j0 = i * FN / RB, j1 = (i + 1) * FN / RB;
                    pc_static_for<j1 - j0>([&](auto Jd) {
                        read_b(std::integral_constant<int, j0 + decltype(Jd)::value>{}, sn, nks, bnext);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;

    pc_barrier();  // barrier 0: stages 0 and 1 have landed
    pc_static_for<FM>([&](auto I) { read_a(I, 0u, 0); });
    pc_static_for<FN>([&](auto J) { read_b(J, 0u, 0, fb[0]); });
    pc_wait_lgkm<0>();
    unsigned so = 0;  // byte offset in the ring of the stage being computed
    for (int ti = 0; ti < ntiles; ++ti) {
        int pi, by, bx;
        decode((int)blockIdx.x + ti * G, pi, by, bx);
        const GemmArgs& p = g.p[pi];
        const int m0 = by * BM, n0 = bx * BN;
        const int nk = p.K >> 6;
        do_cs = LAYOUT == L_TN && p.colsum && n0 == 0 && wn0 == 0;
#pragma unroll
        for (int i = 0; i < FM; ++i) accs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t + 1 < nk; ++t) {
            const unsigned sn = so + STAGE == NS * STAGE ? 0u : so + STAGE;
            half(T1{}, fb[0], fb[1], so, 1);
            half(T1{}, fb[1], fb[0], sn, 0);
            pc_barrier();
            so = sn;
        }
        const unsigned sn = so + STAGE == NS * STAGE ? 0u : so + STAGE;
        half(T1{}, fb[0], fb[1], so, 1);
        if (ti + 1 < ntiles) half(T1{}, fb[1], fb[0], sn, 0);  // the next tile's first fragments (its stage 0 is visible)
        else half(T0{}, fb[1], fb[0], 0u, 0);
        // epilogue: accumulators -> LDS tile -> fused epilogue + 16-B stores by the consumer waves
        if (do_cs && (lane >> 4) == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm0 + 16 * i + (lane & 15);
                if (m < p.M) p.colsum[m] = accs[i][0];
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                *reinterpret_cast<f32x4_t*>(cl + (wm0 + 16 * i + r16) * (BN + 4) + wn0 + 16 * j + 4 * gq) = acc[i][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pc_barrier();  // the tile's extra barrier: every consumer has parked its accumulators
        pc_store_tile<BM, BN, 64 * NC>(p, cl, m0, n0, tid, 0, 0);
        if (ti + 1 < ntiles) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LDS tile has been read before the next tile's stage barrier
            pc_barrier();  // barrier of the next tile's first stage
            so = sn;
        }
    }
}

// Descriptor cache: the product lists of a training job repeat every step (same buffers), so a list is uploaded once and
// kept for the life of the process (a ViT-L/14 step has ~200 distinct lists of one product: 3 KB of device memory each).
struct PcgSig {  // EVERY GemmArgs field the kernel reads (GemmArgs has padding bytes: never compared raw); no padding inside
    const void *A, *B, *C, *C2, *bias, *residual, *actin, *rowbias, *colsum, *trace;
    int64_t sC0, sC1;
    int M, N, K, lda, ldb, ldc, ldact, act, out_f32, rpg, gstride, roff, epi0, exp;
    float alpha;
    int c2_only;
    bool operator==(const PcgSig& o) const { return memcmp(this, &o, sizeof(PcgSig)) == 0; }
};
struct PcgEntry {
    std::vector<PcgSig> sig;
    GemmGroup* dev;
    int G, cfg, layout;
    uint64_t stamp;  // last use (least recently used entry is evicted beyond PCG_CACHE_MAX)
};
constexpr size_t PCG_CACHE_MAX = 512;  // distinct product lists kept (a ViT-L/14 step has ~200 of one product each)
static std::unordered_map<uint64_t, std::vector<PcgEntry>> g_pcg_cache;
static size_t g_pcg_entries = 0;
static uint64_t g_pcg_clock = 0;
static std::mutex g_pcg_mutex;  // the cache is process-wide: two host threads may launch grouped products concurrently

struct PcgTile { int bm, bn, maxg; };
static PcgTile pcg_tile(int layout, int cfg) {
    // cfg 0: the layout's default — TN 96 x 96 (SWAT_WGRAD_TILE=64: 64 x 64, two workgroups per CU, measured slower), NT / NN
    // 128 x 64; 70 = 128 x 64, 71 = 96 x 96, 72 = 64 x 64 (two per CU)
    static const int wt = getenv("SWAT_WGRAD_TILE") ? atoi(getenv("SWAT_WGRAD_TILE")) : 96;  // tuning
    if (cfg == 0) cfg = layout == L_TN ? (wt == 64 ? 72 : 71) : 70;
    if (cfg == 70) return PcgTile{128, 64, 256};
    if (cfg == 72) return PcgTile{64, 64, 512};
    return PcgTile{96, 96, 256};
}

template <int LAYOUT>
static void pcg_launch(const PcgTile& t, int G, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const GemmGroup* d) {
#define SWAT_PCG(BM_, BN_)                                                                                             \
    {                                                                                                                  \
        auto kern = gemm_pcg_kernel<LAYOUT, BM_, BN_, 2, 2, 4, 4>;                                                     \
        if (ev0) hipExtLaunchKernelGGL(kern, dim3(G), dim3(512), 0, s, ev0, ev1, 0, d);                                \
        else hipLaunchKernelGGL(kern, dim3(G), dim3(512), 0, s, d);                                                    \
    }
    if (t.bm == 128) SWAT_PCG(128, 64) else if (t.bm == 96) SWAT_PCG(96, 96) else SWAT_PCG(64, 64)
#undef SWAT_PCG
}

int launch_gemm_pc_group(int layout, const GemmArgs* probs, int n, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, int cfg) {
    SWAT_REQUIRE(n >= 1 && n <= PCG_MAX, "gemm group: 1-%d products", PCG_MAX);
    const PcgTile t = pcg_tile(layout, cfg);
    std::vector<PcgSig> sig(n);
    uint64_t key = 1469598103934665603ull ^ (uint64_t)(t.bm * 1024 + t.bn) ^ ((uint64_t)layout << 40);
    for (int i = 0; i < n; ++i) {
        const GemmArgs& a = probs[i];
        memset(&sig[i], 0, sizeof(PcgSig));
        sig[i].A = a.A; sig[i].B = a.B; sig[i].C = a.C; sig[i].C2 = a.C2; sig[i].bias = a.bias; sig[i].residual = a.residual;
        sig[i].actin = a.actin; sig[i].rowbias = a.rowbias; sig[i].colsum = a.colsum; sig[i].trace = a.trace;
        sig[i].sC0 = a.sC0; sig[i].sC1 = a.sC1;
        sig[i].M = a.M; sig[i].N = a.N; sig[i].K = a.K; sig[i].lda = a.lda; sig[i].ldb = a.ldb; sig[i].ldc = a.ldc;
        sig[i].ldact = a.ldact; sig[i].act = a.act; sig[i].out_f32 = a.out_f32; sig[i].rpg = a.rpg; sig[i].gstride = a.gstride;
        sig[i].roff = a.roff; sig[i].epi0 = a.epi0; sig[i].exp = a.exp; sig[i].alpha = a.alpha; sig[i].c2_only = a.c2_only;
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&sig[i]);
        for (size_t j = 0; j < sizeof(PcgSig); ++j) key = (key ^ b[j]) * 1099511628211ull;
    }
    std::lock_guard<std::mutex> lock(g_pcg_mutex);
    PcgEntry* e = nullptr;
    auto& bucket = g_pcg_cache[key];
    for (auto& c : bucket)
        if (c.cfg == t.bm * 1024 + t.bn && c.layout == layout && c.sig == sig) e = &c;
    if (!e) {
        const size_t bytes = offsetof(GemmGroup, p) + (size_t)n * sizeof(GemmArgs);
        GemmGroup* d = nullptr;
        if (hipMalloc((void**)&d, bytes) != hipSuccess) {
            set_error("gemm group: hipMalloc of the descriptor failed");
            return -1;
        }
        GemmGroup* up = (GemmGroup*)calloc(1, sizeof(GemmGroup));  // planned copies: tile grid, walk order, per-workgroup stream length
        SWAT_REQUIRE(up, "gemm group: out of host memory");
        int total = 0;
        for (int i = 0; i < n; ++i) {
            up->p[i] = probs[i];
            (void)plan_grid(up->p[i], t.bm, t.bn, 1);
            if (up->p[i].swz == 0) up->p[i].swz = 1;  // small grids: any walk
            total += up->p[i].tiles_m * up->p[i].tiles_n;
            up->tile_end[i] = total;
        }
        up->n = n;
        up->total = total;
        const int G = total < t.maxg ? total : t.maxg;
        for (int b = 0; b < G; ++b) {
            int st = 0;
            for (int vt = b; vt < total; vt += G) {
                int pi, id;
                pcg_decode(up->tile_end, n, total, vt, pi, id);
                st += up->p[pi].K >> 6;
            }
            up->wg_stages[b] = st;
        }
        // synchronous upload (once per distinct list): the staging copy can be freed right away
        const hipError_t ce = hipMemcpy(d, up, bytes, hipMemcpyHostToDevice);
        free(up);
        if (ce != hipSuccess) {
            (void)hipFree(d);
            set_error("gemm group: descriptor upload failed: %s", hipGetErrorString(ce));
            return -1;
        }
        if (g_pcg_entries >= PCG_CACHE_MAX) {  // evict the least recently used list (its descriptor is no longer referenced by any
            uint64_t oldest = ~0ull;           // launch the host has not already enqueued: hipFree synchronises before it frees)
            std::vector<PcgEntry>* ob = nullptr;
            size_t oi = 0;
            for (auto& kv : g_pcg_cache)
                for (size_t j = 0; j < kv.second.size(); ++j)
                    if (kv.second[j].stamp < oldest) { oldest = kv.second[j].stamp; ob = &kv.second; oi = j; }
            if (ob) {
                (void)hipFree((*ob)[oi].dev);
                ob->erase(ob->begin() + (long)oi);
                --g_pcg_entries;
            }
        }
        auto& bucket2 = g_pcg_cache[key];  // (eviction may have touched the map's buckets)
        bucket2.push_back(PcgEntry{sig, d, G, t.bm * 1024 + t.bn, layout, 0});
        e = &bucket2.back();
        ++g_pcg_entries;
    }
    e->stamp = ++g_pcg_clock;
    if (layout == L_NT) pcg_launch<L_NT>(t, e->G, s, ev0, ev1, e->dev);
    else if (layout == L_NN) pcg_launch<L_NN>(t, e->G, s, ev0, ev1, e->dev);
    else pcg_launch<L_TN>(t, e->G, s, ev0, ev1, e->dev);
    return check_launch("swat_gemm group (pc)");
}

// ---- launcher: configurations 50.. (swat_gemm_set_tile pins one; pick_cfg in gemm.hip chooses) ----
template <int LAYOUT>
static int launch_pc_layout(const GemmArgs& a, int batch, int cfg, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
#define SWAT_PC(LMASK_, BM_, BN_, CM_, CN_, NL_, NS_)                                                                \
    if constexpr (((LMASK_) >> LAYOUT) & 1) {                                                                         \
        GemmArgs b = a;                                                                                               \
        dim3 grid = plan_grid(b, BM_, BN_, batch);                                                                    \
        auto kern = gemm_pc_kernel<LAYOUT, BM_, BN_, CM_, CN_, NL_, NS_>;                                             \
        if (ev0) hipExtLaunchKernelGGL(kern, grid, dim3(64 * (CM_ * CN_ + NL_)), 0, s, ev0, ev1, 0, b);              \
        else hipLaunchKernelGGL(kern, grid, dim3(64 * (CM_ * CN_ + NL_)), 0, s, b);                                   \
    } else {                                                                                                          \
        set_error("gemm_pc: tile configuration %d is not built for layout %d", cfg, LAYOUT);                          \
        return -1;                                                                                                    \
    }
    // layout masks: 1 = NT, 2 = NN, 4 = TN
    switch (cfg) {
        case 50: SWAT_PC(7, 128, 128, 2, 2, 4, 4) break;
        case 51: SWAT_PC(7, 160, 128, 2, 2, 4, 4) break;
        case 53: SWAT_PC(7, 96, 64, 2, 2, 4, 4) break;
        case 54: SWAT_PC(4, 96, 96, 2, 2, 4, 4) break;
        case 55: SWAT_PC(7, 128, 64, 2, 2, 4, 4) break;
        case 56: SWAT_PC(7, 64, 64, 2, 2, 4, 4) break;
        case 58: SWAT_PC(3, 128, 128, 2, 4, 4, 4) break;
        case 59: SWAT_PC(3, 192, 128, 3, 2, 4, 4) break;   // 6 consumer waves of 64 x 64, the whole 160 KB of LDS
        case 66: SWAT_PC(3, 192, 128, 3, 4, 4, 4) break;   // 12 consumer waves of 64 x 32
        case 60: SWAT_PC(3, 96, 64, 2, 2, 4, 7) break;
        case 61: SWAT_PC(4, 96, 96, 2, 2, 4, 6) break;
        case 62: SWAT_PC(7, 64, 64, 2, 2, 4, 8) break;
        case 63: SWAT_PC(1, 128, 128, 2, 2, 4, 5) break;
        case 64: SWAT_PC(3, 96, 64, 2, 2, 2, 7) break;
        case 65: SWAT_PC(3, 96, 64, 2, 2, 8, 7) break;
        default: set_error("gemm_pc: unknown tile configuration %d", cfg); return -1;
    }
#undef SWAT_PC
    return check_launch("swat_gemm_bf16 (pc)");
}

int launch_gemm_pc(int layout, const GemmArgs& a, int batch, int cfg, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (layout == L_NT) return launch_pc_layout<L_NT>(a, batch, cfg, s, ev0, ev1);
    if (layout == L_NN) return launch_pc_layout<L_NN>(a, batch, cfg, s, ev0, ev1);
    return launch_pc_layout<L_TN>(a, batch, cfg, s, ev0, ev1);
}

}  // namespace swat
