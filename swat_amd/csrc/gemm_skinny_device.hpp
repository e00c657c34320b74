// swat_amd/csrc/gemm_skinny_device.hpp — device part of the remainder-row product (csrc/gemm_skinny.hip), shared with the tail phase of
// the 256 x 128 kernel (csrc/gemm_xl.hip).  See gemm_skinny.hip for the design.
#pragma once
#include "gemm_device.hpp"

namespace swat {

constexpr int SK_WAVES = 8;

// One wave's share (an eighth of the reduction) of C[M <= 16 FM, 16 columns n0 ..] = A op(B): returns its partial accumulators.
// bounce: this wave's [2][1024]-byte LDS scratch (NN only).
template <int LAYOUT, int FM>
__device__ __forceinline__ void skinny_partial(const GemmArgs& p, int n0, int wave, int lane, unsigned char* bounce, f32x4_t (&acc)[FM][1]) {
    static_assert(LAYOUT == L_NT || LAYOUT == L_NN, "forward and data-gradient products");
    const int r16 = lane & 15, g = lane >> 4;
    const int M = p.M, N = p.N;
    const int nhw = (p.K >> 5) / SK_WAVES;          // 32-deep half steps per wave (K % 256 == 0)
    const int h0 = wave * nhw;
    // A fragments: lane (r16, g) reads k = 32 h + 8 g .. + 7 of row 16 i + r16 (rows beyond M: the last row, masked in the epilogue)
    const unsigned short* arow[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = 16 * i + r16;
        m = m < M ? m : M - 1;
        arow[i] = p.A + (int64_t)m * p.lda + 8 * g + 32 * h0;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const unsigned short* bsrc;
    int64_t bstep;
    if constexpr (LAYOUT == L_NT) {
        int n = n0 + r16;
        n = n < N ? n : N - 1;
        bsrc = p.B + (int64_t)n * p.ldb + 8 * g + 32 * h0;   // lane (r16, g): k = 32 h + 8 g .. + 7 of weight row n
        bstep = 32;
    } else {
        // B = W [K][N]: lane loads 16 B = columns 8 (lane & 1) .. + 7 of k-row 32 h + (lane >> 1) — 32 rows x 32 B per half step —
        // parks them row-major in the wave's bounce buffer; the fragment (8 consecutive k of column r16) comes back through two
        // transposed reads
        int nc = n0 + 8 * (lane & 1);
        nc = nc + 8 <= p.ldb ? nc : 0;   // columns beyond the matrix: any readable address (masked in the epilogue)
        bstep = (int64_t)32 * p.ldb;
        bsrc = p.B + (int64_t)(lane >> 1) * p.ldb + nc + h0 * bstep;
    }
    const unsigned wr_off = (unsigned)((lane >> 1) * 32 + 16 * (lane & 1));
    const unsigned rd_lo = (unsigned)((8 * g + (r16 >> 2)) * 32 + 8 * (r16 & 3)), rd_hi = rd_lo + 4 * 32;
    // groups of U half steps: all of a group's loads (U x (FM + 1) x 16 B per lane) go out before its first MFMA; the eight waves of
    // the workgroup (each on its own eighth of the reduction) hide one another's latencies
    constexpr int U = 8;
    for (int hb = 0; hb < nhw; hb += U) {
        bf16x8_t fa[U][FM];
        u32x4_t fb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int h = hb + u;
            h = h < nhw ? h : nhw - 1;
            fb[u] = *reinterpret_cast<const u32x4_t*>(bsrc + h * bstep);
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[u][i] = *reinterpret_cast<const bf16x8_t*>(arow[i] + 32 * h);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (hb + u < nhw) {
                bf16x8_t b;
                if constexpr (LAYOUT == L_NT) {
                    b = __builtin_bit_cast(bf16x8_t, fb[u]);
                } else {
                    unsigned char* buf = bounce + (u & 1) * 1024;
                    *reinterpret_cast<u32x4_t*>(buf + wr_off) = fb[u];
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(buf + rd_lo));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(buf + rd_hi));
                    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    b = __builtin_bit_cast(bf16x8_t, v);
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, fa[u][i], acc[i][0], 0, 0, 0);
            }
        }
    }
}

}  // namespace swat
