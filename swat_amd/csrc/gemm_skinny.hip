// swat_amd/csrc/gemm_skinny.hip — C[M <= 64, N] = A[M, K] op(B): the rows a many-row product has beyond its last whole 256-row tile.
//
// ref: the same nn.Linear forward / data-gradient products as csrc/gemm_xl.hip (utils/training.py:1474, 1486-1488).  ViT-L/14 at
// B = 32 has B*T = 8 224 = 32 x 256 + 32 rows: the 256 x 128 kernel takes the 8 192, and these 32 rows were first handed to the
// small-tile kernels — N / 64 workgroups walking the whole reduction one 64-deep step at a time: 56 us per data gradient, 5.4 ms per
// step, a fifth of all kernel time for 0.4 % of the flops.
//
// Here a workgroup owns 16 output columns and its EIGHT waves each an eighth of the reduction: the weight slice [16][K] (NT) or
// [K][16] (NN) streams from L2 once, straight into MFMA B fragments (NN: through a 1 KiB LDS bounce per wave and 32-deep step and
// ds_read_b64_tr_b16), against the M <= 64 rows of A re-read from L2 by every workgroup (M K 2 bytes: a few hundred KB) — N / 16
// workgroups spread over the chip, eight waves each to hide one another's load latencies (a first form with ONE wave per 16 columns
// and a register double buffer took 55 us per data gradient: hipcc answers loads that are in flight across a loop's back edge with
// s_waitcnt vmcnt(0)).  The eight partial sums meet in LDS and are added in wave order: no workspace, no atomics, results
// bit-identical from run to run.
// Epilogue: gemm_epilogue of gemm_device.hpp (bias, fp32 residual, activation + second output, dgrad x act', bf16 / fp32).
#include <hip/hip_ext.h>
#include <type_traits>
#include "gemm_device.hpp"
#include "../../include/swat_hip.h"

namespace swat {

constexpr int SK_WAVES = 8;

template <int LAYOUT, int FM>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_skinny_kernel(const GemmArgs p) {
    static_assert(LAYOUT == L_NT || LAYOUT == L_NN, "forward and data-gradient products");
    __shared__ __attribute__((aligned(16))) unsigned char bounce[SK_WAVES][2][32 * 32];   // NN: [32 k][16 n] bf16 per wave and 32-deep step
    __shared__ __attribute__((aligned(16))) float part[SK_WAVES][FM][64][4];             // the waves' partial accumulators
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 16;
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
    f32x4_t acc[FM][1];
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
                    unsigned char* buf = bounce[wave][u & 1];
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
    // the eight partial sums, added in wave order by wave 0 (a fixed order: results are bit-identical from run to run)
#pragma unroll
    for (int i = 0; i < FM; ++i) *reinterpret_cast<f32x4_t*>(&part[wave][i][lane][0]) = acc[i][0];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        f32x4_t s = *reinterpret_cast<const f32x4_t*>(&part[0][i][lane][0]);
#pragma unroll
        for (int w = 1; w < SK_WAVES; ++w) {
            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(&part[w][i][lane][0]);
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        acc[i][0] = s;
    }
    gemm_epilogue<FM, 1>(p, acc, 0, n0, lane, 0, 0);
}

// -> 0 launched, 1 not served (the caller takes another kernel), < 0 error
int launch_gemm_skinny(int layout, const GemmArgs& a, hipStream_t s) {
    if ((layout != L_NT && layout != L_NN) || a.M < 1 || a.M > 64 || a.N < 1 || a.K % (32 * SK_WAVES) || a.K < 32 * SK_WAVES || a.inner > 1 || a.lda % 8 || a.ldb % 8 ||
        a.rpg || a.rowbias || a.colsum || ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15))
        return 1;
    if (layout == L_NN && a.N % 8) return 1;
    const dim3 grid((a.N + 15) / 16), block(64 * SK_WAVES);
    const bool two = a.M <= 32;
    if (layout == L_NT) {
        if (two) hipLaunchKernelGGL((gemm_skinny_kernel<L_NT, 2>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_skinny_kernel<L_NT, 4>), grid, block, 0, s, a);
    } else {
        if (two) hipLaunchKernelGGL((gemm_skinny_kernel<L_NN, 2>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_skinny_kernel<L_NN, 4>), grid, block, 0, s, a);
    }
    return check_launch("gemm_skinny_kernel");
}

}  // namespace swat
