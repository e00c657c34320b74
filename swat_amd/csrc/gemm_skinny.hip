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
#include "gemm_skinny_device.hpp"
#include "../../include/swat_hip.h"

namespace swat {


template <int LAYOUT, int FM>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_skinny_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char bounce[SK_WAVES][2][32 * 32];   // NN: [32 k][16 n] bf16 per wave and 32-deep step
    __shared__ __attribute__((aligned(16))) float part[SK_WAVES][FM][64][4];             // the waves' partial accumulators
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 16;
    f32x4_t acc[FM][1];
    skinny_partial<LAYOUT, FM>(p, n0, wave, lane, &bounce[wave][0][0], acc);
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
bool gemm_skinny_serves(int layout, const GemmArgs& a) {
    if ((layout != L_NT && layout != L_NN) || a.M < 1 || a.M > 64 || a.N < 1 || a.K % (32 * SK_WAVES) || a.K < 32 * SK_WAVES || a.inner > 1 || a.lda % 8 || a.ldb % 8 ||
        a.rpg || a.rowbias || a.colsum || ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15))
        return false;
    return !(layout == L_NN && a.N % 8);
}

int launch_gemm_skinny(int layout, const GemmArgs& a, hipStream_t s) {
    if (!gemm_skinny_serves(layout, a)) return 1;
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
