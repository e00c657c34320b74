// swat_amd/csrc/attention.hip — whole-head attention in one workgroup for short sequences (T <= 64, head width 64).
//
// ViT-B/32 at 224 px has T = 50 tokens: the batched-GEMM form of nn.MultiheadAttention (QK^T, softmax, PV and five
// more launches backward) is launch- and latency-bound there (~50 us per layer).  Here one 4-wave workgroup owns one
// (image, head): wave w computes query rows 16w..16w+15.  Every product is on the bf16 matrix cores
// (v_mfma_f32_16x16x32_bf16, fp32 accumulate) with the same rounding points as the batched form: P and dS are rounded
// to bf16 before they are multiplied again, the softmax statistics stay fp32.  The backward recomputes S and P from
// Q and K instead of reading a saved P.
//
// Operand layouts (mfma(X, Y): D[n][m] = sum_k X[n][k] Y[m][k]; lane (r16, g) of X supplies row n = r16, k = 8g..8g+7;
// a lane of D holds m = r16, n = 4g..4g+3):
//  * k-contiguous operands (Q, K, V, dO rows: 64 contiguous head columns) are loaded straight from global memory,
//    16 B per lane.
//  * operands whose reduction index runs down the rows of a matrix (V and K in P V / dS K, Q / dO / dS / P in the
//    transposed products) are staged in LDS tiles and read with ds_read_b64_tr_b16.
//  * P and dS feed the next product from the accumulator registers: a lane's 8 values of k-step ks are keys
//    32ks + 4g + {0..3} and 32ks + 16 + 4g + {0..3}; the LDS operand uses the same key order, so no shuffle is needed.
// ref: nn.MultiheadAttention inside open_clip's ResidualAttentionBlock, call sites utils/training.py:332,1474
#include "attention.hpp"
#include "../../include/swat_hip.h"

namespace swat {

void launch_attn_fwd_tiled2(const unsigned short* qkv, unsigned short* out, float* lse, int B, int T, int d, int H, int causal, hipStream_t s);  // attention_tiled2.hip

__global__ __launch_bounds__(256) void attn_fwd_small_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                             int T, int d, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short vt[64 * ATP];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const unsigned short* base = qkv + (int64_t)b * T * ld + h * 64;
    stage_tile(vt, base + 2 * d, ld, T);
    af32x4_t s[4];
    scores_softmax(base, ld, d, T, wave, r16, g, scale, s);
    const abf16x8_t pf[2] = {acc_frag(s, 0), acc_frag(s, 1)};
    __syncthreads();
    const int q = wave * 16 + r16;
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) {
        af32x4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(vt, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * jc, r16), pf[ks], o, 0, 0, 0);
        if (q < T) store4(out + ((int64_t)b * T + q) * d + h * 64 + 16 * jc + 4 * g, o);
    }
}

__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const unsigned short* __restrict__ qkv,
                                                             const unsigned short* __restrict__ dout,
                                                             unsigned short* __restrict__ dqkv, int T, int d, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short kt[64 * ATP], qt[64 * ATP], dot_[64 * ATP], dst[64 * ATP], pt[64 * ATP];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const unsigned short* base = qkv + (int64_t)b * T * ld + h * 64;
    const unsigned short* dob = dout + (int64_t)b * T * d + h * 64;
    unsigned short* dbase = dqkv + (int64_t)b * T * ld + h * 64;
    stage_tile(qt, base, ld, T);
    stage_tile(kt, base + d, ld, T);
    stage_tile(dot_, dob, d, T);

    af32x4_t p[4];
    scores_softmax(base, ld, d, T, wave, r16, g, scale, p);
    // dP = dO V^T for the wave's query rows
    const int q = wave * 16 + r16, qq = q < T ? q : T - 1;
    abf16x8_t dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) dof[ks] = ld_frag(dob + (int64_t)qq * d + 32 * ks + 8 * g);
    af32x4_t ds[4];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int key = 16 * j + r16, kk = key < T ? key : T - 1;
        ds[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            ds[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(base + 2 * d + (int64_t)kk * ld + 32 * ks + 8 * g), dof[ks], ds[j], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) dot += p[j][e] * ds[j][e];  // masked keys: p = 0
    }
    dot = xor_sum(dot);
    const bool qv = q < T;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = 16 * j + 4 * g + e;
            ds[j][e] = (qv && key < T) ? scale * p[j][e] * (ds[j][e] - dot) : 0.f;
            if (!qv) p[j][e] = 0.f;  // clamped duplicate rows must not enter the sums over queries
        }
        store4(dst + q * ATP + 16 * j + 4 * g, ds[j]);
        store4(pt + q * ATP + 16 * j + 4 * g, p[j]);
    }
    __syncthreads();  // kt / qt / dot_ staged, dst / pt complete
    // dQ = dS K for the wave's query rows
    const abf16x8_t dsf[2] = {acc_frag(ds, 0), acc_frag(ds, 1)};
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) {
        af32x4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(kt, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * jc, r16), dsf[ks], o, 0, 0, 0);
        if (qv) store4(dbase + (int64_t)q * ld + 16 * jc + 4 * g, o);
    }
    // dK = dS^T Q and dV = P^T dO for key rows 16w..16w+15 (reduction over the queries, natural k order)
    const int key = wave * 16 + r16;
    abf16x8_t dst_f[2], pt_f[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        dst_f[ks] = tr_frag(dst, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * wave, r16);
        pt_f[ks] = tr_frag(pt, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * wave, r16);
    }
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) {
        af32x4_t dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(qt, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * jc, r16), dst_f[ks], dk, 0, 0, 0);
            dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(dot_, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * jc, r16), pt_f[ks], dv, 0, 0, 0);
        }
        if (key < T) {
            store4(dbase + d + (int64_t)key * ld + 16 * jc + 4 * g, dk);
            store4(dbase + 2 * d + (int64_t)key * ld + 16 * jc + 4 * g, dv);
        }
    }
}

// (Longer sequences, 64 < T <= 320: csrc/attention_tiled2.hip behind swat_attn_fwd_tiled / swat_attn_bwd_tiled.)

}  // namespace swat

SWAT_API int swat_attn_fwd_small(const void* qkv_bf16, void* out_bf16, int B, int T, int d, int H, swat_stream_t stream) {
    SWAT_REQUIRE(qkv_bf16 && out_bf16 && B > 0 && T > 0 && T <= 64 && H > 0 && d == 64 * H,
                 "swat_attn_fwd_small: needs T <= 64 and head width 64 (T=%d d=%d H=%d)", T, d, H);
    hipLaunchKernelGGL(swat::attn_fwd_small_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)qkv_bf16,
                       (unsigned short*)out_bf16, T, d, H, 0.125f);
    return swat::check_launch("swat_attn_fwd_small");
}

SWAT_API int swat_attn_bwd_small(const void* qkv_bf16, const void* dout_bf16, void* dqkv_bf16, int B, int T, int d, int H,
                                 swat_stream_t stream) {
    SWAT_REQUIRE(qkv_bf16 && dout_bf16 && dqkv_bf16 && B > 0 && T > 0 && T <= 64 && H > 0 && d == 64 * H,
                 "swat_attn_bwd_small: needs T <= 64 and head width 64 (T=%d d=%d H=%d)", T, d, H);
    hipLaunchKernelGGL(swat::attn_bwd_small_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)qkv_bf16,
                       (const unsigned short*)dout_bf16, (unsigned short*)dqkv_bf16, T, d, H, 0.125f);
    return swat::check_launch("swat_attn_bwd_small");
}

SWAT_API int swat_attn_fwd_tiled(const void* qkv_bf16, void* out_bf16, float* lse, int B, int T, int d, int H, int causal,
                                 swat_stream_t stream) {
    SWAT_REQUIRE(qkv_bf16 && out_bf16 && B > 0 && T > 0 && T <= 320 && H > 0 && d == 64 * H,
                 "swat_attn_fwd_tiled: needs T <= 320 and head width 64 (T=%d d=%d H=%d)", T, d, H);
    swat::launch_attn_fwd_tiled2((const unsigned short*)qkv_bf16, (unsigned short*)out_bf16, lse, B, T, d, H, causal, (hipStream_t)stream);
    return swat::check_launch("swat_attn_fwd_tiled");
}
