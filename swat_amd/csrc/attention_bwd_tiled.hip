// swat_amd/csrc/attention_bwd_tiled.hip — attention backward for 64 < T <= 320, head width 64, in two launches.
//
// Both kernels recompute, for a (query tile, key tile) pair, the wave's 16 x 64 block of P = exp(scale * Q K^T - lse)
// (lse saved by the forward), dP = dO V^T and dS = scale * P * (dP - delta) with delta = rowsum(dO * O):
//   * attn_bwd_dq_kernel   — grid (image*head, query tile): walks the key tiles, dQ += dS K for the wave's query rows.
//   * attn_bwd_dkdv_kernel — grid (image*head, key tile): walks the query tiles; P and dS go through LDS tiles so that
//     the wave owning 16 KEY rows can add dS^T Q and P^T dO to its dK / dV accumulators.
// Every output element is summed by one wave in a fixed order (no atomics); P and dS are rounded to bf16 before they
// are multiplied again, as in the batched form.  (One workgroup per head doing all tile pairs needs 320+ accumulator
// registers per lane at T = 257 and left the chip at 4 waves per CU: 240 us per ViT-L/14 layer vs these two.)
// ref: autograd of nn.MultiheadAttention inside open_clip's ResidualAttentionBlock
#include "attention.hpp"
#include "../../include/swat_hip.h"

namespace swat {

// the wave's query rows of tile q0: Q / dO operand fragments, delta and lse of row q (lane r16)
struct QRows {
    abf16x8_t qf[2], dof[2];
    float delta, lq;
    bool valid;
};
__device__ __forceinline__ QRows load_qrows(const unsigned short* base, const unsigned short* ob, const unsigned short* dob,
                                            const float* lrow, int64_t ld, int d, int T, int q, int g) {
    QRows r;
    const int qq = q < T ? q : T - 1;
    r.valid = q < T;
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        r.qf[ks] = ld_frag(base + (int64_t)qq * ld + 32 * ks + 8 * g);
        const au32x4_t dov = *reinterpret_cast<const au32x4_t*>(dob + (int64_t)qq * d + 32 * ks + 8 * g);
        const au32x4_t ov = *reinterpret_cast<const au32x4_t*>(ob + (int64_t)qq * d + 32 * ks + 8 * g);
        r.dof[ks] = __builtin_bit_cast(abf16x8_t, dov);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            delta += __uint_as_float(dov[i] << 16) * __uint_as_float(ov[i] << 16) +
                     __uint_as_float(dov[i] & 0xffff0000u) * __uint_as_float(ov[i] & 0xffff0000u);
    }
    r.delta = xor_sum(delta);
    r.lq = lrow[qq];
    return r;
}

// P and dS blocks (fp32 holding bf16-rounded P, unrounded dS) of the wave's rows against key tile k0
__device__ __forceinline__ void pair_blocks(const unsigned short* base, int64_t ld, int d, int T, int k0, int r16, int g,
                                            float scale, const QRows& qr, af32x4_t (&p)[4], af32x4_t (&ds)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int key = k0 + 16 * j + r16, kk = key < T ? key : T - 1;
        p[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
        ds[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            p[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(base + d + (int64_t)kk * ld + 32 * ks + 8 * g), qr.qf[ks], p[j], 0, 0, 0);
            ds[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(base + 2 * d + (int64_t)kk * ld + 32 * ks + 8 * g), qr.dof[ks], ds[j], 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kcol = k0 + 16 * j + 4 * g + e;
            const bool ok = qr.valid && kcol < T;
            const float pv = ok ? bf2f(f2bf(__expf(p[j][e] * scale - qr.lq))) : 0.f;
            p[j][e] = pv;
            ds[j][e] = ok ? scale * pv * (ds[j][e] - qr.delta) : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ out,
                                                          const unsigned short* __restrict__ dout, const float* __restrict__ lse,
                                                          unsigned short* __restrict__ dqkv, int T, int d, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short kt_[64 * ATP];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const unsigned short* base = qkv + (int64_t)b * T * ld + h * 64;
    const int q = blockIdx.y * 64 + wave * 16 + r16;
    const QRows qr = load_qrows(base, out + (int64_t)b * T * d + h * 64, dout + (int64_t)b * T * d + h * 64,
                                lse + (int64_t)blockIdx.x * T, ld, d, T, q, g);
    af32x4_t dq[4];
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) dq[jc] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < T; k0 += 64) {
        __syncthreads();  // the previous key tile is no longer read
        stage_tile(kt_, base + d, ld, T, k0);
        af32x4_t p[4], ds[4];
        pair_blocks(base, ld, d, T, k0, r16, g, scale, qr, p, ds);
        const abf16x8_t dsf[2] = {acc_frag(ds, 0), acc_frag(ds, 1)};
        __syncthreads();  // K tile staged
#pragma unroll
        for (int jc = 0; jc < 4; ++jc)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                dq[jc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(kt_, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * jc, r16), dsf[ks], dq[jc], 0, 0, 0);
    }
    if (qr.valid) {
        unsigned short* dst = dqkv + ((int64_t)b * T + q) * ld + h * 64;
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) store4(dst + 16 * jc + 4 * g, dq[jc]);
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ out,
                                                            const unsigned short* __restrict__ dout, const float* __restrict__ lse,
                                                            unsigned short* __restrict__ dqkv, int T, int d, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short qt_[64 * ATP], dot_[64 * ATP], dst_[64 * ATP], pt_[64 * ATP];
    const int b = blockIdx.x / H, h = blockIdx.x % H, k0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const unsigned short* base = qkv + (int64_t)b * T * ld + h * 64;
    const unsigned short* ob = out + (int64_t)b * T * d + h * 64;
    const unsigned short* dob = dout + (int64_t)b * T * d + h * 64;
    const float* lrow = lse + (int64_t)blockIdx.x * T;
    af32x4_t dk[4], dv[4];
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) {
        dk[jc] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
        dv[jc] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    for (int q0 = 0; q0 < T; q0 += 64) {
        __syncthreads();  // the previous query tile's LDS tiles are no longer read
        stage_tile(qt_, base, ld, T, q0);
        stage_tile(dot_, dob, d, T, q0);
        const QRows qr = load_qrows(base, ob, dob, lrow, ld, d, T, q0 + wave * 16 + r16, g);
        af32x4_t p[4], ds[4];
        pair_blocks(base, ld, d, T, k0, r16, g, scale, qr, p, ds);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            store4(dst_ + (wave * 16 + r16) * ATP + 16 * j + 4 * g, ds[j]);
            store4(pt_ + (wave * 16 + r16) * ATP + 16 * j + 4 * g, p[j]);
        }
        __syncthreads();  // qt_ / dot_ staged, dst_ / pt_ complete
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const abf16x8_t dsf = tr_frag(dst_, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * wave, r16);
            const abf16x8_t pf = tr_frag(pt_, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * wave, r16);
#pragma unroll
            for (int jc = 0; jc < 4; ++jc) {
                dk[jc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(qt_, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * jc, r16), dsf, dk[jc], 0, 0, 0);
                dv[jc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(dot_, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * jc, r16), pf, dv[jc], 0, 0, 0);
            }
        }
    }
    const int key = k0 + wave * 16 + r16;
    if (key < T) {
        unsigned short* dst = dqkv + ((int64_t)b * T + key) * ld + h * 64;
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) {
            store4(dst + d + 16 * jc + 4 * g, dk[jc]);
            store4(dst + 2 * d + 16 * jc + 4 * g, dv[jc]);
        }
    }
}

}  // namespace swat

namespace swat {
void launch_attn_bwd_dq2(const unsigned short* qkv, const unsigned short* out, const unsigned short* dout, const float* lse,
                         unsigned short* dqkv, float* delta, int B, int T, int d, int H, hipStream_t s);  // attention_tiled2.hip
void launch_attn_bwd_dkdv2(const unsigned short* qkv, const unsigned short* dout, const float* lse, const float* delta,
                           unsigned short* dqkv, int B, int T, int d, int H, hipStream_t s);
}

SWAT_API int64_t swat_attn_bwd_tiled_workspace_bytes(int B, int T, int H) { return (int64_t)B * H * T * 4 + 256; }

SWAT_API int swat_attn_bwd_tiled(const void* qkv_bf16, const void* out_bf16, const void* dout_bf16, const float* lse,
                                 void* dqkv_bf16, int B, int T, int d, int H, void* workspace, swat_stream_t stream) {
    SWAT_REQUIRE(qkv_bf16 && out_bf16 && dout_bf16 && lse && dqkv_bf16 && workspace && B > 0 && T > 0 && T <= 320 && H > 0 && d == 64 * H,
                 "swat_attn_bwd_tiled: needs T <= 320 and head width 64 (T=%d d=%d H=%d)", T, d, H);
    hipStream_t s = (hipStream_t)stream;
    swat::launch_attn_bwd_dq2((const unsigned short*)qkv_bf16, (const unsigned short*)out_bf16, (const unsigned short*)dout_bf16, lse,
                              (unsigned short*)dqkv_bf16, (float*)workspace, B, T, d, H, s);
    swat::launch_attn_bwd_dkdv2((const unsigned short*)qkv_bf16, (const unsigned short*)dout_bf16, lse, (const float*)workspace,
                                (unsigned short*)dqkv_bf16, B, T, d, H, s);
    return swat::check_launch("swat_attn_bwd_tiled");
}
