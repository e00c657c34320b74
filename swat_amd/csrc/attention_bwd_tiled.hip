// swat_amd/csrc/attention_bwd_tiled.hip — attention backward for 64 < T <= 64 * NT (NT <= 5), head width 64.
//
// One 4-wave workgroup per (image, head) walks every (query tile, key tile) pair.  For a pair the wave recomputes its
// 16 x 64 block of P = exp(scale * Q K^T - lse) (lse saved by the forward), dP = dO V^T and dS = scale * P * (dP - delta)
// (delta = rowsum(dO * O)), adds dS K to the dQ accumulator of its 16 query rows, and — after P and dS have been laid
// into LDS tiles — adds dS^T Q and P^T dO to the dK / dV accumulators of its 16 key rows of that key tile.  dK / dV live
// in registers for all NT key tiles (NT x 32 accumulator registers per lane) until the last query tile, so nothing is
// summed through memory and the result does not depend on scheduling.  Rounding points as in the batched form: P and dS
// are rounded to bf16 before they are multiplied again.
// ref: autograd of nn.MultiheadAttention inside open_clip's ResidualAttentionBlock
#include "attention.hpp"
#include "../../include/swat_hip.h"

namespace swat {

template <int NT>
__global__ __launch_bounds__(256) void attn_bwd_tiled_kernel(const unsigned short* __restrict__ qkv,
                                                             const unsigned short* __restrict__ out,
                                                             const unsigned short* __restrict__ dout,
                                                             const float* __restrict__ lse, unsigned short* __restrict__ dqkv,
                                                             int T, int d, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short kt_[64 * ATP], qt_[64 * ATP], dot_[64 * ATP], dst_[64 * ATP], pt_[64 * ATP];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const unsigned short* base = qkv + (int64_t)b * T * ld + h * 64;
    const unsigned short* ob = out + (int64_t)b * T * d + h * 64;
    const unsigned short* dob = dout + (int64_t)b * T * d + h * 64;
    unsigned short* dbase = dqkv + (int64_t)b * T * ld + h * 64;
    const float* lrow = lse + (int64_t)blockIdx.x * T;

    af32x4_t dk[NT][4], dv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) {
            dk[t][jc] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
            dv[t][jc] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
        }

#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        const int q0 = qt * 64;
        if (q0 >= T) break;
        __syncthreads();  // the previous query tile's last pair is done with qt_ / dot_
        stage_tile(qt_, base, ld, T, q0);
        stage_tile(dot_, dob, d, T, q0);
        const int q = q0 + wave * 16 + r16, qq = q < T ? q : T - 1;
        const bool qv = q < T;
        abf16x8_t qf[2], dof[2];
        float delta = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = ld_frag(base + (int64_t)qq * ld + 32 * ks + 8 * g);
            const au32x4_t dov = *reinterpret_cast<const au32x4_t*>(dob + (int64_t)qq * d + 32 * ks + 8 * g);
            const au32x4_t ov = *reinterpret_cast<const au32x4_t*>(ob + (int64_t)qq * d + 32 * ks + 8 * g);
            dof[ks] = __builtin_bit_cast(abf16x8_t, dov);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                delta += __uint_as_float(dov[i] << 16) * __uint_as_float(ov[i] << 16) +
                         __uint_as_float(dov[i] & 0xffff0000u) * __uint_as_float(ov[i] & 0xffff0000u);
        }
        delta = xor_sum(delta);
        const float lq = lrow[qq];
        af32x4_t dq[4];
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) dq[jc] = (af32x4_t){0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const int k0 = kt * 64;
            if (k0 < T) {
                __syncthreads();  // the previous pair is done with kt_ / dst_ / pt_ (and qt_ / dot_ are staged)
                stage_tile(kt_, base + d, ld, T, k0);
                af32x4_t p[4], ds[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = k0 + 16 * j + r16, kk = key < T ? key : T - 1;
                    p[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
                    ds[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        p[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(base + d + (int64_t)kk * ld + 32 * ks + 8 * g), qf[ks], p[j], 0, 0, 0);
                        ds[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(base + 2 * d + (int64_t)kk * ld + 32 * ks + 8 * g), dof[ks], ds[j], 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kcol = k0 + 16 * j + 4 * g + e;
                        const bool ok = qv && kcol < T;
                        const float pv = ok ? bf2f(f2bf(__expf(p[j][e] * scale - lq))) : 0.f;
                        p[j][e] = pv;
                        ds[j][e] = ok ? scale * pv * (ds[j][e] - delta) : 0.f;
                    }
                    store4(dst_ + (wave * 16 + r16) * ATP + 16 * j + 4 * g, ds[j]);
                    store4(pt_ + (wave * 16 + r16) * ATP + 16 * j + 4 * g, p[j]);
                }
                __syncthreads();  // kt_ staged, dst_ / pt_ complete
                const abf16x8_t dsf[2] = {acc_frag(ds, 0), acc_frag(ds, 1)};
                abf16x8_t dst_f[2], pt_f[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    dst_f[ks] = tr_frag(dst_, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * wave, r16);
                    pt_f[ks] = tr_frag(pt_, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * wave, r16);
                }
#pragma unroll
                for (int jc = 0; jc < 4; ++jc)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        dq[jc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(kt_, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * jc, r16), dsf[ks], dq[jc], 0, 0, 0);
                        dk[kt][jc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(qt_, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * jc, r16), dst_f[ks], dk[kt][jc], 0, 0, 0);
                        dv[kt][jc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(dot_, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, 16 * jc, r16), pt_f[ks], dv[kt][jc], 0, 0, 0);
                    }
            }
        }
        if (qv) {
#pragma unroll
            for (int jc = 0; jc < 4; ++jc) store4(dbase + (int64_t)q * ld + 16 * jc + 4 * g, dq[jc]);
        }
    }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const int key = kt * 64 + wave * 16 + r16;
        if (key < T) {
#pragma unroll
            for (int jc = 0; jc < 4; ++jc) {
                store4(dbase + d + (int64_t)key * ld + 16 * jc + 4 * g, dk[kt][jc]);
                store4(dbase + 2 * d + (int64_t)key * ld + 16 * jc + 4 * g, dv[kt][jc]);
            }
        }
    }
}

}  // namespace swat

SWAT_API int swat_attn_bwd_tiled(const void* qkv_bf16, const void* out_bf16, const void* dout_bf16, const float* lse,
                                 void* dqkv_bf16, int B, int T, int d, int H, swat_stream_t stream) {
    SWAT_REQUIRE(qkv_bf16 && out_bf16 && dout_bf16 && lse && dqkv_bf16 && B > 0 && T > 0 && T <= 320 && H > 0 && d == 64 * H,
                 "swat_attn_bwd_tiled: needs T <= 320 and head width 64 (T=%d d=%d H=%d)", T, d, H);
    const int nt = (T + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
#define SWAT_ATTN_BWD(NT_)                                                                                             \
    hipLaunchKernelGGL(swat::attn_bwd_tiled_kernel<NT_>, dim3(B * H), dim3(256), 0, s, (const unsigned short*)qkv_bf16, \
                       (const unsigned short*)out_bf16, (const unsigned short*)dout_bf16, lse, (unsigned short*)dqkv_bf16, T, d, H, 0.125f)
    switch (nt) {
        case 1: SWAT_ATTN_BWD(1); break;
        case 2: SWAT_ATTN_BWD(2); break;
        case 3: SWAT_ATTN_BWD(3); break;
        case 4: SWAT_ATTN_BWD(4); break;
        default: SWAT_ATTN_BWD(5); break;
    }
#undef SWAT_ATTN_BWD
    return swat::check_launch("swat_attn_bwd_tiled");
}
