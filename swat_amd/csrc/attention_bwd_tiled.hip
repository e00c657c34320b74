// swat_amd/csrc/attention_bwd_tiled.hip — entry point of the attention backward for 64 < T <= 320, head width 64: two launches
// (csrc/attention_tiled2.hip): dQ per row tile of queries (also leaves delta = rowsum(dO * O) in the workspace), then dK / dV per
// row tile of keys.  Both recompute P = exp(scale * Q K^T - lse) from the forward's row log-sum-exp; every output element is
// summed by one wave in a fixed order (no atomics).
// ref: autograd of nn.MultiheadAttention inside open_clip's ResidualAttentionBlock
#include "attention.hpp"
#include "../../include/swat_hip.h"

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
