// swat_amd/csrc/attention.hpp — operand helpers shared by the attention kernels (attention.hip, attention_bwd_tiled.hip).
//
// mfma(X, Y): D[n][m] = sum_k X[n][k] Y[m][k]; lane (r16, g) of X supplies row n = r16, k = 8g..8g+7; a lane of D holds
// m = r16, n = 4g..4g+3.  k-contiguous operands come straight from global memory (ld_frag), operands whose reduction
// index runs down the rows of a matrix come from 64 x 64 LDS tiles through ds_read_b64_tr_b16 (tr_frag), and a 16 x 64
// accumulator block feeds the next product from registers (acc_frag: the lane's 8 values of k-step ks are columns
// 32ks + 4g + {0..3} and 32ks + 16 + 4g + {0..3}; the LDS operand of that product is read in the same order).
#pragma once
#include "common.hpp"

namespace swat {

typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8_t;
typedef __attribute__((ext_vector_type(4))) float af32x4_t;
typedef __attribute__((ext_vector_type(4))) short as16x4_t;
typedef __attribute__((ext_vector_type(8))) short as16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int au32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int au32x2_t;

constexpr int ATP = 72;  // LDS tile pitch in elements (144 B: 16-B aligned rows that step 4 banks)

__device__ __forceinline__ abf16x8_t ld_frag(const unsigned short* p) {
    return __builtin_bit_cast(abf16x8_t, *reinterpret_cast<const au32x4_t*>(p));
}

// lane (r16, g) <- tile[krow0 + 0..3][n0 + r16], tile[krow1 + 0..3][n0 + r16]
__device__ __forceinline__ abf16x8_t tr_frag(const unsigned short* tile, int krow0, int krow1, int n0, int r16) {
    const unsigned short* a0 = tile + (krow0 + (r16 >> 2)) * ATP + n0 + 4 * (r16 & 3);
    const unsigned short* a1 = tile + (krow1 + (r16 >> 2)) * ATP + n0 + 4 * (r16 & 3);
    const as16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) as16x4_t*)a0);
    const as16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) as16x4_t*)a1);
    const as16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(abf16x8_t, v);
}

// stage rows row0..row0+63 (clamped to T-1) x 64 head columns of one operand into an LDS tile (256 threads)
__device__ __forceinline__ void stage_tile(unsigned short* tile, const unsigned short* src, int64_t ld, int T, int row0 = 0) {
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {
        const int row = i >> 3, ch = i & 7;
        const int rr = row0 + row < T ? row0 + row : T - 1;
        *reinterpret_cast<au32x4_t*>(tile + row * ATP + ch * 8) = *reinterpret_cast<const au32x4_t*>(src + (int64_t)rr * ld + ch * 8);
    }
}

__device__ __forceinline__ float xor_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float xor_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

__device__ __forceinline__ abf16x8_t acc_frag(const af32x4_t (&s)[4], int ks) {
    au32x4_t v;
    v[0] = pack_bf16x2(s[2 * ks][0], s[2 * ks][1]);
    v[1] = pack_bf16x2(s[2 * ks][2], s[2 * ks][3]);
    v[2] = pack_bf16x2(s[2 * ks + 1][0], s[2 * ks + 1][1]);
    v[3] = pack_bf16x2(s[2 * ks + 1][2], s[2 * ks + 1][3]);
    return __builtin_bit_cast(abf16x8_t, v);
}

__device__ __forceinline__ void store4(unsigned short* p, const af32x4_t& v) {
    au32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *reinterpret_cast<au32x2_t*>(p) = o;
}

// S = Q K^T for the wave's 16 query rows, then the row softmax; returns P (fp32, already rounded through bf16) in s[][]
__device__ __forceinline__ void scores_softmax(const unsigned short* base, int64_t ld, int d, int T, int wave, int r16, int g,
                                               float scale, af32x4_t (&s)[4]) {
    const int q = wave * 16 + r16, qq = q < T ? q : T - 1;
    abf16x8_t qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = ld_frag(base + (int64_t)qq * ld + 32 * ks + 8 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int key = 16 * j + r16, kk = key < T ? key : T - 1;
        s[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(base + d + (int64_t)kk * ld + 32 * ks + 8 * g), qf[ks], s[j], 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = 16 * j + 4 * g + e;
            s[j][e] = key < T ? s[j][e] * scale : -INFINITY;
            mx = fmaxf(mx, s[j][e]);
        }
    mx = xor_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = 16 * j + 4 * g + e;
            s[j][e] = key < T ? __expf(s[j][e] - mx) : 0.f;
            sum += s[j][e];
        }
    const float inv = 1.0f / xor_sum(sum);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[j][e] = bf2f(f2bf(s[j][e] * inv));
}

}  // namespace swat
