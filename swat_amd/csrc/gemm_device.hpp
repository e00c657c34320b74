// swat_amd/csrc/gemm_device.hpp — device code shared by the GEMM kernels (gemm.hip, gemm_pc.hip): the LDS image of a
// k-contiguous / k-strided operand tile, the fused epilogues and the XCD-aware tile order.
#pragma once
#include "kernels.hpp"

namespace swat {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ float act_fwd(float x, int act) {
    if (act == 1) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));  // v_exp + v_rcp, no IEEE division
    if (act == 2) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    return x;
}
__device__ __forceinline__ float act_bwd(float x, int act) {
    if (act == 1) {
        const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
        return s * (1.0f + 1.702f * x * (1.0f - s));
    }
    if (act == 2) {
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
        return cdf + x * pdf;
    }
    return 1.0f;
}

// ---- the shared LDS image --------------------------------------------------------------------------------------
template <int TILE_ROWS>
__device__ __forceinline__ int st_f(int k) {  // column-group swizzle of a k-strided tile
    if constexpr (TILE_ROWS == 128) return (k & 3) | ((k >> 1) & 4);
    else return ((k >> 1) & 1) | ((k >> 2) & 2);
}
// byte offset of 16-B chunk (row, slot t) of a k-contiguous tile
__device__ __forceinline__ int kc_off(int row, int t) { return row * 128 + ((t ^ ((row >> 1) & 7)) << 4); }
// byte offset of 16-B chunk (k-row, 16-B column chunk c16) of a k-strided tile with TILE_ROWS columns
template <int TILE_ROWS>
__device__ __forceinline__ int st_off(int k, int c16) {
    return k * (TILE_ROWS * 2) + ((((c16 >> 1) ^ st_f<TILE_ROWS>(k)) << 5) | ((c16 & 1) << 4));
}

// fragment of 8 bf16 for the 32-deep sub-step ks: lane (r16 = lane & 15, g = lane >> 4) gets k = 32 ks + 8 g + 0..7
template <bool KCONTIG, int TILE_ROWS>
__device__ __forceinline__ bf16x8_t read_frag(const unsigned char* tile, int base_row, int ks, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    if constexpr (KCONTIG) {
        const s16x8_t v = *reinterpret_cast<const s16x8_t*>(tile + kc_off(base_row + r16, 4 * ks + g));
        return __builtin_bit_cast(bf16x8_t, v);
    } else {
        constexpr int RS = TILE_ROWS * 2;
        const int q = r16 >> 2, pp = r16 & 3;
        const int k = 32 * ks + 8 * g + q;
        const int c32 = base_row >> 4;
        const unsigned char* lo_p = tile + k * RS + ((c32 ^ st_f<TILE_ROWS>(k)) << 5) + 8 * pp;
        const unsigned char* hi_p = tile + (k + 4) * RS + ((c32 ^ st_f<TILE_ROWS>(k + 4)) << 5) + 8 * pp;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lo_p));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(hi_p));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    }
}

// ---- k-strided tile of R columns as power-of-two parts -----------------------------------------------------------
__host__ __device__ __forceinline__ constexpr int stp_start(int R, int col) {  // first column of the part that holds column `col`
    int s = 0;
    while (R - s >= 128) {
        if (col < s + 128) return s;
        s += 128;
    }
    for (int w = 64; w >= 16; w >>= 1)
        if (R - s >= w) {
            if (col < s + w) return s;
            s += w;
        }
    return s;
}
__host__ __device__ __forceinline__ constexpr int stp_width(int R, int col) {
    int s = 0;
    while (R - s >= 128) {
        if (col < s + 128) return 128;
        s += 128;
    }
    for (int w = 64; w >= 16; w >>= 1)
        if (R - s >= w) {
            if (col < s + w) return w;
            s += w;
        }
    return 16;
}
__host__ __device__ __forceinline__ constexpr int stp_f(int w, int k) {  // 32-B group XOR of k-row k inside a part of width w
    return w == 128 ? ((k & 3) | ((k >> 1) & 4)) : w == 64 ? (((k >> 1) & 1) | ((k >> 2) & 2)) : w == 32 ? ((k >> 3) & 1) : 0;
}
// byte offset of the 16-B chunk (k-row k, columns col .. col+7; col % 8 == 0) in the image of a [64][R] tile
template <int R>
__host__ __device__ __forceinline__ constexpr int stp_off(int k, int col) {
    const int s = stp_start(R, col), w = stp_width(R, col);
    const int c32 = (col - s) >> 4, e = (col >> 3) & 1;
    const int unit = w == 16 ? (k ^ ((k >> 1) & 4)) : k * (w >> 4) + (c32 ^ stp_f(w, k));
    return s * 128 + unit * 32 + e * 16;
}
// inverse: which (k, col) lives at byte offset `byte` (multiple of 16) of the image
template <int R>
__host__ __device__ __forceinline__ constexpr void stp_inv(int byte, int& k, int& col) {
    const int s = stp_start(R, byte >> 7), w = stp_width(R, byte >> 7);  // part p covers bytes [128 s, 128 (s + w))
    const int rel = byte - s * 128, unit = rel >> 5, e = (rel >> 4) & 1;
    int c32 = 0;
    if (w == 16) k = unit ^ ((unit >> 1) & 4);
    else {
        const int lg = w == 128 ? 3 : w == 64 ? 2 : 1;
        k = unit >> lg;
        c32 = (unit & ((1 << lg) - 1)) ^ stp_f(w, k);
    }
    col = s + 16 * c32 + 8 * e;
}

// ---- epilogue shared by both kernels: lane holds C[m][n..n+3], m = mbase + 16 i + (lane & 15), n = nbase + 16 j + 4 (lane >> 4)
// bf16-output epilogue with 16-byte stores.  Two accumulator tiles that are neighbours in n trade halves with
// v_permlane16_swap: afterwards lane (r16, g) holds row r16 of tile j + (g & 1), columns 8 * (g >> 1) .. + 7 — eight
// consecutive n — so every bf16 output is ONE global_store_dwordx4 per lane where the 4-column layout needs two dwordx2
// (and the activation input one 16-B load).  Measured on the 1600 x 3072 outputs (tools/block_timeline.py): halving the
// store count took ~15 % off the epilogue; most of its time was arithmetic (IEEE division and software bf16 rounding,
// now v_rcp_f32 and v_cvt_pk_bf16_f32), 4.8 -> 2.8 us per round.  Staging the tile through LDS for whole-line stores
// bought nothing on top.  Loads of the row-wise operands run one accumulator row ahead of the stores (loads and stores
// share the in-order vmcnt counter).
template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_wide(const GemmArgs& p, f32x4_t (&acc)[FM][FN], int mbase, int nbase, int lane,
                                                   int zo, int zi) {
    static_assert(FN % 2 == 0, "tile pairs");
    constexpr int NP = FN / 2;
    const int M = p.M, N = p.N;
    const int r16 = lane & 15, g = lane >> 4;
    const int ncol = nbase + 16 * (g & 1) + 8 * (g >> 1);  // + 32 * pair
    const int64_t coff = zo * p.sC0 + zi * p.sC1;
    const bool full_epi = !(p.epi0 && zi != 0);
    const bool has_bias = p.bias && full_epi, has_rb = p.rowbias && full_epi, has_res = p.residual && full_epi;
    struct RowIn {
        float4 rb[NP][2], res[NP][2];
        u32x4_t f[NP];
        int orow;
    };
    float4 bias4[NP][2];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int n = ncol + 32 * q;
        bias4[q][0] = bias4[q][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_bias && n < N) {
            bias4[q][0] = *reinterpret_cast<const float4*>(p.bias + n);
            bias4[q][1] = *reinterpret_cast<const float4*>(p.bias + n + 4);
        }
    }
    auto load_row = [&](int i, RowIn& in) {
        const int m = mbase + 16 * i + r16;
        int orow = m, brow = 0;
        if (p.rpg > 0) {
            brow = (m % p.rpg) + p.roff;
            orow = (m / p.rpg) * p.gstride + brow;
        }
        in.orow = orow;
        if (m >= M) return;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int n = ncol + 32 * q;
            if (n >= N) continue;
            if (has_rb) {
                in.rb[q][0] = *reinterpret_cast<const float4*>(p.rowbias + (int64_t)brow * N + n);
                in.rb[q][1] = *reinterpret_cast<const float4*>(p.rowbias + (int64_t)brow * N + n + 4);
            }
            if (has_res) {
                in.res[q][0] = *reinterpret_cast<const float4*>(p.residual + coff + (int64_t)orow * p.ldc + n);
                in.res[q][1] = *reinterpret_cast<const float4*>(p.residual + coff + (int64_t)orow * p.ldc + n + 4);
            }
            if (p.actin) in.f[q] = *reinterpret_cast<const u32x4_t*>(p.actin + (int64_t)m * p.ldact + n);
        }
    };
    RowIn cur, nxt;
    load_row(0, cur);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if (i + 1 < FM) load_row(i + 1, nxt);
        const int m = mbase + 16 * i + r16;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // every lane takes part in the swap (no divergence above this point)
                // inline asm: hipcc (ROCm 7.2) folds a run of __builtin_amdgcn_permlane16_swap calls into the first one.
                // The two v_nop are the wait states the swap needs after a VALU write of its operands.
                float lo = acc[i][2 * q][e], hi = acc[i][2 * q + 1][e];
                asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                v[e] = lo * p.alpha;
                v[4 + e] = hi * p.alpha;
            }
            const int n = ncol + 32 * q;
            if (m >= M || n >= N) continue;
            if (has_bias) {
                v[0] += bias4[q][0].x; v[1] += bias4[q][0].y; v[2] += bias4[q][0].z; v[3] += bias4[q][0].w;
                v[4] += bias4[q][1].x; v[5] += bias4[q][1].y; v[6] += bias4[q][1].z; v[7] += bias4[q][1].w;
            }
            if (has_rb) {
                v[0] += cur.rb[q][0].x; v[1] += cur.rb[q][0].y; v[2] += cur.rb[q][0].z; v[3] += cur.rb[q][0].w;
                v[4] += cur.rb[q][1].x; v[5] += cur.rb[q][1].y; v[6] += cur.rb[q][1].z; v[7] += cur.rb[q][1].w;
            }
            if (has_res) {
                v[0] += cur.res[q][0].x; v[1] += cur.res[q][0].y; v[2] += cur.res[q][0].z; v[3] += cur.res[q][0].w;
                v[4] += cur.res[q][1].x; v[5] += cur.res[q][1].y; v[6] += cur.res[q][1].z; v[7] += cur.res[q][1].w;
            }
            if (p.actin) {  // dgrad through the activation: dF = dG * act'(F)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const unsigned w = cur.f[q][h];
                    v[2 * h] *= act_bwd(bf2f((unsigned short)(w & 0xffffu)), p.act);
                    v[2 * h + 1] *= act_bwd(bf2f((unsigned short)(w >> 16)), p.act);
                }
            }
            const int64_t o = coff + (int64_t)cur.orow * p.ldc + n;
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
        cur = nxt;
    }
}

template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4_t (&acc)[FM][FN], int mbase, int nbase, int lane,
                                              int zo, int zi) {
    if constexpr (FN % 2 == 0) {
        // 16-byte rows: every pointer the wide form touches with 16-B accesses is 16-B aligned at n % 8 == 0
        if (!p.out_f32 && p.N % 8 == 0 && p.ldc % 8 == 0 && (p.sC0 | p.sC1) % 8 == 0 && (!p.actin || p.ldact % 8 == 0) &&
            (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.C2) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(p.actin) & 15) == 0) {
            gemm_epilogue_wide<FM, FN>(p, acc, mbase, nbase, lane, zo, zi);
            return;
        }
    }
    // Loads and stores share one in-order counter (vmcnt): a load issued after a store cannot be waited for without
    // also waiting for that store's acknowledgement (~0.5 us in a burst).  So every operand this epilogue reads is
    // requested BEFORE the stores it could queue behind: the bias vectors once, and the row-wise operands (row bias,
    // residual, activation input) of accumulator row i+1 ahead of the stores of row i.
    const int M = p.M, N = p.N;
    const int r16 = lane & 15, g = lane >> 4;
    const int Np4 = (N + 3) & ~3;
    const int64_t coff = zo * p.sC0 + zi * p.sC1;
    const bool full_epi = !(p.epi0 && zi != 0);
    const bool has_bias = p.bias && full_epi, has_rb = p.rowbias && full_epi, has_res = p.residual && full_epi;
    struct RowIn {
        float4 rb[FN], res[FN];
        ushort4 f[FN];
        int orow;
    };
    float4 bias4[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = nbase + 16 * j + 4 * g;
        bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_bias && n < Np4) bias4[j] = *reinterpret_cast<const float4*>(p.bias + n);
    }
    auto load_row = [&](int i, RowIn& in) {
        const int m = mbase + 16 * i + r16;
        int orow = m, brow = 0;
        if (p.rpg > 0) {
            brow = (m % p.rpg) + p.roff;
            orow = (m / p.rpg) * p.gstride + brow;
        }
        in.orow = orow;
        if (m >= M) return;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = nbase + 16 * j + 4 * g;
            if (n >= Np4) continue;
            if (has_rb) in.rb[j] = *reinterpret_cast<const float4*>(p.rowbias + (int64_t)brow * N + n);
            if (has_res) in.res[j] = *reinterpret_cast<const float4*>(p.residual + coff + (int64_t)orow * p.ldc + n);
            if (p.actin) in.f[j] = *reinterpret_cast<const ushort4*>(p.actin + (int64_t)m * p.ldact + n);
        }
    };
    RowIn cur, nxt;
    load_row(0, cur);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if (i + 1 < FM) load_row(i + 1, nxt);
        const int m = mbase + 16 * i + r16;
        if (m < M) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = nbase + 16 * j + 4 * g;
                if (n >= Np4) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * p.alpha;
                if (has_bias) { v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w; }
                if (has_rb) { v[0] += cur.rb[j].x; v[1] += cur.rb[j].y; v[2] += cur.rb[j].z; v[3] += cur.rb[j].w; }
                if (has_res) { v[0] += cur.res[j].x; v[1] += cur.res[j].y; v[2] += cur.res[j].z; v[3] += cur.res[j].w; }
                if (p.actin) {  // dgrad through the activation: dF = dG * act'(F)
                    const ushort4 f4 = cur.f[j];
                    v[0] *= act_bwd(bf2f(f4.x), p.act); v[1] *= act_bwd(bf2f(f4.y), p.act);
                    v[2] *= act_bwd(bf2f(f4.z), p.act); v[3] *= act_bwd(bf2f(f4.w), p.act);
                }
                const int64_t o = coff + (int64_t)cur.orow * p.ldc + n;
                if (p.out_f32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    ushort4 o4 = make_ushort4(f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3]));
                    if (!p.c2_only) *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(p.C) + o) = o4;
                    if (p.C2) {  // second output: activation of the ROUNDED pre-activation (what backward re-reads)
                        ushort4 a4 = make_ushort4(f2bf(act_fwd(bf2f(o4.x), p.act)), f2bf(act_fwd(bf2f(o4.y), p.act)),
                                                  f2bf(act_fwd(bf2f(o4.z), p.act)), f2bf(act_fwd(bf2f(o4.w), p.act)));
                        *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(p.C2) + o) = a4;
                    }
                }
            }
        }
        cur = nxt;
    }
}



// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one, speed only — never correctness).
// Give each XCD a CONTIGUOUS chunk of the tile list, ordered so its tiles share A-row / B-column panels.
__device__ __forceinline__ void tile_coords(const GemmArgs& p, int& bx, int& by) {
    if (p.swz == 0) { bx = blockIdx.x; by = blockIdx.y; return; }
    const int nblk = p.tiles_m * p.tiles_n;
    const int lb = blockIdx.x;
    const int xcd = lb & 7, q = nblk >> 3, r = nblk & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lb >> 3);  // bijective for any nblk
    if (p.swz == 1) { by = id % p.tiles_m; bx = id / p.tiles_m; }
    else { bx = id % p.tiles_n; by = id / p.tiles_n; }
}

}  // namespace swat
