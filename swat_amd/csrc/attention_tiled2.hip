// swat_amd/csrc/attention_tiled2.hip — the tiled attention kernels (64 < T <= 320, head width 64; round 3): forward, dQ and
// dK / dV behind swat_attn_fwd_tiled / swat_attn_bwd_tiled (entry points in attention.hip / attention_bwd_tiled.hip).
// ref: nn.MultiheadAttention inside open_clip's ResidualAttentionBlock, call sites
// utils/training.py:332,1474 (+ autograd).  fp32 statistics, P rounded to bf16 where it becomes an MFMA operand (and dS likewise);
// unlike the T <= 64 kernels, dS is formed from the unrounded P.  What changed against round 2 is the data movement and the
// per-score instruction count, which bound these kernels (MFMA busy ~13 % at T = 257):
//   * a workgroup holds W = 4..9 waves of 16 rows (8 when T > 144), so a staged tile serves up to 144 rows; waves whose rows
//     all lie past T skip the arithmetic;
//   * the tiles of the OTHER side (K and V for forward / dQ; Q, dO, lse, delta for dK / dV) are staged in LDS, double-buffered:
//     the next tile's global loads are issued before the current tile's MFMAs and written to the other buffer after them — ONE
//     barrier per tile, no global load inside the MFMA chains (round 2 read every K / V fragment straight from global memory,
//     per wave, between two barriers);
//   * dK / dV gets S^T from the MFMA with the operand roles swapped, which is already the operand layout of the two products
//     that contract over queries: P and dS never pass through LDS;
//   * dQ's kernel also leaves delta = rowsum(dO * O) per row for the dK / dV kernel;
//   * per score: exp2 of one fma (scale and log2 e folded), masks only in the tile that holds T or a causal diagonal, tiles /
//     16-key blocks a wave cannot see skipped, the forward's row sums on the idle matrix pipe (an MFMA against ones).
// Measurements, counters and the dead ends (inline-asm max over MFMA results, occupancy hints, smaller workgroups): DESIGN.md section 4.
#include "attention.hpp"
#include "../../include/swat_hip.h"
#include <cstdlib>
#include <cstring>

namespace swat {

__device__ __forceinline__ abf16x8_t lds_frag(const unsigned short* p) {
    return __builtin_bit_cast(abf16x8_t, *(const __attribute__((address_space(3))) au32x4_t*)p);
}

// K and V rows k0 .. k0+63 (clamped to T-1) of one head: global -> registers (issue), registers -> LDS tiles (commit)
template <int NT>   // threads of the workgroup
struct KvStage {
    static constexpr int N = (1024 + NT - 1) / NT;   // 16-B chunks per thread: 2 tiles x 64 rows x 8 chunks
    au32x4_t r[N];
    __device__ __forceinline__ void issue(const unsigned short* base, int64_t ld, int d, int T, int k0, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + i * NT;
            if (c < 1024) {
                const int which = c >> 9, row = (c >> 3) & 63, ch = c & 7;
                const int rr = k0 + row < T ? k0 + row : T - 1;
                r[i] = *reinterpret_cast<const au32x4_t*>(base + (1 + which) * d + (int64_t)rr * ld + ch * 8);
            }
        }
    }
    __device__ __forceinline__ void commit(unsigned short* kt, unsigned short* vt, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + i * NT;
            if (c < 1024) {
                const int which = c >> 9, row = (c >> 3) & 63, ch = c & 7;
                *(__attribute__((address_space(3))) au32x4_t*)((which ? vt : kt) + row * ATP + ch * 8) = r[i];
            }
        }
    }
};


// one key tile of the forward for the wave's 16 queries.  MASKED: the tile holds keys past T (rem < 64) or causal positions.
template <bool MASKED>
__device__ __forceinline__ void fwd2_tile(const unsigned short* ktile, const unsigned short* vtile, const abf16x8_t (&qf)[2], af32x4_t (&o)[4],
                                          af32x4_t& lacc, float& m, float c2, int rem, int causal, int k0, int q, int r16, int g) {
    af32x4_t s[4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
        if (!MASKED || 16 * j < rem) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(ktile + (16 * j + r16) * ATP + 32 * ks + 8 * g), qf[ks], s[j], 0, 0, 0);
            if (MASKED) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = 16 * j + 4 * g + e;
                    if (kk >= rem || (causal && k0 + kk > q)) s[j][e] = -INFINITY;
                }
            }
            mx = fmaxf(fmaxf(mx, fmaxf(s[j][0], s[j][1])), fmaxf(s[j][2], s[j][3]));   // (not inline asm: MFMA -> VALU hazard nops are the compiler's)
        }
    }
    const float mnew = fmaxf(m, xor_max(mx) * c2);   // finite from the first tile on: key 0 is never masked
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);
    m = mnew;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!MASKED || 16 * j < rem) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[j][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][e], c2, -mnew));   // exp2(-inf) = 0
        }
    }
    const abf16x8_t pf[2] = {acc_frag(s, 0), acc_frag(s, 1)};
    const abf16x8_t ones = __builtin_bit_cast(abf16x8_t, (au32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
#pragma unroll
    for (int e = 0; e < 4; ++e) lacc[e] *= alpha;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        if (!MASKED || 32 * ks < rem) lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[ks], lacc, 0, 0, 0);
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[jc][e] *= alpha;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            if (!MASKED || 32 * ks < rem)
                o[jc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(vtile, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * jc, r16), pf[ks], o[jc], 0, 0, 0);
    }
}


template <int W, int OCC>
__global__ __launch_bounds__(64 * W, OCC) void attn_fwd_tiled2_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                                 float* __restrict__ lse, int T, int d, int H, float scale, int causal) {
    constexpr int NT = 64 * W;
    __shared__ __attribute__((aligned(16))) unsigned short tiles[4 * 64 * ATP];   // [buf][K | V][64][ATP]
    const int b = blockIdx.x / H, h = blockIdx.x % H, q0 = blockIdx.y * 16 * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const unsigned short* base = qkv + (int64_t)b * T * ld + h * 64;
    const int q = q0 + wave * 16 + r16, qq = q < T ? q : T - 1;
    const bool active = q0 + wave * 16 < T;   // a wave whose 16 rows all lie past T only helps staging
    abf16x8_t qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = ld_frag(base + (int64_t)qq * ld + 32 * ks + 8 * g);
    af32x4_t o[4];
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) o[jc] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
    // running maximum in log2 units (m = max(s) * scale * log2 e: one fma + v_exp per element), row sums of the ROUNDED p in
    // an accumulator of their own (an MFMA against ones: the matrix pipe idles here, the VALU is the bound)
    float m = -INFINITY;
    af32x4_t lacc = {0.f, 0.f, 0.f, 0.f};
    const float c2 = scale * 1.44269504f;
    const int qhi = q0 + 16 * W - 1;   // last query row of the workgroup
    const int nkt = causal ? ((qhi < T ? qhi : T - 1) / 64 + 1) : (T + 63) / 64;
    KvStage<NT> st;
    st.issue(base, ld, d, T, 0, tid);
    st.commit(tiles, tiles + 64 * ATP, tid);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 64;
        const unsigned short* ktile = tiles + (kt & 1) * 2 * 64 * ATP;
        const unsigned short* vtile = ktile + 64 * ATP;
        if (kt + 1 < nkt) st.issue(base, ld, d, T, k0 + 64, tid);
        // keys of this tile the wave's 16 rows can see: up to T, and under the causal mask up to the wave's last row — 16-key blocks /
        // 32-key halves past that are skipped (T = 257: a 1-key last tile; T = 77 causal: rows 0-63 never touch the second tile)
        const int wrow = q0 + wave * 16;
        int rem = T - k0;
        if (causal && wrow + 16 - k0 < rem) rem = wrow + 16 - k0;
        if (!active || rem <= 0) {
        } else if (rem >= 64 && (!causal || k0 + 63 <= wrow)) fwd2_tile<false>(ktile, vtile, qf, o, lacc, m, c2, 64, 0, 0, 0, r16, g);
        else fwd2_tile<true>(ktile, vtile, qf, o, lacc, m, c2, rem, causal, k0, q, r16, g);
        if (kt + 1 < nkt) {
            unsigned short* nk = tiles + ((kt + 1) & 1) * 2 * 64 * ATP;
            st.commit(nk, nk + 64 * ATP, tid);
        }
        __syncthreads();   // the next tile is staged; this tile is no longer read
    }
    if (q < T) {
        const float l = lacc[0], inv = 1.0f / l;
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[jc][e] *= inv;
            store4(out + ((int64_t)b * T + q) * d + h * 64 + 16 * jc + 4 * g, o[jc]);
        }
        if (lse && g == 0) lse[(int64_t)blockIdx.x * T + q] = m * 0.69314718f + __logf(l);
    }
}

template <int W, int OCC>
__global__ __launch_bounds__(64 * W, OCC) void attn_bwd_dq2_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ out,
                                                              const unsigned short* __restrict__ dout, const float* __restrict__ lse,
                                                              unsigned short* __restrict__ dqkv, float* __restrict__ delta_out, int T, int d,
                                                              int H, float scale) {
    constexpr int NT = 64 * W;
    __shared__ __attribute__((aligned(16))) unsigned short tiles[4 * 64 * ATP];
    const int b = blockIdx.x / H, h = blockIdx.x % H, q0 = blockIdx.y * 16 * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const unsigned short* base = qkv + (int64_t)b * T * ld + h * 64;
    const unsigned short* ob = out + (int64_t)b * T * d + h * 64;
    const unsigned short* dob = dout + (int64_t)b * T * d + h * 64;
    const int q = q0 + wave * 16 + r16, qq = q < T ? q : T - 1;
    const bool valid = q < T, active = q0 + wave * 16 < T;
    KvStage<NT> st;
    st.issue(base, ld, d, T, 0, tid);
    abf16x8_t qf[2], dof[2];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = ld_frag(base + (int64_t)qq * ld + 32 * ks + 8 * g);
        const au32x4_t dov = *reinterpret_cast<const au32x4_t*>(dob + (int64_t)qq * d + 32 * ks + 8 * g);
        const au32x4_t ov = *reinterpret_cast<const au32x4_t*>(ob + (int64_t)qq * d + 32 * ks + 8 * g);
        dof[ks] = __builtin_bit_cast(abf16x8_t, dov);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dl += __uint_as_float(dov[i] << 16) * __uint_as_float(ov[i] << 16) +
                  __uint_as_float(dov[i] & 0xffff0000u) * __uint_as_float(ov[i] & 0xffff0000u);
    }
    const float delta = xor_sum(dl);
    const float c2 = scale * 1.44269504f, lq2 = lse[(int64_t)blockIdx.x * T + qq] * 1.44269504f, dsc = delta * scale;
    if (valid && g == 0) delta_out[(int64_t)blockIdx.x * T + q] = delta;   // the dK / dV kernel reads it
    st.commit(tiles, tiles + 64 * ATP, tid);
    __syncthreads();
    af32x4_t dq[4];
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) dq[jc] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nkt = (T + 63) / 64;
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 64;
        const unsigned short* ktile = tiles + (kt & 1) * 2 * 64 * ATP;
        const unsigned short* vtile = ktile + 64 * ATP;
        if (kt + 1 < nkt) st.issue(base, ld, d, T, k0 + 64, tid);
        if (active) {
            af32x4_t p[4], ds[4];
            const int rem = T - k0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
                ds[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
                if (16 * j < rem) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        p[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(ktile + (16 * j + r16) * ATP + 32 * ks + 8 * g), qf[ks], p[j], 0, 0, 0);
                        ds[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(vtile + (16 * j + r16) * ATP + 32 * ks + 8 * g), dof[ks], ds[j], 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)   // dS = P * scale * (dP - delta), P = exp2(s * scale * log2 e - lse * log2 e): four VALU ops
                        ds[j][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[j][e], c2, -lq2)) * __builtin_fmaf(ds[j][e], scale, -dsc);
                    if (rem < 64) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (16 * j + 4 * g + e >= rem) ds[j][e] = 0.f;   // keys past T
                    }
                }
            }
            const abf16x8_t dsf[2] = {acc_frag(ds, 0), acc_frag(ds, 1)};
#pragma unroll
            for (int jc = 0; jc < 4; ++jc)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    if (32 * ks < rem)
                        dq[jc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(ktile, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * jc, r16), dsf[ks], dq[jc], 0, 0, 0);
        }
        if (kt + 1 < nkt) {
            unsigned short* nk = tiles + ((kt + 1) & 1) * 2 * 64 * ATP;
            st.commit(nk, nk + 64 * ATP, tid);
        }
        __syncthreads();
    }
    if (valid) {
        unsigned short* dst = dqkv + ((int64_t)b * T + q) * ld + h * 64;
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) store4(dst + 16 * jc + 4 * g, dq[jc]);
    }
}


// Q and dO rows q0 .. q0+63 (clamped) of one head plus their lse / delta: global -> registers -> LDS
template <int NT>
struct QdoStage {
    static constexpr int N = (1024 + NT - 1) / NT;
    au32x4_t r[N];
    float stat;
    __device__ __forceinline__ void issue(const unsigned short* qbase, int64_t ld, const unsigned short* dob, int d, const float* lrow,
                                          const float* drow, int T, int q0, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + i * NT;
            if (c < 1024) {
                const int which = c >> 9, row = (c >> 3) & 63, ch = c & 7;
                const int rr = q0 + row < T ? q0 + row : T - 1;
                r[i] = *reinterpret_cast<const au32x4_t*>(which ? dob + (int64_t)rr * d + ch * 8 : qbase + (int64_t)rr * ld + ch * 8);
            }
        }
        if (tid < 128) {
            const int row = tid & 63, rr = q0 + row < T ? q0 + row : T - 1;
            stat = (tid < 64 ? lrow : drow)[rr];
        }
    }
    __device__ __forceinline__ void commit(unsigned short* qt, unsigned short* dot, float* stats, int tid, float mul) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + i * NT;
            if (c < 1024) {
                const int which = c >> 9, row = (c >> 3) & 63, ch = c & 7;
                *(__attribute__((address_space(3))) au32x4_t*)((which ? dot : qt) + row * ATP + ch * 8) = r[i];
            }
        }
        if (tid < 128) stats[tid] = stat * mul;   // lse in log2 units, delta times the softmax scale
    }
};

// dK, dV: the wave owns 16 KEY rows and walks the query tiles.  S^T comes out of the MFMA with the roles swapped
// (A = Q rows of the tile, B = the wave's K rows): the accumulator then has queries down its rows and the wave's keys across
// the lanes, which IS the operand layout of the two products that contract over queries (dV += P^T dO, dK += dS^T Q) —
// no trip of P / dS through LDS and no second barrier (the first form transposed both through LDS tiles).
template <int W, int OCC>
__global__ __launch_bounds__(64 * W, OCC) void attn_bwd_dkdv2_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                unsigned short* __restrict__ dqkv, int T, int d, int H, float scale) {
    constexpr int NT = 64 * W;
    __shared__ __attribute__((aligned(16))) unsigned short tiles[4 * 64 * ATP];   // [buf][Q | dO][64][ATP]
    __shared__ __attribute__((aligned(16))) float stats[2 * 128];                  // [buf][lse | delta][64]
    const int b = blockIdx.x / H, h = blockIdx.x % H, k0 = blockIdx.y * 16 * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const unsigned short* base = qkv + (int64_t)b * T * ld + h * 64;
    const unsigned short* dob = dout + (int64_t)b * T * d + h * 64;
    const float* lrow = lse + (int64_t)blockIdx.x * T;
    const float* drow = delta + (int64_t)blockIdx.x * T;
    const int key = k0 + wave * 16 + r16, kk = key < T ? key : T - 1;
    const bool active = k0 + wave * 16 < T;
    QdoStage<NT> st;
    st.issue(base, ld, dob, d, lrow, drow, T, 0, tid);
    abf16x8_t kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = ld_frag(base + d + (int64_t)kk * ld + 32 * ks + 8 * g);
        vf[ks] = ld_frag(base + 2 * d + (int64_t)kk * ld + 32 * ks + 8 * g);
    }
    const float c2 = scale * 1.44269504f, smul = tid < 64 ? 1.44269504f : scale;
    st.commit(tiles, tiles + 64 * ATP, stats, tid, smul);
    __syncthreads();
    af32x4_t dk[4], dv[4];
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) {
        dk[jc] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
        dv[jc] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const int nqt = (T + 63) / 64;
    for (int qt = 0; qt < nqt; ++qt) {
        const int q0 = qt * 64, rem = T - q0;
        const unsigned short* qtile = tiles + (qt & 1) * 2 * 64 * ATP;
        const unsigned short* dotile = qtile + 64 * ATP;
        const float* stt = stats + (qt & 1) * 128;
        if (qt + 1 < nqt) st.issue(base, ld, dob, d, lrow, drow, T, q0 + 64, tid);
        if (active) {
            af32x4_t p[4], ds[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
                ds[j] = (af32x4_t){0.f, 0.f, 0.f, 0.f};
                if (16 * j < rem) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        p[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(qtile + (16 * j + r16) * ATP + 32 * ks + 8 * g), kf[ks], p[j], 0, 0, 0);
                        ds[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(dotile + (16 * j + r16) * ATP + 32 * ks + 8 * g), vf[ks], ds[j], 0, 0, 0);
                    }
                    const af32x4_t lq2 = *(const __attribute__((address_space(3))) af32x4_t*)(stt + 16 * j + 4 * g);
                    const af32x4_t dsc = *(const __attribute__((address_space(3))) af32x4_t*)(stt + 64 + 16 * j + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        p[j][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[j][e], c2, -lq2[e]));
                        ds[j][e] = p[j][e] * __builtin_fmaf(ds[j][e], scale, -dsc[e]);
                    }
                    if (rem < 64) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (16 * j + 4 * g + e >= rem) p[j][e] = ds[j][e] = 0.f;   // queries past T
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (32 * ks < rem) {
                    const abf16x8_t pf = acc_frag(p, ks), dsf = acc_frag(ds, ks);
#pragma unroll
                    for (int jc = 0; jc < 4; ++jc) {
                        dv[jc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(dotile, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * jc, r16), pf, dv[jc], 0, 0, 0);
                        dk[jc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(qtile, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * jc, r16), dsf, dk[jc], 0, 0, 0);
                    }
                }
            }
        }
        if (qt + 1 < nqt) {
            unsigned short* nq = tiles + ((qt + 1) & 1) * 2 * 64 * ATP;
            st.commit(nq, nq + 64 * ATP, stats + ((qt + 1) & 1) * 128, tid, smul);
        }
        __syncthreads();
    }
    if (key < T) {
        unsigned short* dst = dqkv + ((int64_t)b * T + key) * ld + h * 64;
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) {
            store4(dst + d + 16 * jc + 4 * g, dk[jc]);
            store4(dst + 2 * d + 16 * jc + 4 * g, dv[jc]);
        }
    }
}

static int att_w();
// waves per workgroup and row tiles per (image, head): one workgroup of up to 9 waves when that covers T; otherwise 8-wave
// workgroups (two per SIMD: a 9-wave group leaves one SIMD with 3 waves and the other three waiting for it at every
// barrier), the last of which may hold waves with no rows — those only help staging (T = 257: 8 + 8 + 1 fragments)
static void tiled2_shape(int T, int& W, int& tiles) {
    const int frags = (T + 15) / 16;
    if (frags <= 9) { W = frags < 4 ? 4 : frags; tiles = 1; }
    else { W = 8; tiles = (frags + 7) / 8; }
    if (att_w() >= 4 && att_w() <= 9) { W = att_w(); tiles = (frags + W - 1) / W; }
}

#define SWAT_ATT2W(KERN, OCC, ...)                                                                      \
    switch (W) {                                                                                        \
        case 4: hipLaunchKernelGGL((KERN<4, OCC>), grid, dim3(256), 0, s, __VA_ARGS__); break;          \
        case 5: hipLaunchKernelGGL((KERN<5, OCC>), grid, dim3(320), 0, s, __VA_ARGS__); break;          \
        case 6: hipLaunchKernelGGL((KERN<6, OCC>), grid, dim3(384), 0, s, __VA_ARGS__); break;          \
        case 7: hipLaunchKernelGGL((KERN<7, OCC>), grid, dim3(448), 0, s, __VA_ARGS__); break;          \
        case 8: hipLaunchKernelGGL((KERN<8, OCC>), grid, dim3(512), 0, s, __VA_ARGS__); break;          \
        default: hipLaunchKernelGGL((KERN<9, OCC>), grid, dim3(576), 0, s, __VA_ARGS__); break;         \
    }
#define SWAT_ATT2(KERN, WHICH, ...)                                                                     \
    switch (att_occ(WHICH, W)) {                                                                           \
        case 3: SWAT_ATT2W(KERN, 3, __VA_ARGS__) break;                                                 \
        default: SWAT_ATT2W(KERN, 4, __VA_ARGS__) break;                                                \
    }

// tuning knob: SWAT_ATT_OCC = three digits (forward, dQ, dK/dV), each 3 or 4: waves per SIMD the kernels are compiled for
// (5-7 were measured too: the register limit spills and every kernel loses 10-60 %)
// (default: a 9-wave workgroup puts 3 waves on one SIMD and a second one never fits beside it below 5 per SIMD, so it may as
// well have 168 registers; 7- and 8-wave workgroups run two per CU at 4 per SIMD)
static int att_occ(int which, int W) {
    static int v[3] = {-1, 0, 0};
    if (v[0] < 0) {
        const char* e = getenv("SWAT_ATT_OCC");
        for (int i = 0; i < 3; ++i) v[i] = (e && strlen(e) == 3 && e[i] >= '3' && e[i] <= '4') ? e[i] - '0' : 0;
    }
    return v[which] ? v[which] : (W == 9 ? 3 : 4);
}
// experiment knob: SWAT_ATT_W = waves per workgroup (4..9), 0 = the rule below
static int att_w() {
    static int w = -1;
    if (w < 0) { const char* e = getenv("SWAT_ATT_W"); w = e ? atoi(e) : 0; }
    return w;
}

void launch_attn_fwd_tiled2(const unsigned short* qkv, unsigned short* out, float* lse, int B, int T, int d, int H, int causal, hipStream_t s) {
    int W, tiles;
    tiled2_shape(T, W, tiles);
    const dim3 grid(B * H, tiles);
    SWAT_ATT2(attn_fwd_tiled2_kernel, 0, qkv, out, lse, T, d, H, 0.125f, causal)
}

void launch_attn_bwd_dq2(const unsigned short* qkv, const unsigned short* out, const unsigned short* dout, const float* lse,
                         unsigned short* dqkv, float* delta, int B, int T, int d, int H, hipStream_t s) {
    int W, tiles;
    tiled2_shape(T, W, tiles);
    const dim3 grid(B * H, tiles);
    SWAT_ATT2(attn_bwd_dq2_kernel, 1, qkv, out, dout, lse, dqkv, delta, T, d, H, 0.125f)
}

void launch_attn_bwd_dkdv2(const unsigned short* qkv, const unsigned short* dout, const float* lse, const float* delta,
                           unsigned short* dqkv, int B, int T, int d, int H, hipStream_t s) {
    int W, tiles;
    tiled2_shape(T, W, tiles);
    const dim3 grid(B * H, tiles);
    SWAT_ATT2(attn_bwd_dkdv2_kernel, 2, qkv, dout, lse, delta, dqkv, T, d, H, 0.125f)
}

}  // namespace swat
