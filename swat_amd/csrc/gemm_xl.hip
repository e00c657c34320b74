// swat_amd/csrc/gemm_xl.hip — 256 x 128 bf16 MFMA tiles with split roles, persistent over a JOINT tile list of up to 8 products.
//
// ref: the nn.Linear forward / backward products under model.encode_image and loss.backward() (utils/training.py:1474,
// 1486-1488, :332-343): C = A B^T (NT, forward), dX = dY W (NN, data gradient), dW = dY^T X (TN, weight gradient).
//
// Two uses.  (1) ViT-B/32 / B/16 at B = 32: ALL four weight gradients of a transformer block as one launch — their outputs
// together are 216 tiles of 256 x 128, one resident round of the chip, where each product alone only fills the chip with
// 96 x 96 / 64 x 64 tiles that are bound by the per-CU L2 -> LDS path (65 us per block for four launches).
// (2) The many-row products (ViT-L/14: M = 8 224; feature extraction: M = 51 200): whole rounds of 256 x 128 tiles.
//
// Structure (what the measurements of the first form of this kernel — every wave loading AND computing — asked for: its
// main loop ran at half the rate its DMA, LDS and MFMA streams each allow, because an in-order wave that is issuing
// LDS-DMA pieces or fragment reads issues no MFMAs, and the two waves of a SIMD did so in lock-step):
//   * 12 waves: 8 consumers, 4 (M) x 2 (N), 64 x 64 = 4 x 4 accumulator fragments of v_mfma_f32_16x16x32_bf16 each, and 4
//     loaders that only issue global_load_lds_dwordx4 pieces (12 per wave and k-tile) and wait for them with counted vmcnt.
//     Three waves per SIMD: 168 registers per lane.
//   * ring of three 48 KB stages (A 32 KB + B 16 KB per 64-deep k-tile).  ONE raw s_barrier per k-tile, in the MIDDLE of the
//     consumers' step: before it a consumer has read BOTH 32-deep halves of stage g into registers (the second half's reads
//     go out row by row behind the MFMAs of the first), so at the barrier stage g is free and the loaders refill it with
//     k-tile g + 3 while k-tile g + 2 is still in flight: two k-tiles (96 KB per CU) on their way at any time — the ring of
//     the producer / consumer kernel (csrc/gemm_pc.hip) needs four stages for one less.
//   * persistent: a workgroup walks its tiles (XCD-contiguous chunks of the joint list) with the k-tile sequence running
//     across tile seams — while the consumers store a tile the loaders already fill the ring with the next tile's first
//     three k-tiles.  Epilogues from registers (gemm_device.hpp: bias, residual, activation + second output, dgrad x act',
//     fp32 or bf16 with 16-byte stores); bias gradients beside weight gradients from one MFMA against a ones fragment per
//     half step (n-tile 0).
//   * LDS images of gemm_device.hpp: k-contiguous [rows][64] with the 16-B-slot XOR (ds_read_b128), k-strided [64][128]
//     parts with the 32-B-group XOR (ds_read_b64_tr_b16) — for both images the fragments of a lane differ by CONSTANT
//     byte offsets between the two halves and between a fragment's two transposed reads, so a consumer keeps 2 (k-contiguous)
//     or 4 (k-strided) address registers per operand and everything else is an immediate.
// Shapes: M % 256 == 0, N % 128 == 0, K % 64 == 0 (the dispatcher cuts other shapes into a served part and a remainder).
#include <hip/hip_ext.h>
#include <string.h>
#include <mutex>
#include <type_traits>
#include <utility>
#include "gemm_skinny_device.hpp"
#include "../../include/swat_hip.h"

namespace swat {

constexpr int XL_BM = 256, XL_BN = 128, XL_NS = 3, XL_NC = 8, XL_NL = 4, XL_THREADS = 64 * (XL_NC + XL_NL);
constexpr int XL_A_BYTES = XL_BM * 128, XL_B_BYTES = XL_BN * 128, XL_STAGE = XL_A_BYTES + XL_B_BYTES;
constexpr int XL_PA = XL_A_BYTES / 1024, XL_PB = XL_B_BYTES / 1024, XL_LPS = (XL_PA + XL_PB) / XL_NL;   // 12 pieces per loader wave
constexpr int XL_MAXP = 8;

struct XlProb {
    GemmArgs g;
    int tiles_m, tiles_n, tile0;
    int tail;   // NT / NN: rows beyond g.M (<= 32) the launch's TAIL PHASE computes (csrc/gemm_skinny_device.hpp); 0 = none
};
struct XlArgs {
    XlProb p[XL_MAXP];
    int n, total;
    unsigned long long* trace;   // ablation build, exp bit 16
    int exp;                     // ablation build (swat_gemm_set_exp): 1 no DMA after the first ring fill, 2 no MFMA, 4 no epilogue, 256 no fragment reads
};
// the same with the product list in device memory: up to XL_MAXT products (all weight gradients of a tower in one launch)
constexpr int XL_MAXT = 64;
struct XlArgsTable {
    const XlProb* p;
    int n, total;
    unsigned long long* trace;
    int exp;
    XlAdam adam;   // on != 0: the tiles are consumed by AdamW in the epilogue instead of being stored as gradients
};

template <int IMM>
__device__ __forceinline__ bf16x8_t xl_lds_b128(unsigned addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
template <int IMM>
__device__ __forceinline__ bf16x8_t xl_lds_tr(unsigned addr) {   // the two transposed reads of a fragment: k-rows +0..3 and +4..7
    s16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(IMM));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(IMM + 1024));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void xl_pin(bf16x8_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void xl_bar() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N_>
__device__ __forceinline__ void xl_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_ > 15 ? 15 : N_) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N_>
__device__ __forceinline__ void xl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
template <int N_, typename F, int... Is>
__device__ __forceinline__ void xl_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N_, typename F>
__device__ __forceinline__ void xl_static_for(F&& f) { xl_static_for_impl<N_>(f, std::make_integer_sequence<int, N_>{}); }

// tile id -> product, tile row / column.  Tiles of a product walk M first (a run of consecutive ids shares one B panel); when the
// tile grid allows it, in blocks of 8 x 4 tiles so that the 32 tiles an XCD works on at a time share 8 A panels and 4 B panels.
struct XlTile {
    int pi, m0, n0, tn;
};
template <class Args>
__device__ __forceinline__ XlTile xl_decode(const Args& a, int id) {
    int pi = 0;
    if constexpr (std::is_same<Args, XlArgs>::value) {
#pragma unroll
        for (int i = 1; i < XL_MAXP; ++i)
            if (i < a.n && id >= a.p[i].tile0) pi = i;
    } else {
        // (tile0 ascends: binary search — a linear scan over 48 products costs each decode 48 dependent scalar loads)
        int lo = 0, hi = a.n;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (id >= a.p[mid].tile0) lo = mid;
            else hi = mid;
        }
        pi = lo;
    }
    const int t = id - a.p[pi].tile0, tmn = a.p[pi].tiles_m, tnn = a.p[pi].tiles_n;
    int tm, tn;
    if ((tmn & 7) == 0 && (tnn & 3) == 0) {
        const int sb = t >> 5, w = t & 31, sbm = tmn >> 3;
        tm = (sb % sbm) * 8 + (w & 7);
        tn = (sb / sbm) * 4 + (w >> 3);
    } else {
        tm = t % tmn;
        tn = t / tmn;
    }
    return XlTile{pi, tm * XL_BM, tn * XL_BN, tn};
}


// ---- epilogues of a consumer wave's 64 x 64 tile, one instantiation per combination of fused operands (the one generic
// epilogue of gemm_device.hpp holds registers for every combination at once: beside 64 accumulator registers and a 168-register
// budget it spilled, and a tile's epilogue took longer than its main loop).  Same arithmetic and rounding points as gemm_epilogue.
// Lane (r16 = lane & 15, g = lane >> 4) holds C[m = mbase + 16 i + r16][n = nbase + 16 j + 4 g .. + 3] in acc[i][j].
template <bool RES>
__device__ __forceinline__ void xl_epi_f32(const GemmArgs& p, f32x4_t (&acc)[4][4], int mbase, int nbase, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    const int n = nbase + 4 * g;
    float4 b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t o = (int64_t)(mbase + r16) * p.ldc + n;
    float* C = reinterpret_cast<float*>(p.C) + o;
    const float* R = p.residual + o;
    const int64_t rs = (int64_t)16 * p.ldc;
    float4 rc[4], rn[4];
    if constexpr (RES) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rc[j] = *reinterpret_cast<const float4*>(R + 16 * j);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (RES) {
            if (i + 1 < 4) {   // the next row's residual is requested before this row's stores (loads and stores share vmcnt)
#pragma unroll
                for (int j = 0; j < 4; ++j) rn[j] = *reinterpret_cast<const float4*>(R + (i + 1) * rs + 16 * j);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = make_float4(acc[i][j][0] * p.alpha + b[j].x, acc[i][j][1] * p.alpha + b[j].y, acc[i][j][2] * p.alpha + b[j].z,
                                   acc[i][j][3] * p.alpha + b[j].w);
            if constexpr (RES) { v.x += rc[j].x; v.y += rc[j].y; v.z += rc[j].z; v.w += rc[j].w; }
            *reinterpret_cast<float4*>(C + i * rs + 16 * j) = v;
        }
        if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rc[j] = rn[j];
        }
    }
}

template <bool ACTIN, bool SECOND>
__device__ __forceinline__ void xl_epi_bf16(const GemmArgs& p, f32x4_t (&acc)[4][4], int mbase, int nbase, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    // two accumulator tiles that are neighbours in n trade halves (v_permlane16_swap): lane (r16, g) then holds row r16 of tile
    // 2 q + (g & 1), columns 8 (g >> 1) .. + 7 — eight consecutive n, one 16-byte store per output
    const int n = nbase + 16 * (g & 1) + 8 * (g >> 1);   // + 32 q
    float4 b[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        b[q][0] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n + 32 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        b[q][1] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n + 32 * q + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    u32x4_t f[ACTIN ? 8 : 1];
    if constexpr (ACTIN) {   // every activation-input chunk of the tile up front: 32 registers, one latency
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                f[2 * i + q] = *reinterpret_cast<const u32x4_t*>(p.actin + (int64_t)(mbase + 16 * i + r16) * p.ldact + n + 32 * q);
    }
    const int64_t o0 = (int64_t)(mbase + r16) * p.ldc + n;
    const int64_t rs = (int64_t)16 * p.ldc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo = acc[i][2 * q][e], hi = acc[i][2 * q + 1][e];
                asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                v[e] = lo * p.alpha;
                v[4 + e] = hi * p.alpha;
            }
            v[0] += b[q][0].x; v[1] += b[q][0].y; v[2] += b[q][0].z; v[3] += b[q][0].w;
            v[4] += b[q][1].x; v[5] += b[q][1].y; v[6] += b[q][1].z; v[7] += b[q][1].w;
            if constexpr (ACTIN) {   // dgrad through the activation: dF = dG * act'(F)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const unsigned w = f[2 * i + q][h];
                    v[2 * h] *= act_bwd(bf2f((unsigned short)(w & 0xffffu)), p.act);
                    v[2 * h + 1] *= act_bwd(bf2f((unsigned short)(w >> 16)), p.act);
                }
            }
            const int64_t o = o0 + i * rs + 32 * q;
            u32x4_t pk;
#pragma unroll
            for (int h = 0; h < 4; ++h) pk[h] = pack_bf16x2(v[2 * h], v[2 * h + 1]);
            if (!p.c2_only) *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned short*>(p.C) + o) = pk;
            if constexpr (SECOND) {   // second output: activation of the ROUNDED pre-activation (what backward re-reads)
                u32x4_t pk2;
#pragma unroll
                for (int h = 0; h < 4; ++h)
                    pk2[h] = pack_bf16x2(act_fwd(__uint_as_float(pk[h] << 16), p.act), act_fwd(__uint_as_float(pk[h] & 0xffff0000u), p.act));
                *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned short*>(p.C2) + o) = pk2;
            }
        }
}

__device__ __forceinline__ void xl_epilogue(const GemmArgs& p, f32x4_t (&acc)[4][4], int mbase, int nbase, int lane) {
    if (p.out_f32) {
        if (p.residual) xl_epi_f32<true>(p, acc, mbase, nbase, lane);
        else xl_epi_f32<false>(p, acc, mbase, nbase, lane);
    } else if (p.actin) xl_epi_bf16<true, false>(p, acc, mbase, nbase, lane);
    else if (p.C2) xl_epi_bf16<false, true>(p, acc, mbase, nbase, lane);
    else xl_epi_bf16<false, false>(p, acc, mbase, nbase, lane);
}

template <int LAYOUT, bool ABL, class Args>
__device__ __forceinline__ void xl_main(const Args& a, unsigned char* smem) {
    constexpr bool A_KC = (LAYOUT != L_TN);
    constexpr bool B_KC = (LAYOUT == L_NT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto stamp = [&](int i) {
        if constexpr (ABL) {
            if ((a.exp & 16) && a.trace && tid == 0) a.trace[(size_t)blockIdx.x * 8 + i] = wall_clock64();
        }
    };
    stamp(0);
    // this workgroup's tiles: XCD x (blocks x, x + 8, ..: speed only) owns a contiguous chunk of the joint list, its nw workgroups
    // take the chunk's tiles round-robin
    int t_first, t_end, t_step;
    {
        const int lb = blockIdx.x, G = gridDim.x, x = lb & 7, s = lb >> 3;
        const int q = a.total >> 3, r = a.total & 7;
        const int c0 = x * q + (x < r ? x : r), c1 = c0 + q + (x < r ? 1 : 0);
        t_step = (G - x + 7) >> 3;
        t_first = c0 + s;
        t_end = c1;
    }

    if (wave >= XL_NC) {
        // ================================ loader waves ================================
        const int lw = wave - XL_NC;
        // piece q = lw + 4 i of a stage: q < 32 -> A image, else B image; its per-lane position inside the k-tile
        int pk[XL_LPS], pc[XL_LPS];   // k-contiguous: (row, 16-B slot); k-strided: (k-row, column)
#pragma unroll
        for (int i = 0; i < XL_LPS; ++i) {
            const int q = lw + XL_NL * i;
            const bool isA = q < XL_PA;
            const int byte = (isA ? q : q - XL_PA) * 1024 + lane * 16;
            if ((isA && A_KC) || (!isA && B_KC)) {
                const int row = byte >> 7;
                pk[i] = row;
                pc[i] = 8 * (((byte >> 4) & 7) ^ ((row >> 1) & 7));
            } else if (isA) {
                stp_inv<XL_BM>(byte, pk[i], pc[i]);
            } else {
                stp_inv<XL_BN>(byte, pk[i], pc[i]);
            }
        }
        unsigned off[XL_LPS];
        const char *gA = nullptr, *gB = nullptr;
        int64_t stepA = 0, stepB = 0;
        int cur_p = -1, nk = 0;
        int it = t_first, ikt = 0;   // the next k-tile to issue: tile `it`, k-tile `ikt`
        auto open_tile = [&]() {     // uniform: operand bases of tile `it`
            const XlTile T = xl_decode(a, it);
            const GemmArgs& g = a.p[T.pi].g;
            if (T.pi != cur_p) {
                cur_p = T.pi;
#pragma unroll
                for (int i = 0; i < XL_LPS; ++i) {
                    const bool isA = lw + XL_NL * i < XL_PA;
                    off[i] = (unsigned)(pk[i] * (isA ? (A_KC ? g.lda : g.lda) : g.ldb) + pc[i]) * 2u;
                }
                stepA = A_KC ? 128 : (int64_t)128 * g.lda;
                stepB = B_KC ? 128 : (int64_t)128 * g.ldb;
                nk = g.K >> 6;
            }
            gA = reinterpret_cast<const char*>(g.A + (A_KC ? (int64_t)T.m0 * g.lda : (int64_t)T.m0));
            gB = reinterpret_cast<const char*>(g.B + (B_KC ? (int64_t)T.n0 * g.ldb : (int64_t)T.n0));
        };
        auto issue = [&](int slot) {   // k-tile (it, ikt) into ring slot `slot`; advances the cursor
            unsigned char* dst = smem + slot * XL_STAGE + lw * 1024;
#pragma unroll
            for (int i = 0; i < XL_LPS; ++i) {
                const bool isA = lw + XL_NL * i < XL_PA;   // i < 8
                const char* src = (isA ? gA + ikt * stepA : gB + ikt * stepB) + off[i];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + i * (XL_NL * 1024)), 16, 0, 0);
            }
            if (++ikt == nk) {
                ikt = 0;
                it += t_step;
                if (it < t_end) open_tile();
            }
        };
        // total k-tiles of this workgroup (uniform)
        int total_g = 0;
        for (int t = t_first; t < t_end; t += t_step) total_g += a.p[xl_decode(a, t).pi].g.K >> 6;
        if (total_g == 0) return;
        open_tile();
        int issued = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (issued < total_g) { issue(s); ++issued; }
        // k-tile 0 landed (this wave's pieces) before the opening barrier; the third stage goes out behind it, so the consumers do
        // not wait for its 12 issue slots per loader wave
        if (issued == 2) xl_wait_vm<XL_LPS>();
        else xl_wait_vm<0>();
        stamp(1);
        xl_bar();
        if (issued < total_g) { issue(2); ++issued; }
        int slot = 0;
        for (int g = 0; g < total_g; ++g) {
            // k-tile g + 1 landed; g + 2 (the youngest issued so far) may still fly
            if (g + 2 < issued) xl_wait_vm<XL_LPS>();
            else xl_wait_vm<0>();
            xl_bar();   // barrier g: the consumers hold all of k-tile g in registers -> its slot is free
            if (issued < total_g && !(ABL && (a.exp & 1))) { issue(slot); ++issued; }
            else if (ABL && issued < total_g) ++issued;
            slot = slot + 1 == XL_NS ? 0 : slot + 1;
        }
        return;
    }

    // ================================ consumer waves ================================
    const int wr = wave >> 1, wc = wave & 1;
    const int r16 = lane & 15, g4 = lane >> 4;
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
    // fragment addresses inside stage 0, half 0.  k-contiguous: fragment i at + 2048 i, half 1 in ra[1]; k-strided: fragment i in
    // ra[i], half 1 at + 8192, second transposed read at + 1024 (gemm_device.hpp stp_off: f(k) is the same for k, k + 4 and k + 32)
    unsigned ra[4], rb[4];
    if constexpr (A_KC) {
        ra[0] = lds0 + kc_off(64 * wr + r16, g4);
        ra[1] = lds0 + kc_off(64 * wr + r16, 4 + g4);
        ra[2] = ra[3] = 0;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = lds0 + stp_off<XL_BM>(8 * g4 + (r16 >> 2), 64 * wr + 16 * i) + 8 * (r16 & 3);
    }
    if constexpr (B_KC) {
        rb[0] = lds0 + XL_A_BYTES + kc_off(64 * wc + r16, g4);
        rb[1] = lds0 + XL_A_BYTES + kc_off(64 * wc + r16, 4 + g4);
        rb[2] = rb[3] = 0;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = lds0 + XL_A_BYTES + stp_off<XL_BN>(8 * g4 + (r16 >> 2), 64 * wc + 16 * i) + 8 * (r16 & 3);
    }
    auto read_b = [&](auto J, auto KS, unsigned so, bf16x8_t (&f)[4]) {
        constexpr int j = decltype(J)::value, ks = decltype(KS)::value;
        if constexpr (ABL) {
            if (a.exp & 256) return;
        }
        if constexpr (B_KC) f[j] = xl_lds_b128<2048 * j>(rb[ks] + so);
        else f[j] = xl_lds_tr<8192 * ks>(rb[j] + so);
    };
    constexpr int NA = A_KC ? 1 : 2, NB = B_KC ? 1 : 2;   // LDS instructions per fragment

    constexpr bool CS = LAYOUT == L_TN;   // bias gradient beside a weight gradient: column sums of A
    f32x4_t acc[4][4], accs[CS ? 4 : 1];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (CS) accs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
    };
    zero_acc();
    const s16x8_t ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    // Fragment registers: B fragments in two sets (one per 32-deep half: every accumulator row uses all four); A fragments in ONE
    // set refreshed in place — the read of row i's fragment for the next half goes out right behind row i's MFMAs — plus one spare
    // (fx): the second half takes row 3 from fx, so that row 3's next fragment can be fetched BEFORE row 3's MFMAs in both halves and
    // nothing is left in flight when the barrier wants the whole stage in registers.
    bf16x8_t fa[4], fx, fb[2][4];
    if constexpr (ABL) {
        const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_bits);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = fb[0][i] = fb[1][i] = ones;
        fx = ones;
    }
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    bool do_cs = false;
    auto read_a1 = [&](auto I, auto KS, unsigned so, bf16x8_t& f) {   // one A fragment into a given register
        constexpr int i = decltype(I)::value, ks = decltype(KS)::value;
        if constexpr (ABL) {
            if (a.exp & 256) return;
        }
        if constexpr (A_KC) f = xl_lds_b128<2048 * i>(ra[ks] + so);
        else f = xl_lds_tr<8192 * ks>(ra[i] + so);
    };
    // the MFMAs of half H (0: A from fa[0..3], B from fb[0]; 1: A from fa[0..2] + fx, B from fb[1]), row by row; behind rows 0-2
    // the reads of the NEXT half (stage offset sn, half NK) go out: A_0 B_0 B_1 | A_1 B_2 B_3 | A_2 A_3
    auto half = [&](auto H, auto NK, auto WITH_READS, unsigned sn) {
        constexpr int h = decltype(H)::value, o = 1 - h;
        constexpr bool wr_ = decltype(WITH_READS)::value;
        using NKt = decltype(NK);
        xl_static_for<4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (i == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xl_pin(fb[h][j]);
            }
            bf16x8_t& fr = (h == 1 && i == 3) ? fx : fa[i];
            xl_pin(fr);
            bool skip = false;
            if constexpr (ABL) skip = (a.exp & 2) != 0;
            if (!skip) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[h][j], fr, acc[i][j], 0, 0, 0);
                if constexpr (CS) {
                    if (do_cs) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones_bits), fr, accs[i], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (wr_) {
                if constexpr (i == 0) {
                    read_a1(I0{}, NKt{}, sn, fa[0]);
                    read_b(I0{}, NKt{}, sn, fb[o]);
                    read_b(I1{}, NKt{}, sn, fb[o]);
                } else if constexpr (i == 1) {
                    read_a1(I1{}, NKt{}, sn, fa[1]);
                    read_b(I2{}, NKt{}, sn, fb[o]);
                    read_b(I3{}, NKt{}, sn, fb[o]);
                } else if constexpr (i == 2) {
                    read_a1(I2{}, NKt{}, sn, fa[2]);
                    if constexpr (h == 0) read_a1(I3{}, NKt{}, sn, fx);
                    else read_a1(I3{}, NKt{}, sn, fa[3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    auto first_reads = [&](unsigned so_) {   // half 0 of the stage at so_ into (fa, fb[0]): at a tile's start
        read_a1(I0{}, K0{}, so_, fa[0]); read_a1(I1{}, K0{}, so_, fa[1]); read_a1(I2{}, K0{}, so_, fa[2]); read_a1(I3{}, K0{}, so_, fa[3]);
        xl_static_for<4>([&](auto J) { read_b(J, K0{}, so_, fb[0]); });
    };

    if (t_first >= t_end) return;
    // AdamW scalars of the fused epilogue (table form): this step's learning rate and bias corrections, the give-up guard
    float ad_lr = 0.f, ad_bc1 = 1.f, ad_sbc2 = 1.f;
    bool adam_live = false;
    if constexpr (std::is_same<Args, XlArgsTable>::value) {
        if (a.adam.on) {
            ad_lr = a.adam.lr[0];
            ad_bc1 = a.adam.bc[0];
            ad_sbc2 = sqrtf(a.adam.bc[1]);
            adam_live = !(a.adam.guard && __builtin_nontemporal_load(a.adam.guard) != 0u);
            // Stagger.  Every workgroup alternates a main loop (matrix cores, ~34 us) and an AdamW epilogue (852 KB of HBM traffic);
            // launched together they stay in lock-step — all epilogues at once, HBM-bound, then all main loops with the HBM idle.
            // Four phase groups start a quarter of a cycle apart, so that at any time about a quarter of the chip streams
            // parameters while the rest multiplies.
            const unsigned long long t0 = wall_clock64(), dl = (unsigned long long)((blockIdx.x >> 3) & 3) * (unsigned)a.adam.stagger;
            while (wall_clock64() - t0 < dl) __builtin_amdgcn_s_sleep(16);
        }
    }
    stamp(1);
    xl_bar();   // opening barrier: k-tile 0 of the first tile has landed
    stamp(2);
    unsigned so = 0;
    for (int tile = t_first; tile < t_end; tile += t_step) {
        const XlTile T = xl_decode(a, tile);
        const GemmArgs& G = a.p[T.pi].g;
        const int nk = G.K >> 6;
        do_cs = CS && G.colsum != nullptr && T.tn == 0 && wc == 0;
        first_reads(so);   // (the stage has landed: opening barrier, or the last barrier of the previous tile)
        // (the last k-tile is peeled: a branch around the second half inside the loop makes every accumulator a phi and the
        // register allocator answers with copies and spills)
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const unsigned sn = so + XL_STAGE == XL_NS * XL_STAGE ? 0u : so + XL_STAGE;
            xl_wait_lgkm<0>();                        // half 0 of this k-tile
            half(K0{}, K1{}, std::true_type{}, so);   // MFMAs of half 0; reads of half 1 of the same stage
            xl_wait_lgkm<0>();                        // the whole stage is in registers
            xl_bar();                                 // barrier g: the loaders refill this stage; the next k-tile has landed
            half(K1{}, K0{}, std::true_type{}, sn);   // MFMAs of half 1; reads of the next k-tile's half 0
            so = sn;
        }
        xl_wait_lgkm<0>();
        half(K0{}, K1{}, std::true_type{}, so);
        xl_wait_lgkm<0>();
        xl_bar();
        half(K1{}, K0{}, std::false_type{}, 0u);      // last k-tile: the fragment registers are the epilogue's from here
        so = so + XL_STAGE == XL_NS * XL_STAGE ? 0u : so + XL_STAGE;
        stamp(3);
        bool skip_epi = false;
        if constexpr (ABL) skip_epi = (a.exp & 4) != 0;
        if (!skip_epi) {
            // the lane index goes through an opaque asm here: everything the epilogue derives from it (row / column offsets of 16
            // fragments, 64-bit) would otherwise be hoisted out of the tile loop and live — spilled — across the main loop
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            if constexpr (LAYOUT == L_TN) {
                // weight gradient: plain fp32 tile (gemm_xl_serves admits no fused epilogue here) — 16-byte stores, no operand loads
                float* C = reinterpret_cast<float*>(G.C) + (int64_t)(T.m0 + 64 * wr + (lane_e & 15)) * G.ldc + T.n0 + 64 * wc + 4 * (lane_e >> 4);
                bool adam = false;
                if constexpr (std::is_same<Args, XlArgsTable>::value) adam = a.adam.on != 0;
                if (adam) {
                    if constexpr (std::is_same<Args, XlArgsTable>::value) {
                        if (adam_live) {
                            const int64_t off = C - a.adam.gbase;
                            typedef float nt4 __attribute__((ext_vector_type(4)));
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int64_t ro = off + (int64_t)16 * i * G.ldc;
                                nt4 pn[4], mn[4], vn[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) {   // a row of fragments: 12 loads in flight
                                    pn[j] = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(a.adam.p + ro + 16 * j));
                                    mn[j] = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(a.adam.m + ro + 16 * j));
                                    vn[j] = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(a.adam.v + ro + 16 * j));
                                }
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    float pa[4], ma[4], va[4];
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)   // (as adamw_kernel: every operation rounded — the forms agree bit for bit whatever the optimiser fuses elsewhere)
                                        const float gg = acc[i][j][k] * a.adam.gscale;
                                        pa[k] = pn[j][k] * (1.0f - ad_lr * a.adam.wd);
                                        ma[k] = a.adam.b1 * mn[j][k] + (1.0f - a.adam.b1) * gg;
                                        va[k] = a.adam.b2 * vn[j][k] + (1.0f - a.adam.b2) * gg * gg;
                                        const float denom = sqrtf(va[k]) / ad_sbc2 + a.adam.eps;
                                        pa[k] -= (ad_lr / ad_bc1) * (ma[k] / denom);
                                    }
                                    __builtin_nontemporal_store((nt4){pa[0], pa[1], pa[2], pa[3]}, reinterpret_cast<nt4*>(a.adam.p + ro + 16 * j));
                                    __builtin_nontemporal_store((nt4){ma[0], ma[1], ma[2], ma[3]}, reinterpret_cast<nt4*>(a.adam.m + ro + 16 * j));
                                    __builtin_nontemporal_store((nt4){va[0], va[1], va[2], va[3]}, reinterpret_cast<nt4*>(a.adam.v + ro + 16 * j));
                                    *reinterpret_cast<ushort4*>(a.adam.shadow + ro + 16 * j) = make_ushort4(f2bf(pa[0]), f2bf(pa[1]), f2bf(pa[2]), f2bf(pa[3]));
                                }
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *reinterpret_cast<f32x4_t*>(C + (int64_t)16 * i * G.ldc + 16 * j) = acc[i][j];
                }
            } else {
                xl_epilogue(G, acc, T.m0 + 64 * wr, T.n0 + 64 * wc, lane_e);
            }
            if constexpr (CS) {
                if (do_cs && (lane_e >> 4) == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) G.colsum[T.m0 + 64 * wr + 16 * i + (lane_e & 15)] = accs[i][0];
                }
            }
        }
        zero_acc();
    }
    if constexpr (ABL) {
        stamp(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(5);
    }
}

// The launch: the persistent tile loop, then (NT / NN) the TAIL PHASE — the up to 32 rows a product has beyond its last 256-row tile,
// spread over ALL workgroups of the launch: 16 output columns per workgroup and turn, the eight consumer waves an eighth of the reduction
// each, partial sums added in wave order in LDS (the remainder-row kernel's scheme, csrc/gemm_skinny.hip — as a launch of its own it
// cost 8-13 us per product, 192 times per ViT-L/14 step, almost all of it launch latency).
#ifndef XL_LAUNCH_BOUND
#define XL_LAUNCH_BOUND XL_THREADS
#endif
template <int LAYOUT, bool ABL, class Args = XlArgs>
__global__ __launch_bounds__(XL_LAUNCH_BOUND) void gemm_xl_kernel(const Args a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[XL_NS * XL_STAGE];
    xl_main<LAYOUT, ABL, Args>(a, smem);
    if constexpr (LAYOUT != L_TN && std::is_same<Args, XlArgs>::value) {
        bool any = false;
#pragma unroll
        for (int i = 0; i < XL_MAXP; ++i) any = any || (i < a.n && a.p[i].tail > 0);
        if (!any) return;
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();   // every wave is out of the tile loop: the ring is free
        float* part = reinterpret_cast<float*>(smem + 16384);   // [2 turns][8 waves][2 fragments][64 lanes][4]; bounce buffers: smem[0 .. 16 K)
        for (int pi = 0; pi < a.n; ++pi) {
            const int tr = a.p[pi].tail;
            if (tr <= 0) continue;
            GemmArgs gt = a.p[pi].g;
            {
                const int64_t mm = gt.M, co = mm * gt.ldc;
                gt.A = gt.A + mm * gt.lda;
                if (gt.C) gt.C = gt.out_f32 ? (void*)((float*)gt.C + co) : (void*)((unsigned short*)gt.C + co);
                if (gt.C2) gt.C2 = (unsigned short*)gt.C2 + co;
                if (gt.residual) gt.residual = gt.residual + co;
                if (gt.actin) gt.actin = gt.actin + mm * gt.ldact;
                gt.M = tr;
            }
            const int ngroups = (gt.N + 15) >> 4;
            int turn = 0;
            for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
                f32x4_t acc[2][1];
                float* mypart = part + turn * (SK_WAVES * 2 * 64 * 4);
                if (wave < SK_WAVES) {
                    skinny_partial<LAYOUT, 2>(gt, 16 * grp, wave, lane, smem + wave * 2048, acc);
#pragma unroll
                    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4_t*>(mypart + ((wave * 2 + i) * 64 + lane) * 4) = acc[i][0];
                }
                __syncthreads();   // (two turns of partial buffers: the next turn's writes cannot overtake this turn's reads)
                if (wave == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x4_t sum = *reinterpret_cast<const f32x4_t*>(mypart + (i * 64 + lane) * 4);
#pragma unroll
                        for (int w = 1; w < SK_WAVES; ++w) {
                            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(mypart + ((w * 2 + i) * 64 + lane) * 4);
                            sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
                        }
                        acc[i][0] = sum;
                    }
                    gemm_epilogue<2, 1>(gt, acc, 0, 16 * grp, lane, 0, 0);
                }
                turn ^= 1;
            }
        }
    }
}

// Can the 256 x 128 kernel take this product as it stands?  (whole tiles, 16-byte rows, plain batch-1 product)
bool gemm_xl_serves(int layout, const GemmArgs& g) {
    if (g.inner > 1 || g.M % XL_BM || g.N % XL_BN || g.K % 64 || g.K < 64 || g.lda % 8 || g.ldb % 8) return false;
    if (g.out_f32 ? (g.ldc % 4 != 0) : (g.ldc % 8 != 0 || (g.actin && g.ldact % 8 != 0))) return false;
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B) | reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.C2) |
         reinterpret_cast<uintptr_t>(g.actin)) & 15)
        return false;
    if (g.colsum && layout != L_TN) return false;
    if (g.rpg || g.rowbias || (g.actin && (g.C2 || g.out_f32)) || (g.C2 && g.out_f32) || (g.residual && !g.out_f32)) return false;   // combinations the step never asks for
    if (layout == L_TN && (!g.out_f32 || g.bias || g.residual || g.rowbias || g.actin || g.C2 || g.rpg || g.alpha != 1.0f)) return false;
    // 32-bit per-lane piece offsets: (row inside the tile) * ld
    const int64_t spanA = layout == L_TN ? (int64_t)64 * g.lda : (int64_t)XL_BM * g.lda;
    const int64_t spanB = layout == L_NT ? (int64_t)XL_BN * g.ldb : (int64_t)64 * g.ldb;
    return spanA * 2 < (1ll << 31) && spanB * 2 < (1ll << 31);
}

// n <= 8 products of one layout over one joint tile list.  -> 0 launched, 1 not served, < 0 error
int launch_gemm_xl(int layout, const GemmArgs* probs, int n, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, int exp, unsigned long long* trace,
                   const int* tails) {
    if (n < 1 || n > XL_MAXP) return 1;
    XlArgs a{};
    a.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        if (!gemm_xl_serves(layout, probs[i])) return 1;
        a.p[i].g = probs[i];
        a.p[i].tiles_m = probs[i].M / XL_BM;
        a.p[i].tiles_n = probs[i].N / XL_BN;
        a.p[i].tile0 = total;
        a.p[i].tail = (tails && layout != L_TN) ? tails[i] : 0;
        if (a.p[i].tail < 0 || a.p[i].tail > 32) return 1;
        total += a.p[i].tiles_m * a.p[i].tiles_n;
    }
    if (total == 0) return 1;
    a.total = total;
    a.exp = exp & (1 | 2 | 4 | 16 | 256);
    a.trace = trace;
    const dim3 grid(total < 256 ? total : 256), block(XL_THREADS);
#define XL_LAUNCH(L_)                                                                                                  \
    do {                                                                                                               \
        if (a.exp) hipLaunchKernelGGL((gemm_xl_kernel<L_, true>), grid, block, 0, s, a);                               \
        else if (ev0) hipExtLaunchKernelGGL((gemm_xl_kernel<L_, false>), grid, block, 0, s, ev0, ev1, 0, a);           \
        else hipLaunchKernelGGL((gemm_xl_kernel<L_, false>), grid, block, 0, s, a);                                    \
    } while (0)
    if (layout == L_NT) XL_LAUNCH(L_NT);
    else if (layout == L_NN) XL_LAUNCH(L_NN);
    else XL_LAUNCH(L_TN);
#undef XL_LAUNCH
    return check_launch("gemm_xl_kernel");
}

// up to 64 weight gradients (TN) over one joint tile list, the product list in device memory.  The table lives in a per-device
// buffer and is re-sent only when it changes (a training step asks for the same pointers every time).  -> 0 / 1 not served / < 0
int launch_gemm_xl_table(const GemmArgs* probs, int n, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const XlAdam* adam) {
    if (n < 1 || n > XL_MAXT) return 1;
    // a few tables per device (a step asks for the same ones every time: all blocks at once, or groups of blocks): a table is sent
    // once, on the stream of its first launch; a launch on another stream than the one that sent it is not ordered with the copy
    constexpr int SLOTS = 8;
    struct Slot {
        XlProb* dev = nullptr;
        XlProb* pinned = nullptr;
        XlProb last[XL_MAXT];
        int n = 0;
        hipStream_t stream = nullptr;
        unsigned long long used = 0;
    };
    struct Cache {
        Slot slot[SLOTS];
        unsigned long long tick = 0;
    };
    static Cache cache[16];
    static std::mutex cache_mutex;   // (host threads driving different streams of one device share the slots)
    std::lock_guard<std::mutex> lock(cache_mutex);
    int devi = 0;
    if (hipGetDevice(&devi) != hipSuccess || devi < 0 || devi >= 16) return 1;
    Cache& c = cache[devi];
    XlProb t[XL_MAXT];
    memset(t, 0, sizeof(t));
    int total = 0;
    for (int i = 0; i < n; ++i) {
        if (!gemm_xl_serves(L_TN, probs[i])) return 1;
        t[i].g = probs[i];
        t[i].tiles_m = probs[i].M / XL_BM;
        t[i].tiles_n = probs[i].N / XL_BN;
        t[i].tile0 = total;
        total += t[i].tiles_m * t[i].tiles_n;
    }
    if (total == 0) return 1;
    Slot* sl = nullptr;
    for (int k = 0; k < SLOTS; ++k)
        if (c.slot[k].n == n && memcmp(c.slot[k].last, t, sizeof(XlProb) * n) == 0) sl = &c.slot[k];
    if (sl && sl->stream != s) return 1;   // (sent on another stream: the caller takes the per-block launches)
    if (!sl) {
        sl = &c.slot[0];
        for (int k = 1; k < SLOTS; ++k)
            if (c.slot[k].used < sl->used) sl = &c.slot[k];   // least recently used (never used: 0)
        if (!sl->dev) {
            if (hipMalloc((void**)&sl->dev, sizeof(XlProb) * XL_MAXT) != hipSuccess ||
                hipHostMalloc((void**)&sl->pinned, sizeof(XlProb) * XL_MAXT, hipHostMallocDefault) != hipSuccess) {
                sl->dev = nullptr;
                return -1;
            }
        } else if (sl->stream) {
            // the slot's old table may still be read by launches in flight, and its staging buffer by a copy: let them finish
            SWAT_HIP(hipStreamSynchronize(sl->stream));
        }
        memcpy(sl->pinned, t, sizeof(XlProb) * n);
        SWAT_HIP(hipMemcpyAsync(sl->dev, sl->pinned, sizeof(XlProb) * n, hipMemcpyHostToDevice, s));
        memcpy(sl->last, t, sizeof(XlProb) * n);
        sl->n = n;
        sl->stream = s;
    }
    sl->used = ++c.tick;
    XlArgsTable a{};
    a.p = sl->dev; a.n = n; a.total = total; a.trace = nullptr; a.exp = 0;
    if (adam) a.adam = *adam;
    const dim3 grid(total < 256 ? total : 256), block(XL_THREADS);
    if (ev0) hipExtLaunchKernelGGL((gemm_xl_kernel<L_TN, false, XlArgsTable>), grid, block, 0, s, ev0, ev1, 0, a);
    else hipLaunchKernelGGL((gemm_xl_kernel<L_TN, false, XlArgsTable>), grid, block, 0, s, a);
    return check_launch("gemm_xl_kernel (table)");
}

}  // namespace swat
