// swat_amd/csrc/retrieval_bf16.hip — dense (N_query x N_db) candidate scoring at the HBM rate (path B, BASELINE config 4).
//
// ref: the per-class GEMV + sort of retrieval/sample_retrieval.py:724-825 generalised to a query batch.  The exact fp32
// scan on the f32 matrix cores (retrieval_mfma.hip) is bound by the 157 TFLOP/s f32 MFMA rate: 256 GFLOP per pass of 200
// queries over 1.25 M rows = 1.6 ms at peak, 3.7 ms measured, against 0.32 ms for reading the 2.56 GB shard once.  Here the
// pass only SELECTS candidates: database rows stream from HBM once as fp32, are rounded to bf16 in registers and
// multiplied with bf16 queries on v_mfma_f32_16x16x32_bf16 (16x the f32 rate); every (query, row) whose approximate
// score reaches the query's threshold is appended to its candidate list.  The caller re-scores the few hundred
// candidates per query exactly (k-ordered fp32 fma chain, swat_sim_gathered) and proves completeness from the error
// bound |approx - exact| <= (2^-8 + 2^-12) |q| |d| (two bf16 roundings per product + fp32 accumulation), for which this
// kernel also returns max |d|^2 over the rows it saw — so the final top-k indices stay bit-exact.
//
// Structure: 4 waves per workgroup, each wave owns 32 database rows of a 128-row block (two 16-row MFMA fragments)
// against ALL query tiles (NQT x 16 queries: NQT x 2 accumulators of 16x16).  Database fragments never touch LDS: lane
// (r16, g) loads 32 B = k 8g..8g+7 of row r16 — 16 rows x 128 B full lines per instruction pair — one 64-deep chunk
// ahead of the MFMAs.  Query chunks [16 NQT][64] bf16 (L2-resident, 213 KB per 128-row block) go global -> LDS by
// LDS-DMA into two buffers with the 16-B-slot XOR of the GEMM's k-contiguous image (gemm_device.hpp: kc_off).
// Workgroups walk blocks b, b + grid, ...; the chunk stream is continuous across blocks.
//
// Measured (MI355X, 200 queries x 1.25 M rows, tools/perf_dense.py): 0.80 ms = 3.2 TB/s for this kernel (0.40 of the 8 TB/s
// peak; 4.8 TB/s with 32 queries), 1.05 ms for the whole exact top-k (re-scoring, two sorts, proof: 0.21 ms) against
// 3.6 ms for the f32-matrix-core scan.  What bounds it: one 64-deep chunk per wave in flight (12 waves x 4 KB per CU) against
// the HBM latency; deeper register rings do not compose with the per-step LDS-DMA of the query chunk (vmcnt completes in
// order, so waiting for the newest query piece drains every older database load).  Tried and dropped: two fragments per
// wave (128-row blocks, 2 workgroups per CU: 1.00 ms), a fifth wave touching the database 8 / 16 steps ahead with LDS-DMA
// sector loads (1.27 ms: the extra wave costs a workgroup of residency and the touches did not shorten the loads).
#include <stdlib.h>
#include "common.hpp"
#include "../../include/swat_hip.h"

namespace swat {

typedef __attribute__((ext_vector_type(8))) __bf16 rb_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float rb_f32x4_t;
typedef __attribute__((ext_vector_type(8))) short rb_s16x8_t;

__device__ __forceinline__ int rb_kc_off(int row, int t) { return row * 128 + ((t ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ uint32_t rb_ordered_bits(float s) {
    if (s == 0.0f) s = 0.0f;
    uint32_t u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int NQT, int NF>
__global__ __launch_bounds__(256, NF == 1 ? 3 : 2) void sim_candidates_bf16_kernel(const unsigned short* __restrict__ qb, int n_q, int q0,
                                                                  const float* __restrict__ db, int64_t n_db, int d,
                                                                  const float* __restrict__ thr, int64_t row_base,
                                                                  unsigned long long* __restrict__ cand, int cap,
                                                                  unsigned* __restrict__ cnt, unsigned* __restrict__ max_norm2_bits) {
    constexpr int QROWS = 16 * NQT;
    constexpr int QBYTES = QROWS * 128;      // one 64-deep query chunk
    constexpr int QP = QROWS / 8;            // 1 KiB pieces per chunk
    constexpr int LPS = (QP + 3) / 4;        // per wave
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * QBYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int nchunk = d >> 6;
    constexpr int BR = 64 * NF;  // database rows per block: NF fragments of 16 rows per wave
    const int64_t nblocks = (n_db + BR - 1) / BR;
    if ((int64_t)blockIdx.x >= nblocks) return;
    const int64_t my_blocks = (nblocks - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t steps = my_blocks * nchunk;

    // query pieces of this wave: piece p covers chunk rows 8p .. 8p+7; lane -> row 8p + lane / 8, LDS slot lane % 8 holds the
    // 16-B chunk (slot ^ f(row)) of that row
    unsigned qoff[LPS];
    int qdst[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        int p = wave + 4 * i;
        if (p >= QP) p -= 4;  // QP % 4 != 0: repeat the previous piece (same bytes to the same place)
        const int row = 8 * p + (lane >> 3);
        const int t = (lane & 7) ^ ((row >> 1) & 7);
        int rq = q0 + row; rq = rq < n_q ? rq : n_q - 1;
        qoff[i] = (unsigned)(rq * d + 8 * t) * 2u;
        qdst[i] = p * 1024;
    }
    const char* qbase = reinterpret_cast<const char*>(qb);
    auto issue_q = [&](int buf, int c) {
#pragma unroll
        for (int i = 0; i < LPS; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qbase + qoff[i] + c * 128),
                                             (__attribute__((address_space(3))) void*)(smem + buf * QBYTES + qdst[i]), 16, 0, 0);
    };
    // database fragments: frag f, half ks -> lane reads k = 64 c + 32 ks + 8 g .. + 7 of row blk * 128 + wave * 32 + 16 f + r16
    auto db_ptr = [&](int64_t blk, int f) -> const float* {
        int64_t row = blk * BR + wave * (16 * NF) + 16 * f + r16;
        row = row < n_db ? row : n_db - 1;
        return db + row * (int64_t)d + 8 * g;
    };
    float4 nx[NF][2][2];  // [f][ks][lo/hi 16 B]
    auto load_db = [&](int64_t blk, int c) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float* p = db_ptr(blk, f) + 64 * c;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                nx[f][ks][0] = *reinterpret_cast<const float4*>(p + 32 * ks);
                nx[f][ks][1] = *reinterpret_cast<const float4*>(p + 32 * ks + 4);
            }
        }
    };
    // fragment read offsets of the query tiles (k-contiguous image, one ds_read_b128 per tile and half step)
    unsigned qr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qr[ks] = rb_kc_off(r16, 4 * ks + g);
    rb_f32x4_t acc[NF][NQT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NQT; ++t) acc[f][t] = (rb_f32x4_t){0.f, 0.f, 0.f, 0.f};
    float ss[NF], norm2_max = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) ss[f] = 0.f;

    int64_t blk = blockIdx.x;
    int c = 0;
    issue_q(0, 0);
    load_db(blk, 0);
    for (int64_t s = 0; s < steps; ++s) {
        // chunk c of block blk: its query chunk (LDS-DMA) and its database registers were requested one step ago
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        rb_bf16x8_t dfr[NF][2];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float4 a = nx[f][ks][0], b = nx[f][ks][1];
                ss[f] = fmaf(a.x, a.x, ss[f]); ss[f] = fmaf(a.y, a.y, ss[f]); ss[f] = fmaf(a.z, a.z, ss[f]); ss[f] = fmaf(a.w, a.w, ss[f]);
                ss[f] = fmaf(b.x, b.x, ss[f]); ss[f] = fmaf(b.y, b.y, ss[f]); ss[f] = fmaf(b.z, b.z, ss[f]); ss[f] = fmaf(b.w, b.w, ss[f]);
                const unsigned w0 = pack_bf16x2(a.x, a.y), w1 = pack_bf16x2(a.z, a.w), w2 = pack_bf16x2(b.x, b.y), w3 = pack_bf16x2(b.z, b.w);
                const uint4 u = make_uint4(w0, w1, w2, w3);
                dfr[f][ks] = __builtin_bit_cast(rb_bf16x8_t, u);
            }
        // next step's operands: the query chunk into the other buffer (everyone passed the barrier: its readers are done)
        int cn = c + 1;
        int64_t bn = blk;
        if (cn == nchunk) { cn = 0; bn += gridDim.x; }
        if (s + 1 < steps) {
            issue_q((int)((s + 1) & 1), cn);
            load_db(bn, cn);
        }
        const unsigned char* qs = smem + (s & 1) * QBYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < NQT; ++t) {
                const rb_bf16x8_t qf = __builtin_bit_cast(rb_bf16x8_t, *reinterpret_cast<const rb_s16x8_t*>(qs + qr[ks] + t * 2048));
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr[f][ks], qf, acc[f][t], 0, 0, 0);
            }
        if (c + 1 == nchunk) {
            // block done: lane holds S[db row 16 f + 4 g + e][query 16 t + r16]
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float v = ss[f];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int64_t rown = blk * BR + wave * (16 * NF) + 16 * f + r16;
                if (rown < n_db) norm2_max = fmaxf(norm2_max, v);
                ss[f] = 0.f;
#pragma unroll
                for (int t = 0; t < NQT; ++t) {
                    const int q = q0 + 16 * t + r16;
                    const float tq = q < n_q ? thr[q] : INFINITY;  // (once per 128-row block: an L1 hit, not 13 live registers)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sc = acc[f][t][e];
                        const int64_t row = blk * BR + wave * (16 * NF) + 16 * f + 4 * g + e;
                        if (sc >= tq && row < n_db) {
                            const unsigned pos = atomicAdd(&cnt[q], 1u);
                            if (pos < (unsigned)cap)
                                cand[q * (int64_t)cap + pos] = ((unsigned long long)rb_ordered_bits(sc) << 32) |
                                                               (unsigned long long)(0xFFFFFFFFu - (uint32_t)(row_base + row));
                        }
                    }
                    acc[f][t] = (rb_f32x4_t){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        c = cn;
        blk = bn;
    }
    norm2_max = wave_max(norm2_max);
    if (lane == 0) atomicMax(max_norm2_bits, __float_as_uint(norm2_max));  // non-negative floats order like their bits
}

}  // namespace swat
using namespace swat;

// Candidate pass: queries_bf16 [n_q, d] (bf16, row-major), db [n_db, d] fp32.  Every (q, row) with approximate score >=
// thr_dev[q] is appended to cand_keys [n_q, cap] (uint64: ordered(approx) << 32 | ~(row_base + row)), cand_cnt [n_q]
// (uint32, zero it first; counts keep growing past cap: overflow is visible).  max_norm2 (device float, zero it first)
// receives max |d_row|^2.  See include/swat_hip.h.
SWAT_API int swat_sim_candidates_bf16(const void* queries_bf16, int n_q, const float* db, int64_t n_db, int d,
                                      const float* thr_dev, int64_t row_base, void* cand_keys, int cap, void* cand_cnt,
                                      float* max_norm2, swat_stream_t s) {
    SWAT_REQUIRE(queries_bf16 && db && thr_dev && cand_keys && cand_cnt && max_norm2, "swat_sim_candidates_bf16: null pointer");
    SWAT_REQUIRE(d > 0 && d % 64 == 0, "swat_sim_candidates_bf16: d=%d must be a multiple of 64", d);
    SWAT_REQUIRE(n_q > 0 && n_db > 0, "swat_sim_candidates_bf16: empty operand");
    SWAT_REQUIRE(row_base + n_db < (1ll << 32), "swat_sim_candidates_bf16: global rows must fit 32 bits");
    SWAT_REQUIRE((int64_t)n_q * d < (1ll << 30), "swat_sim_candidates_bf16: query matrix too large for 32-bit offsets");
    hipStream_t st = (hipStream_t)s;
    static int variant = -1;  // tuning (SWAT_DENSE_VARIANT): 1 = one 16-row fragment per wave, 3 workgroups per CU (default); 2 = two fragments, 2 per CU
    if (variant < 0) {
        const char* e = getenv("SWAT_DENSE_VARIANT");
        variant = e ? atoi(e) : 1;
    }
#define SWAT_CAND(NQT_, NF_, MAXG_)                                                                                      \
    {                                                                                                                    \
        const int64_t nblocks = (n_db + 64 * NF_ - 1) / (64 * NF_);                                                      \
        const unsigned grid = (unsigned)(nblocks < MAXG_ ? nblocks : MAXG_);                                             \
        hipLaunchKernelGGL((sim_candidates_bf16_kernel<NQT_, NF_>), dim3(grid), dim3(256), 0, st,                        \
                           (const unsigned short*)queries_bf16, n_q, q0, db, n_db, d, thr_dev, row_base,                 \
                           (unsigned long long*)cand_keys, cap, (unsigned*)cand_cnt, (unsigned*)max_norm2);              \
    }
    for (int q0 = 0; q0 < n_q;) {
        const int left = n_q - q0;
        if (left > 128) {  // 13 query tiles = 208 queries, 52 KB of LDS: three (NF 1) or two (NF 2) workgroups per CU, all resident
            if (variant == 2) SWAT_CAND(13, 2, 512) else SWAT_CAND(13, 1, 768)
            q0 += 16 * 13;
        } else if (left > 32) {
            SWAT_CAND(8, 1, 1024)
            q0 += 128;
        } else {
            SWAT_CAND(2, 1, 2048)
            q0 += 32;
        }
    }
#undef SWAT_CAND
    return check_launch("swat_sim_candidates_bf16");
}
