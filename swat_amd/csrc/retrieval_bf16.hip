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
// Two forms.  The DEEP form (sim_candidates_bf16_deep_kernel; d % 128 == 0, n_q <= 4096, queries as the packed image of
// pack_queries_kernel) is the product path for every CLIP width (512 / 768 / 1024):
//   * 12 waves per workgroup, one workgroup per CU, each wave owns 16 database rows of a 192-row block against ALL query tiles
//     (NQT x 16 queries, NQT accumulators of 16x16); the twelve waves share one query chunk, so the L2 -> LDS query stream is
//     0.55x the database bytes (1.66x with 4 waves per workgroup — measured as the largest single cost of that shape);
//   * 32-deep steps; database fragments never touch LDS: lane (r16, g) loads 2 x 16 B = k 4g..4g+3 and 16+4g..16+4g+3 of row r16 straight into the
//     MFMA A-fragment registers, FOUR steps ahead (a ring of four register sets, loop unrolled four times, inline-asm loads
//     off a scalar base so the compiler neither counts nor drains them);
//   * query chunks [16 NQT][32] bf16 come from the packed image — 1 KiB per (chunk, tile), already in LDS order, read
//     linearly by LDS-DMA — two steps ahead into a ring of four LDS buffers; one s_barrier and one COUNTED vmcnt wait per
//     step (vmcnt completes in order: "at most my_pieces + 4 outstanding" = this step's operands have landed, three steps
//     of database rows still fly); fragment reads are inline-asm ds_read_b128 with counted lgkmcnt (a compiler-visible LDS
//     read behind a global_load_lds is guarded with vmcnt(0), which would drain the queue every step);
//   * candidates leave without atomics: ballot prefix + a wave-uniform counter place 12-B records in the wave's private
//     staging slot (dword stores off a scalar base).  An atomic with return in the block epilogue cost 0.23 ms of the
//     0.88 ms pass — its wait drains every database load in flight.  regroup_candidates_kernel (20 us) files the records
//     under their queries with one global atomic per (16 slots, query).
// The 64-DEEP form (sim_candidates_bf16_kernel, row-major bf16 queries, any d % 64 == 0) is the earlier shape — 4 waves,
// 64-row blocks, one step ahead — kept for widths the deep form does not take.
//
// Measured (MI355X, 200 queries x 1.25 M x 512, tools/perf_dense.py, profiles/r02_retrieval_*): deep kernel 0.56-0.60 ms =
// 4.3-4.6 TB/s (0.54-0.57 of the 8 TB/s peak; HBM traffic 2.61 GB per launch against 2.56 GB algorithmic), whole exact top-k
// 0.76-0.79 ms (memset + scan + regroup 0.57-0.62, approximate top-K2 0.04, exact re-score 0.08, exact top-k 0.02, proof +
// flag 0.03) against 3.6 ms for the f32-matrix-core scan.  Timing ablations of the 4-wave shape (SWAT_DENSE_ABL, tuning only):
// no query DMA 0.52 ms, no MFMA / LDS reads 0.59, no barrier 0.83 of 0.94; row-major query pieces (16 half-used lines per
// instruction instead of 8 full ones) cost 0.31 ms; the 12-wave kernel without any candidate (ablation 16) runs 480 us against
// 565 us with them — neither an LDS queue in front of the staging stores nor thresholds in registers moved that.  Tried and dropped: two fragments per wave (2 workgroups per CU:
// 1.00 ms), a fifth wave touching the database 8 / 16 steps ahead (1.27 ms), eight steps of rows in flight (spills).
#include <stdlib.h>
#include <utility>
#include "common.hpp"
#include "../../include/swat_hip.h"

namespace swat {

typedef __attribute__((ext_vector_type(8))) __bf16 rb_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float rb_f32x4_t;
typedef __attribute__((ext_vector_type(8))) short rb_s16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned rb_u32x4_t;

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

template <int N_, typename F, int... Is>
__device__ __forceinline__ void rb_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N_, typename F>
__device__ __forceinline__ void rb_static_for(F&& f) {
    rb_static_for_impl<N_>(f, std::make_integer_sequence<int, N_>{});
}

// inline-asm LDS fragment reads with hand-counted waits: behind a global_load_lds the compiler guards every LDS read it
// knows about with vmcnt(0) (it cannot tell the ring buffers apart), which would drain the loads just issued
template <int IMM>
__device__ __forceinline__ rb_bf16x8_t rb_lds_b128(unsigned addr) {
    rb_bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
template <int N_>
__device__ __forceinline__ void rb_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory");
}
template <int N_>
__device__ __forceinline__ void rb_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
__device__ __forceinline__ void rb_pin(rb_bf16x8_t& v) { asm volatile("" : "+v"(v)); }

// The deep form (d % (32 D) == 0): 32-deep steps, database registers D steps ahead (a ring of D register sets, the step loop
// unrolled D times), query chunks [16 NQT][32] bf16 two steps ahead in a ring of four LDS buffers.  Per step every wave issues,
// in this order, the query pieces of step s + 2 and its database loads of step s + D; vmcnt completes in order, so at the top
// of step s "at most LPS + 4 outstanding" (the database loads of steps s-2+D and s-1+D and the query pieces of step s+1) means
// the query pieces of step s and the database registers of step s have landed while up to D-1 steps of database rows stay in
// flight.  Past the end of the stream the same loads are re-issued on the last position, which keeps the counts uniform.
// DB16: the database operand is the bf16 SHADOW of the index (dense_shadow_kernel: rows rounded once, each 32-k chunk stored in the
// scan's own k order, so a lane's fragment is ONE 16-byte load) — half the HBM bytes per row; the row norms come from the build.
// KD (shadow only): 32-k chunks per step.  KD = 2 halves the barriers and counted waits per row (timing ablations of the shadow scan,
// 0.485 ms: no barrier 0.326, no query DMA 0.365, no MFMA / LDS reads 0.277, no candidates 0.410, none of them 0.220 = the HBM stream).
template <int NQT, int D, int WAVES, int ABL = 0, bool DB16 = false, int KD = 1>  // WAVES x 16 database rows per block share one query chunk.   ABL: timing ablations (tuning only; results are wrong): 1 no MFMA, 2 no query DMA, 4 no barrier
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 3 : 1) void sim_candidates_bf16_deep_kernel(const unsigned short* __restrict__ qp, int n_tiles, int n_q, int q0,
                                                                          const float* __restrict__ db, int64_t n_db, int d,
                                                                          const float* __restrict__ thr, int64_t row_base,
                                                                          uint4* __restrict__ stage, unsigned stage_cap,
                                                                          unsigned* __restrict__ stage_cnt, unsigned* __restrict__ max_norm2_bits) {
    static_assert(D % 4 == 0, "the LDS ring index is static: D must be a multiple of the ring size");
    constexpr int QROWS = 16 * NQT;
    constexpr int QBYTES = QROWS * 64;       // one 32-deep query chunk
    constexpr int QP = NQT;                  // 1 KiB pieces (16 rows x 64 B) per chunk
    constexpr int LPS = (QP + WAVES - 1) / WAVES;  // pieces of the busiest wave; wave w issues my_lps = its share, no repeats
    constexpr int BR = 16 * WAVES;           // database rows per block
    static_assert(LPS <= 4, "counted waits cover up to four pieces per wave");
    constexpr int RING = 4;
    static_assert(KD == 1 || (DB16 && KD == 2), "two chunks per step: the shadow scan only");
    constexpr int STAGE = KD * QBYTES;       // LDS bytes of one step's query chunks
    constexpr int DBL = (DB16 && KD == 1) ? 1 : 2;  // database loads per step
    extern __shared__ __attribute__((aligned(1024))) unsigned char rb_dyn_smem[];   // KD = 2: 4 x 2 x 13 KiB is past the static limit
    __shared__ __attribute__((aligned(1024))) unsigned char rb_st_smem[KD == 1 ? RING * QBYTES : 1024];
    unsigned char* const smem = KD == 1 ? rb_st_smem : rb_dyn_smem;
    __shared__ float sthr[QROWS];  // thresholds of this pass's queries: the block epilogue's common path issues no memory op

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int nchunk = (d >> 5) / KD;  // steps per row block
    const int64_t nblocks = (n_db + BR - 1) / BR;
    if ((int64_t)blockIdx.x >= nblocks) return;
    if (tid < QROWS) sthr[tid] = q0 + tid < n_q ? thr[q0 + tid] : INFINITY;  // visible after the first step's barrier
    const int wave_slot = blockIdx.x * WAVES + wave;
    uint4* __restrict__ my_stage = stage + (size_t)wave_slot * stage_cap;
    unsigned wcount = __builtin_amdgcn_readfirstlane(stage_cnt[wave_slot]);  // earlier passes of the same call
    const int64_t my_blocks = (nblocks - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t steps = my_blocks * nchunk;

    // query pieces of this wave: piece p = tile q0 / 16 + p of the packed image (pack_queries_kernel: the 1 KiB LDS image of
    // 16 queries x 32 k, contiguous per (chunk, tile)) — one linear 1 KiB read per instruction: scalar base + lane * 16
    int qdst[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) qdst[i] = (wave + WAVES * i) * 1024;
    const int my_lps = wave < QP ? (QP - 1 - wave) / WAVES + 1 : 0;  // uniform per wave
    const unsigned lane16 = (unsigned)lane * 16u;
    const char* qbase = reinterpret_cast<const char*>(qp) + (size_t)(q0 >> 4) * 1024;  // uniform
    const size_t chunk_stride = (size_t)n_tiles * 1024;
    auto issue_q = [&](int buf, int c) {
#pragma unroll
        for (int h = 0; h < KD; ++h)
#pragma unroll
            for (int i = 0; i < LPS; ++i)
                if (i < my_lps) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qbase + (size_t)(KD * c + h) * chunk_stride + qdst[i] + lane16),
                                                 (__attribute__((address_space(3))) void*)(smem + buf * STAGE + h * QBYTES + qdst[i]), 16, 0, 0);
    };
    // database fragment of a step: lane reads k = 32 c + 8 g .. + 7 of row blk * BR + wave * 16 + r16 (16 rows x 128 B lines):
    // a scalar base (block, chunk) + one constant 32-bit lane offset; the last, partial block clamps its rows to n_db - 1
    const unsigned row_in_blk = (unsigned)(wave * 16 + r16);
    constexpr unsigned EB = DB16 ? 2u : 4u;  // bytes per database element; a 32-k chunk of a row is 128 B (fp32) / 64 B (shadow)
    const unsigned voff = DB16 ? row_in_blk * (unsigned)d * 2u + 16u * g : (row_in_blk * (unsigned)d + 4u * g) * 4u;  // fp32: k 4g..4g+3 (and +16 below): 64 contiguous bytes per row and instruction; shadow: 16-B unit g of the chunk
    const int64_t last_blk = nblocks - 1;
    const unsigned last_rows = (unsigned)(n_db - last_blk * BR);  // 1 .. BR
    const unsigned lrow = row_in_blk < last_rows ? row_in_blk : last_rows - 1;
    const unsigned voff_last = DB16 ? lrow * (unsigned)d * 2u + 16u * g : (lrow * (unsigned)d + 4u * g) * 4u;
    const size_t blk_bytes = (size_t)BR * d * EB;
    rb_f32x4_t nx[D][2];
    const unsigned qr = (unsigned)(r16 * 64 + ((g ^ (((r16 >> 3) & 1) * 3)) << 4));  // pack_queries_kernel: conflict-free per ds_read_b128 lane group
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
    rb_f32x4_t acc[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) acc[t] = (rb_f32x4_t){0.f, 0.f, 0.f, 0.f};
    float ss = 0.f, norm2_max = 0.f;

    // positions: compute (blk, c); database loads D steps ahead (lblk, lc); query chunk two steps ahead (qc)
    int64_t blk = blockIdx.x, lblk = blockIdx.x, ls = 0;
    int c = 0, lc = 0, qc = 0;
    auto next_q = [&]() { qc = qc + 1 == nchunk ? 0 : qc + 1; };
    auto next_l = [&]() {
        if (ls + 1 < steps) {  // past the end: stay on the last position
            ++ls;
            if (++lc == nchunk) { lc = 0; lblk += gridDim.x; }
        }
    };
    issue_q(0, qc); next_q();
    // database loads are inline asm: the compiler's own bookkeeping would drain the queue (vmcnt(0)) at every use because it
    // cannot count across the loop's merges; the registers are handed to the compiler by the pin behind the counted wait
    auto load_db = [&](rb_f32x4_t (&dst)[2]) {
        const char* base = reinterpret_cast<const char*>(db) + lblk * blk_bytes + (size_t)lc * (32 * EB * KD);  // uniform
        const unsigned off = lblk == last_blk ? voff_last : voff;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[0]) : "v"(off), "s"(base));
        if constexpr (DBL == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(dst[1]) : "v"(off), "s"(base));  // fp32: k + 16; shadow: the next chunk
        next_l();
    };
    load_db(nx[0]);
    issue_q(1, qc); next_q();
    rb_static_for<D - 1>([&](auto R) { load_db(nx[decltype(R)::value + 1]); });

    for (int64_t s0 = 0; s0 < steps; s0 += D) {
        rb_static_for<D>([&](auto R) {
            constexpr int r = decltype(R)::value;
            // two steps of database loads and the next step's query pieces may stay in flight behind this wait
            if (s0 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first D steps drain the prologue
            else if ((ABL & 2) || my_lps == 0) rb_wait_vm<2 * DBL>();
            else if (LPS == 1 || my_lps == 1) rb_wait_vm<2 * DBL + KD>();
            else if (LPS == 2 || my_lps == 2) rb_wait_vm<2 * DBL + 2 * KD>();
            else if (LPS == 3 || my_lps == 3) rb_wait_vm<2 * DBL + 3 * KD>();
            else rb_wait_vm<2 * DBL + 4 * KD>();
            if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
            unsigned w0, w1, w2, w3;
            unsigned x0 = 0u, x1 = 0u, x2 = 0u, x3 = 0u;  // KD = 2: the second chunk's fragment
            if constexpr (DB16) {
                asm volatile("" : "+v"(nx[r][0])::"memory");  // landed: the counted wait above covers it
                const rb_f32x4_t a = nx[r][0];
                w0 = __float_as_uint(a[0]); w1 = __float_as_uint(a[1]); w2 = __float_as_uint(a[2]); w3 = __float_as_uint(a[3]);
                if constexpr (KD == 2) {
                    asm volatile("" : "+v"(nx[r][1])::"memory");
                    const rb_f32x4_t b = nx[r][1];
                    x0 = __float_as_uint(b[0]); x1 = __float_as_uint(b[1]); x2 = __float_as_uint(b[2]); x3 = __float_as_uint(b[3]);
                    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::"memory");
                }
            } else {
                asm volatile("" : "+v"(nx[r][0]), "+v"(nx[r][1])::"memory");  // landed: the counted wait above covers them
                const rb_f32x4_t a = nx[r][0], b = nx[r][1];
                ss = fmaf(a[0], a[0], ss); ss = fmaf(a[1], a[1], ss); ss = fmaf(a[2], a[2], ss); ss = fmaf(a[3], a[3], ss);
                ss = fmaf(b[0], b[0], ss); ss = fmaf(b[1], b[1], ss); ss = fmaf(b[2], b[2], ss); ss = fmaf(b[3], b[3], ss);
                w0 = pack_bf16x2(a[0], a[1]); w1 = pack_bf16x2(a[2], a[3]); w2 = pack_bf16x2(b[0], b[1]); w3 = pack_bf16x2(b[2], b[3]);
            }
            // the registers of this step are consumed HERE, before the loads below are issued: a use behind them would have to
            // wait for the youngest load (the compiler cannot count across the LDS-DMA pieces and drains the queue)
            asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(ss)::"memory");
            const uint4 u = make_uint4(w0, w1, w2, w3);
            const rb_bf16x8_t dfr = __builtin_bit_cast(rb_bf16x8_t, u);
            const rb_bf16x8_t dfr1 = __builtin_bit_cast(rb_bf16x8_t, make_uint4(x0, x1, x2, x3));
            // step s + 2's query chunk (its buffer was last read in step s - 2: everyone is past that), step s + D's rows
            if constexpr (!(ABL & 2)) { issue_q((r + 2) & 3, qc); next_q(); }
            load_db(nx[r]);
            // NQT fragment reads, QD in flight: read t + QD reuses the register of tile t behind its MFMA
            constexpr int NV = KD * NQT;  // fragment v: chunk v / NQT of the step, query tile v % NQT — 1 KiB apart in the stage buffer
            constexpr int QD = NV < 4 ? NV : 4;
            const unsigned qs = lds_base + (unsigned)((r & 3) * STAGE) + qr;
            rb_bf16x8_t qf[QD];
            if constexpr (ABL & 1) {
                acc[r % NQT][0] += __builtin_bit_cast(float, u.x ^ u.y ^ u.z ^ u.w);
            } else {
            rb_static_for<QD>([&](auto T) { qf[decltype(T)::value] = rb_lds_b128<decltype(T)::value * 1024>(qs); });
            rb_static_for<NV>([&](auto T) {
                constexpr int t = decltype(T)::value;
                rb_wait_lgkm<(NV - 1 - t < QD - 1 ? NV - 1 - t : QD - 1)>();
                rb_pin(qf[t % QD]);
                acc[t % NQT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t < NQT ? dfr : dfr1, qf[t % QD], acc[t % NQT], 0, 0, 0);
                if constexpr (t + QD < NV) qf[t % QD] = rb_lds_b128<(t + QD) * 1024>(qs);
            });
            }
            if constexpr (r == D - 1) {  // nchunk % D == 0: a block can only end on the last step of the unrolled group
                if (c + 1 == nchunk) {
                    // block done: lane holds S[db row 4 g + e][query 16 t + r16].  Candidates go to this wave's private staging
                    // slot — 16-B records (key lo, key hi, query), positions from a ballot prefix and a wave-uniform counter: no
                    // atomic, no load, so nothing here waits on the database loads in flight (an atomic with return would drain
                    // them: vmcnt completes in order).  regroup_candidates_kernel files the records under their queries.
                    float v = ss;
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    const unsigned row0 = (unsigned)(blk * BR) + wave * 16, n_db32 = (unsigned)n_db;  // row_base + n_db < 2^32
                    if (row0 + r16 < n_db32) norm2_max = fmaxf(norm2_max, v);
                    ss = 0.f;
                    const unsigned key0 = 0xFFFFFFFFu - (uint32_t)row_base - row0 - 4u * g;
#pragma unroll
                    for (int t = 0; t < NQT; ++t) {
                        asm volatile("" ::: "memory");  // one threshold live at a time (the compiler would hoist all NQT reads)
                        const float tq = sthr[16 * t + r16];
                        // (hits are rare: one ballot over the lane's best of four decides whether the four per-row ballots run)
                        if ((ABL & 31) || __ballot(fmaxf(fmaxf(acc[t][0], acc[t][1]), fmaxf(acc[t][2], acc[t][3])) >= tq))
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float sc = acc[t][e];
                            const unsigned row = row0 + 4 * g + e;
                            const bool hit = ((ABL & 15) ? sc == 123.456f : (ABL & 16) ? sc >= tq + 100.f : sc >= tq) && row < n_db32;
                            const unsigned long long m = __ballot(hit);
                            if (m) {
                                const unsigned pos = wcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                if (hit && pos < stage_cap) {
                                    // three dword stores off a scalar base: a 16-B store needs a fresh aligned register quad, which
                                    // at this pressure costs a spill reload (a scratch load, i.e. a drain of the queue) per hit
                                    const unsigned o = pos * 16u, k_lo = key0 - e, k_hi = rb_ordered_bits(sc), qq = (unsigned)(q0 + 16 * t) + r16;
                                    asm volatile("global_store_dword %0, %1, %2" ::"v"(o), "v"(k_lo), "s"(my_stage) : "memory");
                                    asm volatile("global_store_dword %0, %1, %2 offset:4" ::"v"(o), "v"(k_hi), "s"(my_stage) : "memory");
                                    asm volatile("global_store_dword %0, %1, %2 offset:8" ::"v"(o), "v"(qq), "s"(my_stage) : "memory");
                                }
                                wcount += (unsigned)__popcll(m);
                            }
                        }
                        acc[t] = (rb_f32x4_t){0.f, 0.f, 0.f, 0.f};
                    }
                    c = 0;
                    blk += gridDim.x;
                } else {
                    ++c;
                }
            } else {
                ++c;
            }
        });
    }
    // the re-issued tail pieces must land before the LDS is handed back, and the tail's database registers stay reserved
    // until their loads have landed
#pragma unroll
    for (int r = 0; r < D; ++r) {
        if constexpr (DBL == 1) asm volatile("s_waitcnt vmcnt(0)" ::"v"(nx[r][0]) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::"v"(nx[r][0]), "v"(nx[r][1]) : "memory");
    }
    norm2_max = wave_max(norm2_max);
    if (lane == 0) {
        if constexpr (!DB16) atomicMax(max_norm2_bits, __float_as_uint(norm2_max));
        stage_cnt[wave_slot] = wcount;  // may exceed stage_cap: the regroup pass reports the overflow
    }
}

// What bounds the shadow scan at 13 query tiles (round 5; 200 x 1.25 M x 512, 0.30 ms of scan).  Measured, not assumed:
//   * per-wave cycle sums from the PROF build (tools/prof_dense_waves.py, profiles/r05_dense_wave_cycles.txt): of 568 k cycles a wave
//     spends 5 % in the counted vmcnt wait, 13 % at the step barrier, 56 % in the MFMA phase (its 52 MFMAs, 26 fragment reads, the
//     step's row loads and query DMA) and 20 % in the candidate epilogue; at 16 queries the same kernel runs 207 us = 6.2 TB/s of
//     shadow bytes, i.e. at the achievable HBM rate, with the time in the MFMA phase being load-issue back-pressure.  At 200 queries
//     the rows are NOT late (5 % wait): the eight waves run in lockstep (one barrier per step) and a step is as long as its slowest
//     wave's MFMA phase (two waves share a SIMD's matrix pipe: 1 664 cycles of MFMA per step and SIMD, ~2 300 measured for the
//     second wave of a SIMD) plus, every eighth step, an epilogue both waves of a SIMD run at the same time (~6 k cycles).
//   * the shader clock drops from 2.42 GHz (16 queries) to 2.18 GHz (200 queries) under the matrix load (GRBM_GUI_ACTIVE / duration,
//     tools/pmc_dense_clock.sh): 10 % of the difference between the two is frequency, not cycles.
//   * SQ_LDS_BANK_CONFLICT was 0.48 of SQ_LDS_IDX_ACTIVE through round 4 (query image swizzled for contiguous 16-lane groups, which
//     ds_read_b128 does not have: see pack_queries_kernel); with the right swizzle it is 0 and the LDS is active 21 % of the
//     launch instead of 40 % (profiles/r05_dense_pmc.txt) — at the same duration: the LDS read port was never the limit, and the
//     matrix pipe is 38 % busy.  Instruction-cache misses: 2.8 k per launch (tools/pmc_dense_icache.sh), not a factor either.
//   * candidate hits cost 0.04 ms of the scan (tools/ab_dense_hits.py: thresholds nothing passes, 0.265 ms scan + regroup against
//     0.322): a third of the 26 groups of a block hold one.  Turning a hit into a record with vector code per row (EPI 2) or with a
//     scalar walk over the hit bits cost the same; parking the hit lanes and converting them in one rolled loop (EPI 3) is 4 %
//     faster (0.301 against 0.314 ms scan + regroup, same process).
//   * tried on top of EPI 3 and dropped: the block's first step taking a constant-zero C operand instead of 104 zeroed registers, with
//     a first epilogue round free of the `done` mask — slower (0.312 against 0.301 ms scan + regroup; 0.275 against 0.265 with no
//     hits at all: the second copy of the step's MFMA loop costs more than the moves it saves).
//   * falsified earlier in the round: the epilogue's compare form (0.329 vs 0.334 ms), record stores sharing vmcnt with the row loads
//     (0.331), DRAM locality (tiled shadow, 0.329 vs 0.336; fragment-ordered so that a load instruction reads 1 KiB contiguous: no
//     change), the depth of the row stream (0.339 vs 0.341), wave priorities (s_setprio for the second wave of a SIMD: 0.354 vs
//     0.355), issuing the step's row loads between its MFMAs instead of up front (SP: 2 %, kept).
// One query fragment already feeds two MFMAs (NF = 2); NF = 4 or a 32 x 32 x 16 tile with two row fragments would need 208-224
// accumulator registers (four waves of four fragments with one wave per SIMD and 512 registers were compiled: 269 spills).
//
// The shadow scan with TWO 16-row fragments per wave (8 waves, 256-row blocks, 64-deep steps, database registers four steps ahead):
// every query fragment read from LDS feeds two MFMAs, and a row costs half the barriers, counted waits and query-chunk DMA of the
// one-fragment form above — whose ablations show exactly those as its cost once the rows come from the bf16 shadow (0.449 ms: no
// barrier 0.325, no query DMA 0.331, no MFMA / LDS reads 0.257; the bare row stream 0.220).  Same candidate records bit for bit.
template <int H>
__device__ __forceinline__ void rb_load_row(rb_f32x4_t& dst, unsigned voff, const char* base) {   // half H (1 KiB apart) of a fragment
    if constexpr (H == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base));
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(dst) : "v"(voff), "s"(base));
}
__device__ unsigned long long g_shadow2_prof[256 * 8 * 5];   // PROF: per (workgroup, wave) cycles {wait, barrier, mfma phase, epilogue, whole kernel}
constexpr int SHADOW2_RCAP = 256;   // candidate records a wave keeps in LDS before it stores them (16 B each: 32 KB per workgroup)
constexpr int SHADOW2_LCAP = 128;   // EPI == 3: lane-records (four scores + a tag, 20 B) a wave parks per round of its candidate epilogue
template <int NQT, int EPI = 3, int QA = 2, int SP = 1, bool PROF = false>   // EPI (tuning, A/B runs): how a hit becomes a record — 2 = vector code per row
                                  // of a hit group; 3 = hit lanes parked in LDS, one rolled loop converts them (the product form).  Same set of records.
                                  // QA: steps the query chunks are issued ahead (2 = the product form; 3 = tuning, see the main loop)
                                  // SP: 1 = the step's row loads are issued BETWEEN its MFMAs (the product form), 0 = all at the step's start
                                  // PROF (tuning): every wave adds up the shader cycles it spends in the counted wait, at the barrier, in the
                                  // MFMA phase and in the candidate epilogue (g_shadow2_prof, swat_dense_debug_profile)
__global__ __launch_bounds__(512, 1) void sim_candidates_shadow2_kernel(const unsigned short* __restrict__ qp, int n_tiles, int n_q, int q0,
                                                                        const unsigned short* __restrict__ db, int64_t n_db, int d,
                                                                        const float* __restrict__ thr, int64_t row_base,
                                                                        uint4* __restrict__ stage, unsigned stage_cap,
                                                                        unsigned* __restrict__ stage_cnt) {
    constexpr int WAVES = 8, NF = 2, KD = 2, D = 4, RING = 4;   // (WAVES x NF = 16: 256-row blocks, the tiled shadow image)
    constexpr int QROWS = 16 * NQT;
    constexpr int QBYTES = QROWS * 64;       // one 32-deep query chunk
    constexpr int LPS = (NQT + WAVES - 1) / WAVES;
    constexpr int BR = 16 * WAVES * NF;      // 256 database rows per block
    constexpr int STAGE = KD * QBYTES;
    constexpr int DBL = NF * KD;             // database loads per step and lane
    static_assert(LPS <= 4, "counted waits cover up to four pieces per wave");
    extern __shared__ __attribute__((aligned(1024))) unsigned char rb_dyn_smem[];
    unsigned char* const smem = rb_dyn_smem;
    __shared__ float sthr[QROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int nchunk = d >> 6;  // 64-deep steps per row block
    const int64_t nblocks = (n_db + BR - 1) / BR;
    if ((int64_t)blockIdx.x >= nblocks) return;
    if (tid < QROWS) sthr[tid] = q0 + tid < n_q ? thr[q0 + tid] : INFINITY;
    const int wave_slot = blockIdx.x * WAVES + wave;
    uint4* __restrict__ my_stage = stage + (size_t)wave_slot * stage_cap;
    unsigned wcount = __builtin_amdgcn_readfirstlane(stage_cnt[wave_slot]);
    const int64_t my_blocks = (nblocks - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t steps = my_blocks * nchunk;

    int qdst[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) qdst[i] = (wave + WAVES * i) * 1024;
    const int my_lps = wave < NQT ? (NQT - 1 - wave) / WAVES + 1 : 0;  // uniform per wave
    const unsigned lane16 = (unsigned)lane * 16u;
    const char* qbase = reinterpret_cast<const char*>(qp) + (size_t)(q0 >> 4) * 1024;
    const size_t chunk_stride = (size_t)n_tiles * 1024;
    auto issue_q = [&](int buf, int c) {
#pragma unroll
        for (int h = 0; h < KD; ++h)
#pragma unroll
            for (int i = 0; i < LPS; ++i)
                if (i < my_lps) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qbase + (size_t)(KD * c + h) * chunk_stride + qdst[i] + lane16),
                                                 (__attribute__((address_space(3))) void*)(smem + buf * STAGE + h * QBYTES + qdst[i]), 16, 0, 0);
    };
    // fragment f of the wave: rows (wave * NF + f) * 16 + r16 of the block.  The shadow is TILED (dense_shadow_kernel): a 64-deep step of
    // a block is 256 rows x 128 B, contiguous; the lane's two 16-B units of a row sit 64 B apart.  Rows past n_db are zero in the image.
    unsigned voff[NF];
    const int64_t last_blk = nblocks - 1;
#pragma unroll
    for (int f = 0; f < NF; ++f) voff[f] = (unsigned)(wave * NF + f) * 2048u + (unsigned)lane * 16u;
    const size_t blk_bytes = (size_t)BR * d * 2;
    rb_f32x4_t nx[D][NF][KD];
    const unsigned qr = (unsigned)(r16 * 64 + ((g ^ (((r16 >> 3) & 1) * 3)) << 4));  // pack_queries_kernel: conflict-free per ds_read_b128 lane group
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
    rb_f32x4_t acc[NF][NQT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NQT; ++t) acc[f][t] = (rb_f32x4_t){0.f, 0.f, 0.f, 0.f};

    // EPI == 2: candidate records wait in LDS.  gfx950 counts loads and stores in ONE vmcnt, so a record stored to global memory
    // between a block's row loads and the next block's counted waits makes those waits (vmcnt <= N) demand that many MORE of the row
    // loads in flight be complete: every block's ~40 record stores drained the wave's whole four-step ring (round 4 saw it as "no
    // candidate stores: 0.375 of 0.449 ms" and "epilogue 0.12-0.15 ms in every form" — the forms all stored inside the loop).  The
    // records of a wave (~14 per block, ~270 per launch on a 1.25 M-row shard) are kept in its own RCAP-record LDS buffer behind the
    // query ring (ds_write: lgkmcnt, in order, invisible to vmcnt) and go to the wave's staging slot in ONE burst after the scan;
    // a buffer with fewer than 64 free records at the start of a block's epilogue is flushed there, and a record that still finds it
    // full goes straight to its place in the slot — those blocks pay the old price, nothing is lost.
    constexpr int RCAP = SHADOW2_RCAP;
    const unsigned rec_lds = lds_base + (unsigned)(RING * STAGE) + (unsigned)wave * (unsigned)(RCAP * 16);
    constexpr int LCAP = SHADOW2_LCAP;
    const unsigned lrec_lds = lds_base + (unsigned)(RING * STAGE) + (unsigned)(WAVES * RCAP * 16) + (unsigned)wave * (unsigned)(LCAP * 20);
    unsigned wl = 0;   // records in the LDS buffer (wave uniform)
    auto flush_records = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned in_lds = wl < (unsigned)RCAP ? wl : (unsigned)RCAP;   // (records past RCAP went to their places directly)
        for (unsigned i = (unsigned)lane; i < in_lds; i += 64u) {
            rb_f32x4_t rec;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rec) : "v"(rec_lds + i * 16u) : "memory");
            const unsigned o = (wcount + i) * 16u;
            if (wcount + i < stage_cap) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(o), "v"(rec), "s"(my_stage) : "memory");
        }
        wcount += wl;
        wl = 0;
    };
    unsigned long long pt[5] = {0, 0, 0, 0, 0}, pm = 0;
    auto stamp = [&](int i) {   // (no LDS read of the step is in flight at any stamp: the s_memtime wait does not disturb the counted lgkm waits)
        if constexpr (PROF) {
            const unsigned long long now = __builtin_readcyclecounter();
            if (i >= 0) pt[i] += now - pm;
            pm = now;
        }
    };
    const unsigned long long p_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    int64_t blk = blockIdx.x, lblk = blockIdx.x, ls = 0;
    int c = 0, lc = 0, qc = 0;
    auto next_q = [&]() { qc = qc + 1 == nchunk ? 0 : qc + 1; };
    auto next_l = [&]() {
        if (ls + 1 < steps) {
            ++ls;
            if (++lc == nchunk) { lc = 0; lblk += gridDim.x; }
        }
    };
    issue_q(0, qc); next_q();
    auto step_base = [&]() { return reinterpret_cast<const char*>(db) + lblk * blk_bytes + (size_t)lc * (BR * 128); };  // uniform: this step's 32 KB
    auto load_piece = [&voff](rb_f32x4_t (&dst)[NF][KD], const char* base, auto P) {   // piece p of a step's NF x KD row loads
        constexpr int p = decltype(P)::value;
        rb_load_row<p % KD>(dst[p / KD][p % KD], voff[p / KD], base);
    };
    auto load_db = [&](rb_f32x4_t (&dst)[NF][KD]) {
        const char* base = step_base();
        rb_static_for<DBL>([&](auto P) { load_piece(dst, base, P); });
        next_l();
    };
    load_db(nx[0]);
    issue_q(1, qc); next_q();
    if constexpr (QA == 3) { issue_q(2, qc); next_q(); }
    rb_static_for<D - 1>([&](auto R) { load_db(nx[decltype(R)::value + 1]); });

    for (int64_t s0 = 0; s0 < steps; s0 += D) {
        rb_static_for<D>([&](auto R) {
            constexpr int r = decltype(R)::value;
            // vmcnt retires in issue order and a step issues [query chunk, row loads]: the chunk of step s, issued at step s - 2 (QA = 2),
            // sits BEHIND the rows of step s + 1, so waiting for it also waits for those rows one step early — the row stream is
            // three steps deep at this wait, not the four its registers hold.  QA = 3 issues the chunk at step s - 3 (the ring's
            // fourth slot, free since this step's barrier), ahead of those rows: four steps deep.  Measured (round 5, same process,
            // 200 x 1.25 M x 512): 0.339 vs 0.341 ms — the depth of the row stream is NOT what bounds the pass; QA = 2 stays.
            constexpr int DBW = (QA == 3 ? 3 : 2) * DBL, QW = QA == 3 ? 2 : 1;
            stamp(-1);
            if (s0 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (my_lps == 0) rb_wait_vm<DBW>();
            else if (LPS == 1 || my_lps == 1) rb_wait_vm<DBW + QW * KD>();
            else if (LPS == 2 || my_lps == 2) rb_wait_vm<DBW + QW * 2 * KD>();
            else if (LPS == 3 || my_lps == 3) rb_wait_vm<DBW + QW * 3 * KD>();
            else rb_wait_vm<DBW + QW * 4 * KD>();
            stamp(0);
            __builtin_amdgcn_s_barrier();
            stamp(1);
            rb_bf16x8_t dfr[NF][KD];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int h = 0; h < KD; ++h) {
                    asm volatile("" : "+v"(nx[r][f][h])::"memory");  // landed: the counted wait above covers it
                    rb_f32x4_t v = nx[r][f][h];
                    asm volatile("" : "+v"(v)::"memory");            // consumed HERE, before the loads below overwrite the set
                    dfr[f][h] = __builtin_bit_cast(rb_bf16x8_t, v);
                }
            issue_q((r + QA) & 3, qc); next_q();
            // SP: the hardware keeps only so many bytes of a CU's loads on their way at once (the rest wait in front of the memory
            // pipe, whatever depth the software ring has): a step that issues its 32 KB of row loads at once and then runs its
            // MFMAs leaves the memory pipe idle for most of the MFMA phase — the pass cost 'stream + matrix time' (207 us at 2 query
            // tiles, 316 at 13, 254 k of 687 k cycles MFMA-busy; tools/pmc_dense_clock.sh).  Issued between the MFMAs, one load per
            // NV / DBL fragments, the stream runs under them.  Same order of vmcnt entries ([chunk, rows] per step).
            const char* const sbase = step_base();
            if constexpr (!SP) load_db(nx[r]);
            constexpr int NV = KD * NQT;  // fragment v: chunk v / NQT, query tile v % NQT — 1 KiB apart in the stage buffer
            constexpr int QD = NV < 4 ? NV : 4;
            const unsigned qs = lds_base + (unsigned)((r & 3) * STAGE) + qr;
            rb_bf16x8_t qf[QD];
            rb_static_for<QD>([&](auto T) { qf[decltype(T)::value] = rb_lds_b128<decltype(T)::value * 1024>(qs); });
            rb_static_for<NV>([&](auto T) {
                constexpr int t = decltype(T)::value;
                rb_wait_lgkm<(NV - 1 - t < QD - 1 ? NV - 1 - t : QD - 1)>();
                rb_pin(qf[t % QD]);
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    acc[f][t % NQT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr[f][t / NQT], qf[t % QD], acc[f][t % NQT], 0, 0, 0);
                if constexpr (t + QD < NV) qf[t % QD] = rb_lds_b128<(t + QD) * 1024>(qs);
                if constexpr (SP) {
                    constexpr int GAP = NV / DBL > 0 ? NV / DBL : 1;
                    if constexpr (t % GAP == 0 && t / GAP < DBL) load_piece(nx[r], sbase, std::integral_constant<int, t / GAP>{});
                    if constexpr (NV < DBL && t == NV - 1)
                        rb_static_for<DBL - NV>([&](auto P) { load_piece(nx[r], sbase, std::integral_constant<int, NV + decltype(P)::value>{}); });
                }
            });
            if constexpr (SP) next_l();
            stamp(2);
            if constexpr (r == D - 1) {  // nchunk % D == 0: a block can only end on the last step of the unrolled group
                if (c + 1 == nchunk) {
                    {
                        // The block's candidate epilogue: 2 x NQT groups of 4 x 64 scores per wave.  While it runs the wave issues no
                        // loads; the D steps already in flight keep the memory pipe busy for ~4 us, so the epilogue must stay well
                        // below that (round 4: ~6 us per block — one LDS threshold read, a canonicalising max chain and a vector
                        // branch per group, each waited for in turn: 0.12-0.15 ms of the 0.37 ms pass).  Now: the lane's NQT thresholds
                        // are read from LDS once per epilogue (the fragment registers of the step are dead here), a group is four v_cmp straight into scalar masks (no max, no LDS, nothing to
                        // wait for but the compares) reduced and tested on the SCALAR unit, which also skips empty rows of a hit
                        // group; the bound test against n_db is a scalar mask that differs from all-ones in the last block only.
                        // (EPI 2: records in (f, t, e) order, lanes ascending; EPI 3: (f, t), lanes ascending, e.)
                        const unsigned n_db32 = (unsigned)n_db;
                        if (wl > (unsigned)(RCAP - 64)) flush_records();   // the one flush site inside the scan: rarely taken
                        float tqr[NQT];
    #pragma unroll
                        for (int t = 0; t < NQT; ++t) tqr[t] = sthr[16 * t + r16];
                        if constexpr (EPI == 3) {
                            // A third of the groups hold a hit (1.2 lanes on average, ~9 per wave and block), and what a hit costs is
                            // the epilogue's price: 0.06 ms of the 0.36 ms scan + regroup (tools/ab_dense_hits.py) — as vector code
                            // under exec-mask branches per row of the group (EPI == 2: ~40 instructions per row with a hit, at 26
                            // unrolled sites) and just as much as a scalar walk over the hit bits (tried: v_readlane + SGPR arithmetic,
                            // 0.339 against 0.325 ms; note __builtin_amdgcn_readlane on an element of the accumulator vector read
                            // element 0 for every e with this toolchain — inline asm v_readlane_b32 was needed).  Here a group with a hit only PARKS its hit lanes: the lane's four scores and a
                            // tag (group, lane) go to the wave's lane-record buffer in LDS, ~12 instructions; one rolled loop after the
                            // groups (phase B) turns the lane-records into candidate records, a lane per lane-record.  A buffer that
                            // cannot take a group leaves it for another round over the groups (tiny shards, where every score passes).
                            unsigned done = 0u;
                            bool more;
                            do {
                                more = false;
                                unsigned nl = 0u;
                                rb_static_for<NF>([&](auto F_) {
                                    constexpr int f = decltype(F_)::value;
                                    rb_static_for<NQT>([&](auto T_) {
                                        constexpr int t = decltype(T_)::value;
                                        constexpr unsigned gid = (unsigned)(f * NQT + t);
                                        const float tq = tqr[t];
                                        const bool hl = (acc[f][t][0] >= tq) | (acc[f][t][1] >= tq) | (acc[f][t][2] >= tq) | (acc[f][t][3] >= tq);
                                        const unsigned long long M = __ballot(hl);
                                        if (M) {
                                            if (!((done >> gid) & 1u)) {
                                                const unsigned cm = (unsigned)__popcll(M);
                                                if (nl + cm <= (unsigned)LCAP) {
                                                    const unsigned pos = nl + __builtin_amdgcn_mbcnt_hi((unsigned)(M >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)M, 0u));
                                                    if (hl) {
                                                        unsigned tg = gid << 6;
                                                        asm volatile("" : "+s"(tg));   // (else 26 loop-invariant tags are kept in 26 registers)
                                                        asm volatile("ds_write_b128 %0, %1" ::"v"(lrec_lds + pos * 16u), "v"(acc[f][t]) : "memory");
                                                        asm volatile("ds_write_b32 %0, %1" ::"v"(lrec_lds + (unsigned)(LCAP * 16) + pos * 4u), "v"(tg | (unsigned)lane) : "memory");
                                                    }
                                                    nl += cm;
                                                    done |= 1u << gid;
                                                } else {
                                                    more = true;
                                                }
                                            }
                                        }
                                    });
                                });
                                // phase B: lane-records -> candidate records (a hit row of a lane-record = one record)
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                for (unsigned i0 = 0u; i0 < nl; i0 += 64u) {
                                    const unsigned li = i0 + (unsigned)lane;
                                    const bool on = li < nl;
                                    rb_f32x4_t s4 = (rb_f32x4_t){0.f, 0.f, 0.f, 0.f};
                                    unsigned tag = 0u;
                                    if (on)
                                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(s4), "=&v"(tag)
                                                     : "v"(lrec_lds + li * 16u), "v"(lrec_lds + (unsigned)(LCAP * 16) + li * 4u) : "memory");
                                    const unsigned gid = tag >> 6, sl = tag & 63u;
                                    const unsigned f2 = gid >= (unsigned)NQT ? 1u : 0u, t2 = gid - f2 * (unsigned)NQT;
                                    static_assert(NF == 2, "a tag's group index is split into two fragments");
                                    const float tq2 = sthr[16u * t2 + (sl & 15u)];
                                    const unsigned rowl = (unsigned)(blk * BR) + ((unsigned)(wave * NF) + f2) * 16u + 4u * (sl >> 4);
                                    const unsigned keyl = 0xFFFFFFFFu - (uint32_t)row_base - rowl, qq = (unsigned)q0 + 16u * t2 + (sl & 15u);
    #pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float sc = s4[e];
                                        const bool hit = on && sc >= tq2 && rowl + (unsigned)e < n_db32;
                                        const unsigned long long m = __ballot(hit);
                                        if (m) {
                                            const unsigned pos = wl + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                            if (hit) {
                                                const unsigned k_lo = keyl - (unsigned)e, k_hi = rb_ordered_bits(sc);
                                                if (pos < (unsigned)RCAP) {
                                                    const unsigned a = rec_lds + pos * 16u;
                                                    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(a), "v"(k_lo), "v"(k_hi) : "memory");
                                                    asm volatile("ds_write_b32 %0, %1 offset:8" ::"v"(a), "v"(qq) : "memory");
                                                } else if (wcount + pos < stage_cap) {
                                                    const unsigned o = (wcount + pos) * 16u;
                                                    asm volatile("global_store_dword %0, %1, %2" ::"v"(o), "v"(k_lo), "s"(my_stage) : "memory");
                                                    asm volatile("global_store_dword %0, %1, %2 offset:4" ::"v"(o), "v"(k_hi), "s"(my_stage) : "memory");
                                                    asm volatile("global_store_dword %0, %1, %2 offset:8" ::"v"(o), "v"(qq), "s"(my_stage) : "memory");
                                                }
                                            }
                                            wl += (unsigned)__popcll(m);
                                        }
                                    }
                                }
                            } while (more);
    #pragma unroll
                            for (int f = 0; f < NF; ++f)
    #pragma unroll
                                for (int t = 0; t < NQT; ++t) acc[f][t] = (rb_f32x4_t){0.f, 0.f, 0.f, 0.f};
                        } else {
                        unsigned long long vm[NF][4];
    #pragma unroll
                        for (int f = 0; f < NF; ++f)
    #pragma unroll
                            for (int e = 0; e < 4; ++e) vm[f][e] = ~0ull;
                        if (blk == last_blk) {
    #pragma unroll
                            for (int f = 0; f < NF; ++f)
    #pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    vm[f][e] = __ballot((unsigned)(blk * BR) + (unsigned)((wave * NF + f) * 16) + 4u * g + e < n_db32);
                        }
                        rb_static_for<NF>([&](auto F_) {
                            constexpr int f = decltype(F_)::value;
                            const unsigned row0 = (unsigned)(blk * BR) + (unsigned)((wave * NF + f) * 16);
                            const unsigned key0 = 0xFFFFFFFFu - (uint32_t)row_base - row0 - 4u * g;
                            rb_static_for<NQT>([&](auto T_) {
                                constexpr int t = decltype(T_)::value;
                                const float tq = tqr[t];
                                const unsigned long long m0 = __ballot(acc[f][t][0] >= tq) & vm[f][0];
                                const unsigned long long m1 = __ballot(acc[f][t][1] >= tq) & vm[f][1];
                                const unsigned long long m2 = __ballot(acc[f][t][2] >= tq) & vm[f][2];
                                const unsigned long long m3 = __ballot(acc[f][t][3] >= tq) & vm[f][3];
                                if ((m0 | m1) | (m2 | m3)) {
                                    const unsigned long long ms[4] = {m0, m1, m2, m3};
    #pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const unsigned long long m = ms[e];
                                        if (m) {
                                            const float sc = acc[f][t][e];
                                            const bool hit = sc >= tq && row0 + 4 * g + e < n_db32;
                                            // records wait in the wave's LDS buffer (see flush_records): no global store, hence no
                                            // vmcnt entry, between the row loads of this block and the counted waits of the next
                                            const unsigned pos = wl + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                            if (hit) {
                                                const unsigned k_lo = key0 - e, k_hi = rb_ordered_bits(sc), qq = (unsigned)(q0 + 16 * t) + r16;
                                                if (pos < (unsigned)RCAP) {
                                                    const unsigned a = rec_lds + pos * 16u;
                                                    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(a), "v"(k_lo), "v"(k_hi) : "memory");
                                                    asm volatile("ds_write_b32 %0, %1 offset:8" ::"v"(a), "v"(qq) : "memory");
                                                } else if (wcount + pos < stage_cap) {
                                                    // a burst beyond the buffer (each epilogue starts with >= 64 free records): straight to
                                                    // the record's place in the staging slot
                                                    const unsigned o = (wcount + pos) * 16u;
                                                    asm volatile("global_store_dword %0, %1, %2" ::"v"(o), "v"(k_lo), "s"(my_stage) : "memory");
                                                    asm volatile("global_store_dword %0, %1, %2 offset:4" ::"v"(o), "v"(k_hi), "s"(my_stage) : "memory");
                                                    asm volatile("global_store_dword %0, %1, %2 offset:8" ::"v"(o), "v"(qq), "s"(my_stage) : "memory");
                                                }
                                            }
                                            wl += (unsigned)__popcll(m);
                                        }
                                    }
                                }
                                acc[f][t] = (rb_f32x4_t){0.f, 0.f, 0.f, 0.f};
                            });
                        });
                        }
                    }
                    c = 0;
                    blk += gridDim.x;
                    stamp(3);
                } else {
                    ++c;
                }
            } else {
                ++c;
            }
        });
    }
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int f = 0; f < NF; ++f) asm volatile("s_waitcnt vmcnt(0)" ::"v"(nx[r][f][0]), "v"(nx[r][f][1]) : "memory");
    if constexpr (PROF) {
        pt[4] = __builtin_readcyclecounter() - p_begin;
        if (lane == 0 && blockIdx.x < 256)
            for (int i = 0; i < 5; ++i) g_shadow2_prof[(blockIdx.x * 8 + wave) * 5 + i] = pt[i];
    }
    flush_records();
    if (lane == 0) stage_cnt[wave_slot] = wcount;
}

// Files the staged candidate records under their queries: cand[q, pos] = key with pos from cnt[q].  A block takes 16 wave
// slots, counts its records per query in LDS, reserves one range per (block, query) with a single global atomic and scatters.
// A staging slot that overflowed sets the top bit of cnt[0], which every consumer reads as "list overflowed".
__global__ __launch_bounds__(256) void regroup_candidates_kernel(const uint4* __restrict__ stage, unsigned stage_cap,
                                                                 unsigned* __restrict__ stage_cnt, int n_slots,
                                                                 unsigned long long* __restrict__ cand, int cap,
                                                                 unsigned* __restrict__ cnt, int n_q) {
    extern __shared__ unsigned rg_sh[];
    unsigned* hist = rg_sh;
    unsigned* base = rg_sh + n_q;
    unsigned* fill = rg_sh + 2 * n_q;
    __shared__ unsigned slot_n[16];
    const int tid = threadIdx.x, s0 = blockIdx.x * 16;
    for (int q = tid; q < n_q; q += 256) { hist[q] = 0u; fill[q] = 0u; }
    if (tid < 16) {
        unsigned n = s0 + tid < n_slots ? stage_cnt[s0 + tid] : 0u;
        if (s0 + tid < n_slots) stage_cnt[s0 + tid] = 0u;   // consumed: the next call's scan starts from clean slots (no memset launch)
        if (n > stage_cap) {
            atomicOr(&cnt[0], 0x80000000u);
            n = stage_cap;
        }
        slot_n[tid] = n;
    }
    __syncthreads();
    for (int sl = 0; sl < 16; ++sl) {
        const uint4* src = stage + (size_t)(s0 + sl) * stage_cap;
        for (unsigned i = tid; i < slot_n[sl]; i += 256)
            if (src[i].z < (unsigned)n_q) atomicAdd(&hist[src[i].z], 1u);  // (a record's query is always < n_q; never index LDS on trust)
    }
    __syncthreads();
    for (int q = tid; q < n_q; q += 256)
        if (hist[q]) base[q] = atomicAdd(&cnt[q], hist[q]) & 0x7FFFFFFFu;   // (another block may already have set cnt[0]'s overflow bit:
                                                                            //  taken as part of the position it filed query 0's records nowhere)
    __syncthreads();
    for (int sl = 0; sl < 16; ++sl) {
        const uint4* src = stage + (size_t)(s0 + sl) * stage_cap;
        for (unsigned i = tid; i < slot_n[sl]; i += 256) {
            const uint4 rec = src[i];
            if (rec.z >= (unsigned)n_q) continue;
            const unsigned pos = base[rec.z] + atomicAdd(&fill[rec.z], 1u);
            if (pos < (unsigned)cap) cand[rec.z * (size_t)cap + pos] = ((unsigned long long)rec.y << 32) | rec.x;
        }
    }
}

// Packed query image for the deep form: [d / 32 chunks][n_tiles][1 KiB] bf16, where the 1 KiB of (chunk c, tile T) is the LDS
// image of queries 16T .. 16T+15, k 32c .. 32c+31: 16-B unit L holds row L / 4, k-slot s = (L % 4) ^ (3 * ((L / 32) % 2)), i.e.
// k 4s..4s+3 and 16+4s..16+4s+3 (the k-order of the scan's database fragments; a dot product does not care).  Rows >= n_q are zero.
// The swizzle follows the lane groups ds_read_b128 is really served in on gfx950 — {0-3,12-15,20-27}, {4-11,16-19,28-31},
// {32-35,44-47,52-59}, {36-43,48-51,60-63}, NOT four runs of 16 lanes (MI355X_MICROARCH.md, LDS table): with lane = r16 + 16 g every
// group holds each row r16 once, rows 0-3 and 12-15 with one g and rows 4-11 with g ^ 1, and a unit's bank quad is
// (4 (r16 % 4) + position) % 16 — so the four rows of a group that share r16 % 4 (r16 / 4 = 0..3, reading g, g^1, g^1, g) must
// sit at four different positions: position = g ^ (0, 0, 3, 3)[r16 / 4].  Through round 4 the position was g ^ (r16 / 4), right for
// contiguous groups and 2-way conflicting for the real ones: rocprofv3 counted SQ_LDS_BANK_CONFLICT = 0.48 of SQ_LDS_IDX_ACTIVE on the
// scan (profiles/r05_dense_pmc.txt, "before").
__global__ void pack_queries_kernel(const float* __restrict__ q, int n_q, int d, int n_tiles, unsigned short* __restrict__ out) {
    const int64_t unit = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // one 16-B unit (8 bf16)
    const int64_t total = (int64_t)(d >> 5) * n_tiles * 64;
    if (unit >= total) return;
    const int L = (int)(unit & 63);
    const int64_t ct = unit >> 6;
    const int T = (int)(ct % n_tiles), c = (int)(ct / n_tiles);
    const int row = 16 * T + (L >> 2);
    const int k0 = 32 * c + 4 * ((L & 3) ^ (((L >> 5) & 1) * 3));  // slot s holds k 4s..4s+3 and 16+4s..16+4s+3: the order in which the
    uint4 v = make_uint4(0u, 0u, 0u, 0u);                     // scan's lanes read a database row (64 contiguous bytes per load)
    if (row < n_q) {
        const float* src = q + (int64_t)row * d + k0;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 16);
        v = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
    }
    *reinterpret_cast<uint4*>(out + unit * 8) = v;
}

// bf16 shadow of the database for the candidate scan (built once per index): row r, 32-k chunk c, 16-B unit g holds
// k = 32c + {4g..4g+3, 16+4g..16+4g+3} rounded to nearest even — exactly the fragment lane (r16, g) of the scan builds from the
// fp32 row, so the approximate scores are bit-identical to the fp32-database scan.  One wave per row; the wave also leaves
// max |row|^2 (fp32, lane-strided partial sums + a fixed xor tree: deterministic) for the completeness proof.
__global__ __launch_bounds__(256) void dense_shadow_kernel(const float* __restrict__ db, int64_t n_db, int d, uint4* __restrict__ out,
                                                           unsigned* __restrict__ max_norm2_bits, int tiled) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int units = d >> 3;  // 16-B output units per row
    // tiled (d % 256 == 0, what sim_candidates_shadow2_kernel reads): rows in blocks of 256; inside a block the 64-deep k-steps are
    // the OUTER index and a step is stored in the scan's FRAGMENT order — [block][step][16-row fragment][32-deep half][lane] x 16 B,
    // lane = row % 16 + 16 g — so a scan step of a block is one contiguous 32 KB and every load instruction of a wave reads 1 KiB
    // contiguous (row-major: 16 half lines 1 KB apart per instruction).  Rows past n_db in the last block are zero.
    auto unit_index = [&](int u) -> int64_t {
        if (!tiled) return row * units + u;
        const int c = u >> 2, g = u & 3;
        const int rb = (int)(row & 255);   // fragment rb / 16, half c % 2: 1 KiB in lane order (lane = row % 16 + 16 g)
        return (row >> 8) * (int64_t)(256 * units) + (int64_t)(c >> 1) * (256 * 8) + ((rb >> 4) * 2 + (c & 1)) * 64 + g * 16 + (rb & 15);
    };
    if (row >= n_db) {
        if (tiled && row < ((n_db + 255) >> 8 << 8))
            for (int u = lane; u < units; u += 64) out[unit_index(u)] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const float* src = db + row * d;
    float ss = 0.f;
    for (int u = lane; u < units; u += 64) {
        const int c = u >> 2, g = u & 3;
        const float4 a = *reinterpret_cast<const float4*>(src + 32 * c + 4 * g), b = *reinterpret_cast<const float4*>(src + 32 * c + 16 + 4 * g);
        ss = fmaf(a.x, a.x, ss); ss = fmaf(a.y, a.y, ss); ss = fmaf(a.z, a.z, ss); ss = fmaf(a.w, a.w, ss);
        ss = fmaf(b.x, b.x, ss); ss = fmaf(b.y, b.y, ss); ss = fmaf(b.z, b.z, ss); ss = fmaf(b.w, b.w, ss);
        out[unit_index(u)] = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) atomicMax(max_norm2_bits, __float_as_uint(ss));
}

}  // namespace swat
using namespace swat;

// Candidate pass, row-major queries (any d % 64 == 0): queries_bf16 [n_q, d] (bf16), db [n_db, d] fp32.  Every (q, row) with
// approximate score >= thr_dev[q] is appended to cand_keys [n_q, cap] (uint64: ordered(approx) << 32 | ~(row_base + row)),
// cand_cnt [n_q] (uint32, zero it first; counts keep growing past cap: overflow is visible).  max_norm2 (device float, zero it
// first) receives max |d_row|^2.  See include/swat_hip.h.
SWAT_API int swat_sim_candidates_bf16(const void* queries_bf16, int n_q, const float* db, int64_t n_db, int d,
                                      const float* thr_dev, int64_t row_base, void* cand_keys, int cap, void* cand_cnt,
                                      float* max_norm2, swat_stream_t s) {
    SWAT_REQUIRE(queries_bf16 && db && thr_dev && cand_keys && cand_cnt && max_norm2, "swat_sim_candidates_bf16: null pointer");
    SWAT_REQUIRE(d > 0 && d % 64 == 0, "swat_sim_candidates_bf16: d=%d must be a multiple of 64", d);
    SWAT_REQUIRE(n_q > 0 && n_db > 0, "swat_sim_candidates_bf16: empty operand");
    SWAT_REQUIRE(row_base + n_db < (1ll << 32), "swat_sim_candidates_bf16: global rows must fit 32 bits");
    SWAT_REQUIRE((int64_t)n_q * d < (1ll << 30), "swat_sim_candidates_bf16: query matrix too large for 32-bit offsets");
    hipStream_t st = (hipStream_t)s;
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
        if (left > 128) {  // 13 query tiles = 208 queries, 52 KB of LDS: three workgroups per CU, all resident
            SWAT_CAND(13, 1, 768)
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

// Packed query image of the deep form: bytes to allocate, and the packing itself (fp32 queries [n_q, d] -> bf16 image).
// The image carries 13 zero tiles of slack so that the last pass of swat_sim_candidates_packed never reads past it.
static int dense_tiles(int n_q) { return (n_q + 15) / 16 + 13; }

SWAT_API int64_t swat_dense_packed_bytes(int n_q, int d) {
    if (n_q <= 0 || d <= 0 || d % 128 != 0) return 0;
    return (int64_t)(d / 32) * dense_tiles(n_q) * 1024;
}

SWAT_API int swat_dense_pack_queries(const float* queries, int n_q, int d, void* packed, swat_stream_t s) {
    SWAT_REQUIRE(queries && packed, "swat_dense_pack_queries: null pointer");
    SWAT_REQUIRE(n_q > 0 && d > 0 && d % 128 == 0, "swat_dense_pack_queries: d=%d must be a multiple of 128", d);
    const int n_tiles = dense_tiles(n_q);
    const int64_t units = (int64_t)(d / 32) * n_tiles * 64;
    hipLaunchKernelGGL(pack_queries_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)s, queries, n_q, d, n_tiles,
                       (unsigned short*)packed);
    return check_launch("swat_dense_pack_queries");
}

// Staging area of the deep form: 3072 wave slots (768 workgroups x 4 waves) of 16-B records behind their counters.
static const int kDenseSlots = 3072;
static const int64_t kDenseCntBytes = 16384;

SWAT_API int64_t swat_dense_workspace_bytes(int n_q) {
    if (n_q <= 0 || n_q > 4096) return 0;
    int64_t per_slot = ((int64_t)3 * n_q + 63) / 64 * 64;  // ~6000 records per query in total: three times what the thresholds aim at
    if (per_slot < 256) per_slot = 256;
    return kDenseCntBytes + kDenseSlots * per_slot * 16;
}

// Candidate pass, deep form (d % 128 == 0, n_q <= 4096): as swat_sim_candidates_bf16 with the queries given as the packed
// image and a workspace of swat_dense_workspace_bytes(n_q) bytes for the per-wave staging of the candidate records.
static int g_dense_spread = -1;   // swat_dense_set_spread / SWAT_DENSE_SPREAD (tuning): row loads between the MFMAs (1, default) or at the step's start (0)
static int g_dense_epi = -1;   // swat_dense_set_epilogue / SWAT_DENSE_EPI (tuning): how the shadow scan's epilogue turns a hit into a record, 2 = vector code per row of the group, 3 = scalar walk over the hit bits (default)
static int dense_candidates_packed(const void* packed, int n_q, const float* db, bool shadow, int64_t n_db, int d, const float* thr_dev,
                                   int64_t row_base, void* cand_keys, int cap, void* cand_cnt, float* max_norm2,
                                   void* workspace, int64_t workspace_bytes, swat_stream_t s) {
    SWAT_REQUIRE(packed && db && thr_dev && cand_keys && cand_cnt && (max_norm2 || shadow) && workspace, "swat_sim_candidates_packed: null pointer");
    SWAT_REQUIRE(n_q <= 4096, "swat_sim_candidates_packed: n_q=%d above 4096", n_q);
    SWAT_REQUIRE(workspace_bytes >= swat_dense_workspace_bytes(n_q), "swat_sim_candidates_packed: workspace of %lld bytes is too small",
                 (long long)workspace_bytes);
    const unsigned stage_cap = (unsigned)((workspace_bytes - kDenseCntBytes) / (kDenseSlots * 16));
    unsigned* stage_cnt = (unsigned*)workspace;
    uint4* stage = (uint4*)((char*)workspace + kDenseCntBytes);
    // the slot counters: this call's regroup launch leaves them zero again, so only the form whose contract does not ask the caller
    // for a zeroed workspace pays the memset (the shadow form does: swat_hip.h)
    if (!shadow) SWAT_HIP(hipMemsetAsync(stage_cnt, 0, kDenseSlots * sizeof(unsigned), (hipStream_t)s));
    SWAT_REQUIRE(d > 0 && d % 128 == 0, "swat_sim_candidates_packed: d=%d must be a multiple of 128", d);
    SWAT_REQUIRE(n_q > 0 && n_db > 0, "swat_sim_candidates_packed: empty operand");
    SWAT_REQUIRE(row_base + n_db < (1ll << 32), "swat_sim_candidates_packed: global rows must fit 32 bits");
    hipStream_t st = (hipStream_t)s;
    const int n_tiles = dense_tiles(n_q);
    static const int abl = getenv("SWAT_DENSE_ABL") ? atoi(getenv("SWAT_DENSE_ABL")) : 0;  // tuning: timing ablations, wrong results
    static const int waves = getenv("SWAT_DENSE_WAVES") ? atoi(getenv("SWAT_DENSE_WAVES")) : 12;  // tuning: 4 = 64-row blocks, 3 workgroups per CU
    if (g_dense_epi < 0) g_dense_epi = getenv("SWAT_DENSE_EPI") ? atoi(getenv("SWAT_DENSE_EPI")) : 3;
    SWAT_REQUIRE(g_dense_epi == 2 || g_dense_epi == 3, "SWAT_DENSE_EPI: 2 or 3");
    const int epi = g_dense_epi;
    static const int qa = getenv("SWAT_DENSE_QA") ? atoi(getenv("SWAT_DENSE_QA")) : 2;   // tuning: query chunks 2 (default) or 3 steps ahead
    if (g_dense_spread < 0) g_dense_spread = getenv("SWAT_DENSE_SPREAD") ? atoi(getenv("SWAT_DENSE_SPREAD")) : 1;
    const int spread = g_dense_spread;
    static const int prof = getenv("SWAT_DENSE_PROF") ? atoi(getenv("SWAT_DENSE_PROF")) : 0;   // tuning: the cycle-accounting build of the product form
#define SWAT_CAND_DEEP(NQT_, ABL_)                                                                                       \
    if (shadow && d % 256 == 0) {   /* (the tiled shadow image has one reader) */                                        \
        const int64_t nblocks = (n_db + 255) / 256;                                                                      \
        const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);                                                 \
        auto kern = epi == 3 ? (qa == 3 ? sim_candidates_shadow2_kernel<NQT_, 3, 3, 0> : spread ? sim_candidates_shadow2_kernel<NQT_, 3, 2, 1> : sim_candidates_shadow2_kernel<NQT_, 3, 2, 0>) \
                             : (qa == 3 ? sim_candidates_shadow2_kernel<NQT_, 2, 3, 0> : spread ? sim_candidates_shadow2_kernel<NQT_, 2, 2, 1> : sim_candidates_shadow2_kernel<NQT_, 2, 2, 0>); \
        const int lds = 4 * 2 * 16 * NQT_ * 64 + 8 * SHADOW2_RCAP * 16 + (epi == 3 ? 8 * SHADOW2_LCAP * 20 : 0);                \
        if (prof) kern = sim_candidates_shadow2_kernel<NQT_, 3, 2, 1, true>;                                              \
        SWAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));               \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const unsigned short*)packed, n_tiles, n_q, q0,        \
                           (const unsigned short*)db, n_db, d, thr_dev, row_base, stage, stage_cap, stage_cnt);          \
    } else if (shadow) {                                                                                                 \
        const int64_t nblocks = (n_db + 191) / 192;                                                                      \
        const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);                                                 \
        hipLaunchKernelGGL((sim_candidates_bf16_deep_kernel<NQT_, 4, 12, ABL_, true>), dim3(grid), dim3(768), 0, st,     \
                           (const unsigned short*)packed, n_tiles, n_q, q0, db, n_db, d, thr_dev, row_base, stage,       \
                           stage_cap, stage_cnt, (unsigned*)max_norm2);                                                  \
    } else if (waves == 12) {                                                                                            \
        const int64_t nblocks = (n_db + 191) / 192;                                                                      \
        const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);                                                 \
        hipLaunchKernelGGL((sim_candidates_bf16_deep_kernel<NQT_, 4, 12, ABL_>), dim3(grid), dim3(768), 0, st,           \
                           (const unsigned short*)packed, n_tiles, n_q, q0, db, n_db, d, thr_dev, row_base, stage,       \
                           stage_cap, stage_cnt, (unsigned*)max_norm2);                                                  \
    } else {                                                                                                             \
        const int64_t nblocks = (n_db + 63) / 64;                                                                        \
        const unsigned grid = (unsigned)(nblocks < 768 ? nblocks : 768);                                                 \
        hipLaunchKernelGGL((sim_candidates_bf16_deep_kernel<NQT_, 4, 4, ABL_>), dim3(grid), dim3(256), 0, st,            \
                           (const unsigned short*)packed, n_tiles, n_q, q0, db, n_db, d, thr_dev, row_base, stage,       \
                           stage_cap, stage_cnt, (unsigned*)max_norm2);                                                  \
    }
    for (int q0 = 0; q0 < n_q;) {
        const int left = n_q - q0;
        if (left > 128) {  // 13 query tiles = 208 queries, 53 KB of LDS: three workgroups per CU, all resident
            switch (abl) {
                case 0: SWAT_CAND_DEEP(13, 0) break;
                case 1: SWAT_CAND_DEEP(13, 1) break;
                case 2: SWAT_CAND_DEEP(13, 2) break;
                case 4: SWAT_CAND_DEEP(13, 4) break;
                case 16: SWAT_CAND_DEEP(13, 16) break;
                default: SWAT_CAND_DEEP(13, 7) break;
            }
            q0 += 16 * 13;
        } else if (left > 32) {
            SWAT_CAND_DEEP(8, 0)
            q0 += 128;
        } else {
            SWAT_CAND_DEEP(2, 0)
            q0 += 32;
        }
    }
#undef SWAT_CAND_DEEP
    hipLaunchKernelGGL(regroup_candidates_kernel, dim3(kDenseSlots / 16), dim3(256), 3 * n_q * sizeof(unsigned), st, stage, stage_cap,
                       stage_cnt, kDenseSlots, (unsigned long long*)cand_keys, cap, (unsigned*)cand_cnt, n_q);
    return check_launch("swat_sim_candidates_packed");
}

SWAT_API int swat_dense_set_epilogue(int form) {
    SWAT_REQUIRE(form == 2 || form == 3, "swat_dense_set_epilogue: 2 or 3");
    g_dense_epi = form;
    return 0;
}

/* tuning (SWAT_DENSE_PROF=1): the last shadow scan's per-wave cycle sums, [256 workgroups][8 waves][wait, barrier, mfma phase, epilogue, kernel] */
SWAT_API int swat_dense_debug_profile(unsigned long long* host_out) {
    SWAT_REQUIRE(host_out, "swat_dense_debug_profile: null pointer");
    SWAT_HIP(hipDeviceSynchronize());
    SWAT_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_shadow2_prof), sizeof(unsigned long long) * 256 * 8 * 5));
    return 0;
}

SWAT_API int swat_dense_set_spread(int on) {
    SWAT_REQUIRE(on == 0 || on == 1, "swat_dense_set_spread: 0 or 1");
    g_dense_spread = on;
    return 0;
}

SWAT_API int swat_sim_candidates_packed(const void* packed, int n_q, const float* db, int64_t n_db, int d, const float* thr_dev,
                                        int64_t row_base, void* cand_keys, int cap, void* cand_cnt, float* max_norm2,
                                        void* workspace, int64_t workspace_bytes, swat_stream_t s) {
    return dense_candidates_packed(packed, n_q, db, false, n_db, d, thr_dev, row_base, cand_keys, cap, cand_cnt, max_norm2, workspace,
                                   workspace_bytes, s);
}

// The bf16 shadow of a database shard (see dense_shadow_kernel): bytes, the build (also max |row|^2 of the fp32 rows into
// *max_norm2, which the caller zeroes first), and the candidate pass that scans the shadow instead of the fp32 rows — same
// candidate lists bit for bit (same roundings), half the HBM bytes per row.
SWAT_API int64_t swat_dense_shadow_bytes(int64_t n_db, int d) {
    if (n_db <= 0 || d <= 0 || d % 128 != 0) return 0;
    if (d % 256 == 0) return ((n_db + 255) / 256 * 256) * d * 2;   // tiled image: whole blocks of 256 rows (dense_shadow_kernel)
    return n_db * d * 2;
}

SWAT_API int swat_dense_build_shadow(const float* db, int64_t n_db, int d, void* shadow, float* max_norm2, swat_stream_t s) {
    SWAT_REQUIRE(db && shadow && max_norm2, "swat_dense_build_shadow: null pointer");
    SWAT_REQUIRE(n_db > 0 && d > 0 && d % 128 == 0, "swat_dense_build_shadow: d=%d must be a multiple of 128", d);
    const int tiled = d % 256 == 0 ? 1 : 0;
    const int64_t rows = tiled ? (n_db + 255) / 256 * 256 : n_db;
    SWAT_REQUIRE((rows + 3) / 4 < (1ll << 31), "swat_dense_build_shadow: too many rows for one launch");
    hipLaunchKernelGGL(dense_shadow_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)s, db, n_db, d, (uint4*)shadow,
                       (unsigned*)max_norm2, tiled);
    return check_launch("swat_dense_build_shadow");
}

SWAT_API int swat_sim_candidates_shadow(const void* packed, int n_q, const void* shadow, int64_t n_db, int d, const float* thr_dev,
                                        int64_t row_base, void* cand_keys, int cap, void* cand_cnt, void* workspace,
                                        int64_t workspace_bytes, swat_stream_t s) {
    return dense_candidates_packed(packed, n_q, (const float*)shadow, true, n_db, d, thr_dev, row_base, cand_keys, cap, cand_cnt, nullptr,
                                   workspace, workspace_bytes, s);
}
