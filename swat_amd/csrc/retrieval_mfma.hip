// swat_amd/csrc/retrieval_mfma.hip — exact fp32 similarity products on the matrix cores (path B, dense form).
//
// gfx950's v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fp32 fma chain (MI355X guide, "FP32-input MFMA"), i.e.
// exactly the canonical order of oracle/retrieval_oracle.c: feeding k in ascending order makes the matrix-core result
// BIT-IDENTICAL to the lane-per-row scan kernel and to the CPU oracle.  Two users:
//   * swat_sim_topk_candidates — the (N_query x N_db) scan of BASELINE config 4: S = Q DB^T tile by tile, entries
//     >= a per-query threshold are appended (key = ordered(sim) << 32 | ~row) to a per-query candidate list;
//     swat_candidates_topk sorts a list and emits the (value desc, row asc) top-k.  Exact: the threshold comes from a
//     sample and the caller checks count >= k (else re-runs that query with a lower threshold).
//   * swat_near_dup_marks — remove_near_duplicates2 (ref retrieval/sample_retrieval.py:237-275): marks[j] = 1 iff
//     <f_i, f_j> > thr for some i < j, from S = F F^T.
// Tiling: block = 4 waves, 64 x 64 output tile (one 32x32 accumulator per wave), 64-float k-chunks staged through LDS
// with rows padded by one float so the lane-per-row ds_read_b32 operand fetches are conflict-free.  DB rows are
// streamed once per 64-query tile; fp32 MFMA peak (157 TF) bounds it once N_query > ~40.
#include "common.hpp"
#include "../../include/swat_hip.h"

namespace swat {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr int MK = 64;          // floats per k-chunk
constexpr int MLD = MK + 1;     // padded LDS row (floats)

__device__ __forceinline__ uint32_t ordered_bits_m(float s) {
    if (s == 0.0f) s = 0.0f;
    uint32_t u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// computes the 64x64 tile S[i0.., j0..] = A[i0.., :] . B[j0.., :]^T ; acc layout = v_mfma 32x32 C/D map:
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), for the wave's quadrant (wr, wc).
__device__ __forceinline__ void tile_product(const float* __restrict__ A, int64_t na, const float* __restrict__ B, int64_t nb,
                                             int d, int64_t i0, int64_t j0, float* sA, float* sB, f32x16_t& acc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < d; k0 += MK) {
        // stage 64 rows x 64 floats of each operand: 1024 float4 per operand, 4 per thread
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int c = tid + 256 * it;
            const int row = c >> 4, c4 = c & 15;
            int64_t ra = i0 + row; ra = ra < na ? ra : na - 1;
            int64_t rb = j0 + row; rb = rb < nb ? rb : nb - 1;
            const float4 va = *reinterpret_cast<const float4*>(A + ra * d + k0 + 4 * c4);
            const float4 vb = *reinterpret_cast<const float4*>(B + rb * d + k0 + 4 * c4);
            float* pa = sA + row * MLD + 4 * c4;
            float* pb = sB + row * MLD + 4 * c4;
            pa[0] = va.x; pa[1] = va.y; pa[2] = va.z; pa[3] = va.w;
            pb[0] = vb.x; pb[1] = vb.y; pb[2] = vb.z; pb[3] = vb.w;
        }
        __syncthreads();
        const float* la = sA + (wr + (lane & 31)) * MLD + (lane >> 5);
        const float* lb = sB + (wc + (lane & 31)) * MLD + (lane >> 5);
#pragma unroll 8
        for (int kk = 0; kk < MK; kk += 2)  // k ascending: the MFMA adds k = kk (lanes 0-31) then kk+1 (lanes 32-63)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(la[kk], lb[kk], acc, 0, 0, 0);
        __syncthreads();
    }
}

// grid: (ceil(n_db / 64), ceil(n_q / 64)).  A = queries (rows of S), B = db (columns of S).
__global__ __launch_bounds__(256) void sim_topk_candidates_kernel(const float* __restrict__ queries, int n_q,
                                                                  const float* __restrict__ db, int64_t n_db, int d,
                                                                  const float* __restrict__ thr, int64_t row_base,
                                                                  unsigned long long* __restrict__ cand, int cap,
                                                                  unsigned* __restrict__ cnt, float* __restrict__ S_out,
                                                                  int64_t ld_s) {
    __shared__ float sA[64 * MLD];
    __shared__ float sB[64 * MLD];
    const int64_t j0 = (int64_t)blockIdx.x * 64;
    const int64_t i0 = (int64_t)blockIdx.y * 64;
    f32x16_t acc;
    tile_product(queries, n_q, db, n_db, d, i0, j0, sA, sB, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t col = j0 + (wave & 1) * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = i0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < n_q && col < n_db) {
            const float s = acc[r];
            if (S_out) S_out[row * ld_s + col] = s;
            if (cand && s >= thr[row]) {
                const unsigned pos = atomicAdd(&cnt[row], 1u);
                if (pos < (unsigned)cap)
                    cand[row * (int64_t)cap + pos] = ((unsigned long long)ordered_bits_m(s) << 32) |
                                                     (unsigned long long)(0xFFFFFFFFu - (uint32_t)(row_base + col));
            }
        }
    }
}

// Wide form for the query-against-database scan: a workgroup takes 128 DB rows against a whole group of up to 224
// queries (7 tiles of 32), so the database is streamed from HBM ONCE per group instead of once per 64 queries and every
// DB operand fetched from LDS feeds 7 MFMAs.  Wave w owns DB rows 32w..32w+31 of the tile and all 7 query tiles (7
// accumulators = 112 AGPRs).  16-float k-chunks go global -> LDS by global_load_lds_dwordx4 into two stage buffers
// (the next chunk lands while this one is multiplied; one barrier per chunk; 45 KB of LDS: three workgroups per CU).
// LDS rows are 64 B with the 16-B slot index XORed with (row >> 2) & 3 — applied to the per-lane SOURCE address, the
// DMA destination being lane-linear — which leaves the lane-per-row ds_read_b32 operand fetches 2-way conflicted.
// Same k-ascending accumulation per output element as tile_product: bit-identical results.
constexpr int WQ = 7;                       // query tiles of 32 per group
constexpr int WK = 16;                      // floats per k-chunk
constexpr int WROWS = WQ * 32 + 128;        // rows per stage: queries then database
constexpr int WSTAGE = WROWS * WK * 4;      // bytes
constexpr int WPIECES = WROWS / 16;         // 1 KB DMA pieces per stage (16 rows each): 22, dealt 6 / 6 / 5 / 5 to the waves
constexpr int WPW = (WPIECES + 3) / 4;
__global__ __launch_bounds__(256) void sim_topk_candidates_wide_kernel(const float* __restrict__ queries, int n_q,
                                                                       const float* __restrict__ db, int64_t n_db, int d,
                                                                       const float* __restrict__ thr, int64_t row_base,
                                                                       unsigned long long* __restrict__ cand, int cap,
                                                                       unsigned* __restrict__ cnt, float* __restrict__ S_out,
                                                                       int64_t ld_s) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * WSTAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t j0 = (int64_t)blockIdx.x * 128;
    const int q0 = blockIdx.y * (WQ * 32);
    // per-lane source pointers of this wave's pieces (piece p covers stage rows 16p..16p+15; lane -> row 16p + lane/4,
    // LDS slot lane%4 which must hold the 16-B chunk (slot ^ f(row)) of that row)
    const float* src[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        int row = 16 * (wave + 4 * i) + (lane >> 2);
        row = row < WROWS ? row : WROWS - 1;
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        if (row < WQ * 32) {
            int rq = q0 + row; rq = rq < n_q ? rq : n_q - 1;
            src[i] = queries + (int64_t)rq * d + 4 * chunk;
        } else {
            int64_t rb = j0 + (row - WQ * 32); rb = rb < n_db ? rb : n_db - 1;
            src[i] = db + rb * d + 4 * chunk;
        }
    }
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int i = 0; i < WPW; ++i)
            if (wave + 4 * i < WPIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(smem + stage * WSTAGE + (wave + 4 * i) * 1024),
                                                 16, 0, 0);
    };
    f32x16_t acc[WQ];
#pragma unroll
    for (int t = 0; t < WQ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int r32 = lane & 31, h = lane >> 5, fr = (r32 >> 2) & 3;
    const int nchunk = d / WK;
    issue(0, 0);
    for (int c = 0; c < nchunk; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk c has landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                       // ... everyone's; and everyone is done reading chunk c-1
        if (c + 1 < nchunk) issue((c + 1) & 1, (c + 1) * WK);
        const float* sq = reinterpret_cast<const float*>(smem + (c & 1) * WSTAGE) + r32 * WK;
        const float* sd = reinterpret_cast<const float*>(smem + (c & 1) * WSTAGE) + (WQ * 32 + wave * 32 + r32) * WK;
#pragma unroll
        for (int kk = 0; kk < WK; kk += 2) {  // k ascending: lanes 0-31 add k = kk, lanes 32-63 k = kk + 1
            const int off = (((kk >> 2) ^ fr) << 2) + (kk & 3) + h;
            const float b = sd[off];
#pragma unroll
            for (int t = 0; t < WQ; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sq[t * 32 * WK + off], b, acc[t], 0, 0, 0);
        }
    }
    const int64_t col = j0 + wave * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < WQ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = q0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < n_q && col < n_db) {
                const float s = acc[t][r];
                if (S_out) S_out[(int64_t)row * ld_s + col] = s;
                if (cand && s >= thr[row]) {
                    const unsigned pos = atomicAdd(&cnt[row], 1u);
                    if (pos < (unsigned)cap)
                        cand[row * (int64_t)cap + pos] = ((unsigned long long)ordered_bits_m(s) << 32) |
                                                         (unsigned long long)(0xFFFFFFFFu - (uint32_t)(row_base + col));
                }
            }
        }
}

// marks[j] = 1 iff exists i < j with <f_i, f_j> > thr.  grid over the upper-triangular tiles (x = j tile >= y = i tile).
__global__ __launch_bounds__(256) void near_dup_kernel(const float* __restrict__ F, int64_t n, int d, float thr,
                                                       uint8_t* __restrict__ marks) {
    if (blockIdx.x < blockIdx.y) return;
    __shared__ float sA[64 * MLD];
    __shared__ float sB[64 * MLD];
    const int64_t j0 = (int64_t)blockIdx.x * 64, i0 = (int64_t)blockIdx.y * 64;
    f32x16_t acc;
    tile_product(F, n, F, n, d, i0, j0, sA, sB, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t col = j0 + (wave & 1) * 32 + (lane & 31);
    bool hit = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = i0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < col && col < n && acc[r] > thr) hit = true;
    }
    if (hit) marks[col] = 1;  // every writer stores the same value
}

// One block per query: sort its candidate keys (<= CAND_MAX) descending, emit the first k.  Bitonic network in LDS on 1024
// threads; a compare-exchange stage of stride <= 64 stays inside the 128-key block its wave owns (pairs i0 .. i0+63 touch
// keys 2 i0 .. 2 i0 + 127), so only the stages of stride >= 128 — 10 of the 66 for 2048 keys — pay a workgroup barrier; the
// others are ordered by the wave's own in-order LDS queue.
constexpr int CAND_MAX = 8192;
// The completeness proof of the dense top-k, folded into the last sort of a query (swat_candidates_topk_proof): see dense_proof_kernel
// in retrieval.hip for the condition; fails is a device counter some earlier kernel of the call has zeroed.
struct TopkProof {
    const unsigned* cand_cnt;   // candidates the scan found per query (null: no proof)
    const float* approx_val;    // [n_q, K2]: slot K2 - 1 = the K2-th best approximate score
    const float* thr;           // pruning thresholds
    const float* max_norm2;     // max |row|^2
    const float* qnorm_max;     // max |q|
    int* fails;
    int scan_cap, K2, need;
    float eps;
    unsigned* zero_cnt;         // not null: cnt[q] = 0 once read (the next call's regroup adds into clean counters)
};

__global__ __launch_bounds__(1024) void candidates_topk_kernel(const unsigned long long* __restrict__ cand, int cap,
                                                               const unsigned* __restrict__ cnt, int k,
                                                               int64_t* __restrict__ out_idx, float* __restrict__ out_val,
                                                               int64_t* __restrict__ out_cnt, const TopkProof pr) {
    __shared__ unsigned long long keys[CAND_MAX];
    const int q = blockIdx.x, tid = threadIdx.x;
    unsigned n = cnt[q] & 0x7FFFFFFFu;   // bit 31 of cnt[0] = "a staging slot overflowed" (the proof reads it); the entries beyond the
    if (n > (unsigned)cap) n = cap;      // count were never written
    int np2 = 1;
    while (np2 < (int)n) np2 <<= 1;
    for (int i = tid; i < np2; i += 1024) keys[i] = i < (int)n ? cand[q * (int64_t)cap + i] : 0ull;
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 128) __syncthreads();  // reads keys other waves wrote
            for (int i = tid; i < np2 / 2; i += 1024) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = keys[lo], c = keys[hi];
                if ((a < c) == desc) { keys[lo] = c; keys[hi] = a; }
            }
            if (stride >= 128) {
                __syncthreads();
            } else {  // wave-local stage: the wave's LDS operations execute in order; keep the compiler from caching keys
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();
    const int kk = (int)n < k ? (int)n : k;
    int valid = 0;  // key 0 = an empty slot (merged shard lists carry -1 padded rows): sorts last, emitted as (-1, -inf)
    for (int r = tid; r < k; r += 1024) {
        int64_t idx = -1;
        float val = -INFINITY;
        if (r < kk && keys[r] != 0ull) {
            const unsigned long long key = keys[r];
            idx = (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
            const uint32_t o = (uint32_t)(key >> 32);
            val = __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
            ++valid;
        }
        out_idx[(int64_t)q * k + r] = idx;
        out_val[(int64_t)q * k + r] = val;
    }
    __shared__ int n_valid;
    if (tid == 0) n_valid = 0;
    __syncthreads();
    if (valid) atomicAdd(&n_valid, valid);
    __syncthreads();
    if (tid == 0) {
        out_cnt[q] = n_valid;
        if (pr.cand_cnt) {
            const unsigned c = pr.cand_cnt[q];
            const float bound = c >= (unsigned)pr.K2 ? pr.approx_val[(int64_t)q * pr.K2 + pr.K2 - 1] : pr.thr[q];
            const float delta = pr.eps * pr.qnorm_max[0] * sqrtf(pr.max_norm2[0]);
            float kth_val = -INFINITY;
            if (n_valid >= pr.need) {
                const unsigned long long key = keys[pr.need - 1];
                const uint32_t o = (uint32_t)(key >> 32);
                kth_val = __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
            }
            const bool ok = n_valid >= pr.need && c <= (unsigned)pr.scan_cap && kth_val > bound + delta;
            if (!ok) atomicAdd(pr.fails, 1);
            if (pr.zero_cnt) pr.zero_cnt[q] = 0u;
        }
    }
}

// One block per query: the k2 LARGEST of its candidate keys, UNORDERED — what the exact re-scoring step needs (it re-sorts by the
// exact value anyway), at a third of the full sort's time: the keys sit in LDS, an MSB-first radix select (8 x 8 bits over the
// 64-bit key) finds the k2-th largest one, a collect pass writes the keys above it to slots 0 .. k2-2 (in arrival order) and the
// k2-th key itself to slot k2-1, where the completeness proof reads "the K2-th best approximate score".  Fewer than k2
// candidates: all of them, padded with (-1, -inf).
__global__ __launch_bounds__(1024) void candidates_select_kernel(const unsigned long long* __restrict__ cand, int cap,
                                                                 const unsigned* __restrict__ cnt, int k2,
                                                                 int64_t* __restrict__ out_idx, float* __restrict__ out_val,
                                                                 int64_t* __restrict__ out_cnt, int* __restrict__ zero_me) {
    __shared__ unsigned long long keys[CAND_MAX];
    __shared__ unsigned hist[256];
    if (zero_me && blockIdx.x == 0 && threadIdx.x == 0) *zero_me = 0;   // the failure counter of this call (read two launches later)
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned s_remaining, s_count;
    const int q = blockIdx.x, tid = threadIdx.x;
    unsigned n = cnt[q] & 0x7FFFFFFFu;   // (bit 31 of cnt[0] = a staging slot overflowed: the proof sees it through cap)
    if (n > (unsigned)cap) n = cap;
    for (int i = tid; i < (int)n; i += 1024) keys[i] = cand[q * (int64_t)cap + i];
    if (tid == 0) s_count = 0;
    __syncthreads();
    const int kk = (int)n < k2 ? (int)n : k2;
    unsigned long long kth = 0ull;   // n <= k2: every key qualifies (keys are non-zero)
    if ((int)n > k2) {
        unsigned long long prefix = 0ull, mask = 0ull;
        unsigned remaining = (unsigned)k2;
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < (int)n; i += 1024) {
                const unsigned long long key = keys[i];
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFF], 1u);
            }
            __syncthreads();
            if (tid < 64) {   // lane l owns digits 255-4l .. 252-4l; a shuffle prefix counts the keys in larger digits
                unsigned h[4], mine = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { h[j] = hist[255 - 4 * tid - j]; mine += h[j]; }
                unsigned incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
                    if (tid >= o) incl += up;
                }
                unsigned before = incl - mine;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (before < remaining && before + h[j] >= remaining) {  // true for exactly one (lane, j)
                        s_remaining = remaining - before;
                        s_prefix = prefix | ((unsigned long long)(255 - 4 * tid - j) << shift);
                    }
                    before += h[j];
                }
            }
            __syncthreads();
            remaining = s_remaining;
            prefix = s_prefix;
            mask |= 0xFFull << shift;
            __syncthreads();
        }
        kth = prefix;   // keys are unique: exactly k2 - 1 keys are larger
    }
    auto emit = [&](int pos, unsigned long long key) {
        const uint32_t o = (uint32_t)(key >> 32);
        out_idx[(int64_t)q * k2 + pos] = (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
        out_val[(int64_t)q * k2 + pos] = __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
    };
    for (int i = tid; i < (int)n; i += 1024) {
        const unsigned long long key = keys[i];
        if (key > kth) {
            const unsigned pos = atomicAdd(&s_count, 1u);
            if ((int)pos < k2) emit((int)pos, key);
        } else if (key == kth && kth != 0ull) {
            emit(k2 - 1, key);
        }
    }
    for (int r = kk + tid; r < k2; r += 1024) {
        out_idx[(int64_t)q * k2 + r] = -1;
        out_val[(int64_t)q * k2 + r] = -INFINITY;
    }
    if (tid == 0) out_cnt[q] = kk;
}

// (value, global row) lists -> the sort keys candidates_topk_kernel orders: ordered(value) << 32 | ~row; row < 0 -> 0 (sorts
// last).  -0.0 and 0.0 are one key, as in segment_topk_kernel.
__global__ void pack_list_keys_kernel(const float* __restrict__ val, const int64_t* __restrict__ idx, int64_t n,
                                      unsigned long long* __restrict__ keys, unsigned* __restrict__ cnt, int n_q, int m) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_q) cnt[i] = (unsigned)m;
    if (i >= n) return;
    const int64_t r = idx[i];
    float v = val[i];
    if (v == 0.0f) v = 0.0f;
    const uint32_t b = __float_as_uint(v);
    const uint32_t o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    keys[i] = r < 0 ? 0ull : (((unsigned long long)o << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)r));
}

}  // namespace swat
using namespace swat;

// k-way merge of per-shard top-k lists (SURVEY 8(e) retrieval row: the rank-0 / every-rank merge after the all_gather):
// idx / val [n_q, m] = the shards' lists side by side (global rows, -1 padded), m <= 8192.  Output: the k best by
// (value desc, global row asc).  workspace: swat_merge_topk_workspace_bytes(n_q, m).
SWAT_API int64_t swat_merge_topk_workspace_bytes(int n_q, int m) {
    return (int64_t)n_q * m * 8 + (int64_t)n_q * 4 + 256;
}

SWAT_API int swat_merge_topk_lists(const int64_t* idx, const float* val, int n_q, int m, int k, void* workspace,
                                   int64_t* out_idx, float* out_val, int64_t* out_cnt, swat_stream_t s) {
    SWAT_REQUIRE(idx && val && workspace && out_idx && out_val && out_cnt, "swat_merge_topk_lists: null pointer");
    SWAT_REQUIRE(n_q >= 1 && m >= 1 && m <= CAND_MAX && k >= 1 && k <= m, "swat_merge_topk_lists: need 1 <= k <= m <= %d", CAND_MAX);
    unsigned long long* keys = (unsigned long long*)workspace;
    unsigned* cnt = (unsigned*)((char*)workspace + ((int64_t)n_q * m * 8 + 255) / 256 * 256);
    const int64_t n = (int64_t)n_q * m;
    hipLaunchKernelGGL(pack_list_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, val, idx, n, keys, cnt, n_q, m);
    hipLaunchKernelGGL(candidates_topk_kernel, dim3(n_q), dim3(1024), 0, (hipStream_t)s, keys, m, cnt, k, out_idx, out_val, out_cnt, TopkProof{});
    return check_launch("swat_merge_topk_lists");
}

SWAT_API int swat_sim_topk_candidates(const float* queries, int n_q, const float* db, int64_t n_db, int d,
                                      const float* thr_dev, int64_t row_base, void* cand_keys, int cap, void* cand_cnt,
                                      float* S_out, int64_t ld_s, swat_stream_t s) {
    SWAT_REQUIRE(d > 0 && d % MK == 0, "swat_sim_topk_candidates: d=%d must be a multiple of %d", d, MK);
    SWAT_REQUIRE(n_q > 0 && n_db > 0, "swat_sim_topk_candidates: empty operand");
    SWAT_REQUIRE(row_base + n_db < (1ll << 32), "swat_sim_topk_candidates: global rows must fit 32 bits");
    SWAT_REQUIRE((cand_keys && cand_cnt && thr_dev) || S_out, "swat_sim_topk_candidates: nothing to produce");
    if (n_q > 64) {  // many queries: stream the database once per 224 of them
        dim3 grid((unsigned)((n_db + 127) / 128), (unsigned)((n_q + WQ * 32 - 1) / (WQ * 32)));
        hipLaunchKernelGGL(sim_topk_candidates_wide_kernel, grid, dim3(256), 0, (hipStream_t)s, queries, n_q, db, n_db, d,
                           thr_dev, row_base, (unsigned long long*)cand_keys, cap, (unsigned*)cand_cnt, S_out, ld_s);
    } else {
        dim3 grid((unsigned)((n_db + 63) / 64), (unsigned)((n_q + 63) / 64));
        hipLaunchKernelGGL(sim_topk_candidates_kernel, grid, dim3(256), 0, (hipStream_t)s, queries, n_q, db, n_db, d, thr_dev,
                           row_base, (unsigned long long*)cand_keys, cap, (unsigned*)cand_cnt, S_out, ld_s);
    }
    return check_launch("swat_sim_topk_candidates");
}

SWAT_API int swat_candidates_topk(const void* cand_keys, int cap, const void* cand_cnt, int n_q, int k, int64_t* out_idx,
                                  float* out_val, int64_t* out_cnt, swat_stream_t s) {
    SWAT_REQUIRE(cap >= 1 && cap <= CAND_MAX && k >= 1 && k <= cap, "swat_candidates_topk: need 1 <= k <= cap <= %d", CAND_MAX);
    hipLaunchKernelGGL(candidates_topk_kernel, dim3(n_q), dim3(1024), 0, (hipStream_t)s, (const unsigned long long*)cand_keys,
                       cap, (const unsigned*)cand_cnt, k, out_idx, out_val, out_cnt, TopkProof{});
    return check_launch("swat_candidates_topk");
}

// swat_candidates_topk over the exactly re-scored keys + swat_dense_topk_proof in the same launch (one block per query: its thread 0
// checks the query once its list is sorted).  fails: DEVICE counter, zeroed by the swat_candidates_select launch of the same call
// (its zero_fails argument) or by the caller; scan_cnt / scan_cap: what the candidate pass counted / its list capacity.
SWAT_API int swat_candidates_topk_proof(const void* cand_keys, int cap, const void* cand_cnt, int n_q, int k, int64_t* out_idx,
                                        float* out_val, int64_t* out_cnt, void* scan_cnt, int scan_cap, int K2,
                                        const float* approx_val, const float* thr_dev, int need, float eps, const float* max_norm2,
                                        const float* qnorm_max, int* fails, int zero_scan_cnt, swat_stream_t s) {
    SWAT_REQUIRE(cap >= 1 && cap <= CAND_MAX && k >= 1 && k <= cap, "swat_candidates_topk_proof: need 1 <= k <= cap <= %d", CAND_MAX);
    SWAT_REQUIRE(scan_cnt && approx_val && thr_dev && max_norm2 && qnorm_max && fails && need >= 1 && need <= k && K2 >= 1,
                 "swat_candidates_topk_proof: null pointer or bad need / K2");
    TopkProof pr{(const unsigned*)scan_cnt, approx_val, thr_dev, max_norm2, qnorm_max, fails, scan_cap, K2, need, eps,
                 zero_scan_cnt ? (unsigned*)scan_cnt : nullptr};
    hipLaunchKernelGGL(candidates_topk_kernel, dim3(n_q), dim3(1024), 0, (hipStream_t)s, (const unsigned long long*)cand_keys,
                       cap, (const unsigned*)cand_cnt, k, out_idx, out_val, out_cnt, pr);
    return check_launch("swat_candidates_topk_proof");
}

SWAT_API int swat_candidates_select(const void* cand_keys, int cap, const void* cand_cnt, int n_q, int k2, int64_t* out_idx,
                                    float* out_val, int64_t* out_cnt, int* zero_fails, swat_stream_t s) {
    SWAT_REQUIRE(cap >= 1 && cap <= CAND_MAX && k2 >= 1 && k2 <= cap, "swat_candidates_select: need 1 <= k2 <= cap <= %d", CAND_MAX);
    hipLaunchKernelGGL(candidates_select_kernel, dim3(n_q), dim3(1024), 0, (hipStream_t)s, (const unsigned long long*)cand_keys,
                       cap, (const unsigned*)cand_cnt, k2, out_idx, out_val, out_cnt, zero_fails);
    return check_launch("swat_candidates_select");
}

SWAT_API int swat_near_dup_marks(const float* feats, int64_t n, int d, float thr, uint8_t* marks, swat_stream_t s) {
    SWAT_REQUIRE(d > 0 && d % MK == 0, "swat_near_dup_marks: d=%d must be a multiple of %d", d, MK);
    if (n <= 0) return 0;
    SWAT_HIP(hipMemsetAsync(marks, 0, (size_t)n, (hipStream_t)s));
    const unsigned t = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(near_dup_kernel, dim3(t, t), dim3(256), 0, (hipStream_t)s, feats, n, d, thr, marks);
    return check_launch("swat_near_dup_marks");
}
