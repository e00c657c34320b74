// swat_amd/csrc/retrieval.hip — path B kernels for gfx950: similarity scan + filtered segmented top-k.
//
// ref (relative to the reference tree): retrieval/sample_retrieval.py:397-416, :335-353 (similarity),
// :754/:807 + :439-482 / :492-540 (rank + filter).  Semantics are pinned by oracle/retrieval_oracle.c.
#include "common.hpp"
#include "../../include/swat_hip.h"

namespace swat {

// --------------------------------------------------------------------------------------------
// B4: similarity scan.  HBM-bound: 4*d bytes per row in, 4 bytes out.
//
// Canonical arithmetic: ONE fp32 fma chain per row over k ascending.  To keep the chain on one
// lane while the HBM reads stay coalesced, each wave stages a [64 rows x KC floats] chunk
// through LDS: loads are 16 B/lane covering 4 rows x 256 B per wave-instruction, the LDS image
// has rows padded by one 16-B slot so both the ds_write_b128 (8 contiguous lanes) and the
// lane-per-row ds_read_b128 (16-lane groups -> 16 distinct slots) are conflict-free.
// The next chunk's loads are issued before the current chunk is consumed (register staging).
// Waves are independent (no workgroup barrier): 4 waves/block, 2 blocks/CU by LDS.
// --------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f4v_t;
__device__ __forceinline__ float4 nt_load4(const float* p) {
    const f4v_t v = __builtin_nontemporal_load(reinterpret_cast<const f4v_t*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}

constexpr int KC = 64;              // floats per row per chunk (256 B)
constexpr int ROW_F4 = KC / 4 + 1;  // padded row stride in float4

// GATHER: output row r scores database row row_index[r] (the exact re-scoring of the dense scan's candidates: the same
// k-ordered fma chain on a row list; row_index[r] < 0 -> -inf), n_rows = number of output rows.
template <bool GATHER>
__global__ __launch_bounds__(256) void sim_segmented_kernel(const float* __restrict__ feats, int64_t n_rows, int d,
                                                            const float* __restrict__ queries,
                                                            const int32_t* __restrict__ row_seg,
                                                            float* __restrict__ sims,
                                                            const int64_t* __restrict__ row_index,
                                                            unsigned long long* __restrict__ keys_out, int64_t row_base) {
    __shared__ float4 smem[4 * 64 * ROW_F4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float4* tile = smem + wave * 64 * ROW_F4;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (row0 >= n_rows) return;  // wave-uniform
    const int64_t last = n_rows - 1;
    const int64_t myrow = row0 + lane < last ? row0 + lane : last;
    // Rows are class-grouped, so a 64-row wave tile almost always sits inside ONE segment: then the query is
    // wave-uniform and is fetched through the scalar cache (s_load) instead of 16 more vector loads per chunk, which
    // would double the work of the CU's vector-memory front end on an otherwise pure HBM stream.
    const int64_t lastrow = row0 + 63 < last ? row0 + 63 : last;
    const int seg_lo = row_seg ? row_seg[row0] : 0;
    const int seg_hi = row_seg ? row_seg[lastrow] : 0;
    const bool uniform_q = __builtin_amdgcn_readfirstlane(seg_lo) == __builtin_amdgcn_readfirstlane(seg_hi);
    const float* q = queries + (row_seg ? (int64_t)row_seg[myrow] * d : 0);
    const float* qu = queries + (int64_t)__builtin_amdgcn_readfirstlane(seg_lo) * d;

    const int lr = lane >> 4;    // row inside a 4-row load group
    const int slot = lane & 15;  // 16-B slot inside the 256-B row chunk
    const float* src[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        int64_t r = row0 + 4 * j + lr;
        r = r < last ? r : last;
        if constexpr (GATHER) {
            r = row_index[r] - row_base;  // the list holds global rows (row_base + local), negative = padding
            r = r < 0 ? 0 : r;
        }
        src[j] = feats + r * (int64_t)d + slot * 4;
    }
    float4 stage[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) stage[j] = nt_load4(src[j]);

    float acc = 0.0f;
    const int nchunk = d / KC;
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
        for (int j = 0; j < 16; ++j) tile[(4 * j + lr) * ROW_F4 + slot] = stage[j];
        if (c + 1 < nchunk) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                stage[j] = nt_load4(src[j] + (c + 1) * KC);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (uniform_q) {
            const float* qc = qu + c * KC;  // wave-uniform address -> scalar loads; SGPR operand of v_fma
#pragma unroll
            for (int g = 0; g < KC / 4; ++g) {
                const float4 x = tile[lane * ROW_F4 + g];
                acc = fmaf(x.x, qc[4 * g + 0], acc);
                acc = fmaf(x.y, qc[4 * g + 1], acc);
                acc = fmaf(x.z, qc[4 * g + 2], acc);
                acc = fmaf(x.w, qc[4 * g + 3], acc);
            }
        } else {
            const float4* qc = reinterpret_cast<const float4*>(q + c * KC);
#pragma unroll
            for (int g = 0; g < KC / 4; ++g) {
                const float4 x = tile[lane * ROW_F4 + g];
                const float4 w = qc[g];
                acc = fmaf(x.x, w.x, acc);
                acc = fmaf(x.y, w.y, acc);
                acc = fmaf(x.z, w.z, acc);
                acc = fmaf(x.w, w.w, acc);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (GATHER) {
        if (row0 + lane < n_rows) {
            const int64_t gr = row_index[row0 + lane];
            if (gr < 0) acc = -INFINITY;
            if (sims) sims[row0 + lane] = acc;
            // (value desc, global row asc) sort key of the dense top-k (csrc/retrieval_mfma.hip: candidates_topk_kernel); 0 sorts last
            if (keys_out) {
                float sv = acc == 0.0f ? 0.0f : acc;
                const uint32_t u = __float_as_uint(sv);
                const uint32_t o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                keys_out[row0 + lane] = gr < 0 ? 0ull : (((unsigned long long)o << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)gr));
            }
        }
        return;
    }
    if (row0 + lane < n_rows) sims[row0 + lane] = acc;
}

// Completeness proof of the dense top-k (swat_amd/retrieval/dense.py): query q passes iff it kept `need` rows, no candidate
// list overflowed and its need-th exact value is strictly above bound_q + delta, bound_q = the K2-th best approximate score
// (or the pruning threshold when fewer than K2 candidates exist), delta = eps |q|max |d|max.  fails[0] = failing queries.
__global__ void dense_proof_kernel(const unsigned* __restrict__ cand_cnt, int cap, int K2, const float* __restrict__ approx_val,
                                   const float* __restrict__ thr, const float* __restrict__ exact_val, int k,
                                   const int64_t* __restrict__ out_cnt, int need, float eps_qn, const float* __restrict__ max_norm2,
                                   const float* __restrict__ qnorm_max, int n_q, int* __restrict__ fails) {
    const float delta = eps_qn * (qnorm_max ? qnorm_max[0] : 1.0f) * sqrtf(max_norm2[0]);
    int bad = 0;
    for (int q = threadIdx.x; q < n_q; q += blockDim.x) {
        const unsigned c = cand_cnt[q];
        const float bound = c >= (unsigned)K2 ? approx_val[(int64_t)q * K2 + K2 - 1] : thr[q];
        const bool ok = out_cnt[q] >= need && c <= (unsigned)cap && exact_val[(int64_t)q * k + need - 1] > bound + delta;
        bad += ok ? 0 : 1;
    }
    // one workgroup: the count is STORED (not added), so the flag may live in host-mapped pinned memory (no device atomics
    // there) and needs no reset between calls
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    if (bad) atomicAdd(&total, bad);
    __syncthreads();
    if (threadIdx.x == 0) fails[0] = total;
}

// --------------------------------------------------------------------------------------------
// B5: filtered segmented top-k.  One workgroup per class segment.
//   key(row) = ordered(sim) << 32 | (0xFFFFFFFF - local_row)   (0 = filtered out)
// so "larger key" == (sim desc, row asc): exactly CPython's stable sorted(reverse=True) order.
// MSB-first radix select (8 x 8 bit) finds the k-th largest key, a collect pass gathers the
// keys >= it into LDS, a bitonic sort orders them.  Keys are rebuilt from (sim, t2i, excluded)
// on every pass instead of being materialised (9 B/row/pass, L2-resident for typical classes).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ordered_bits(float s) {
    if (s == 0.0f) s = 0.0f;  // -0.0 == 0.0 for Python's sort: one key
    uint32_t u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered_bits(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

constexpr int TOPK_MAX = 2048;

__global__ __launch_bounds__(256) void segment_topk_kernel(const float* __restrict__ sims, const float* __restrict__ t2i,
                                                           const uint8_t* __restrict__ excluded,
                                                           const int64_t* __restrict__ seg_offsets, float thr,
                                                           float t2i_thr, int k, int64_t* __restrict__ out_idx,
                                                           float* __restrict__ out_val, int64_t* __restrict__ out_cnt) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long keys[TOPK_MAX];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned s_remaining, s_count, s_valid;

    const int seg = blockIdx.x;
    const int64_t b = seg_offsets[seg];
    const int64_t n = seg_offsets[seg + 1] - b;
    const int tid = threadIdx.x;

    auto make_key = [&](int64_t i) -> unsigned long long {
        const float s = sims[b + i];
        bool ok = s >= thr;
        if (t2i) ok = ok && (t2i[b + i] >= t2i_thr);
        if (excluded) ok = ok && (excluded[b + i] == 0);
        if (!ok) return 0ull;
        return ((unsigned long long)ordered_bits(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
    };

    // count valid rows
    if (tid == 0) { s_valid = 0; s_count = 0; }
    __syncthreads();
    unsigned local = 0;
    for (int64_t i = tid; i < n; i += 256) local += make_key(i) != 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += (unsigned)__shfl_xor((int)local, o, 64);
    if ((tid & 63) == 0) atomicAdd(&s_valid, local);
    __syncthreads();
    const unsigned n_valid = s_valid;
    const unsigned kk = n_valid < (unsigned)k ? n_valid : (unsigned)k;

    unsigned long long kth = 1ull;  // n_valid <= k: every valid key qualifies
    if (n_valid > (unsigned)k) {
        unsigned long long prefix = 0ull, mask = 0ull;
        unsigned remaining = (unsigned)k;
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            hist[tid] = 0;
            __syncthreads();
            for (int64_t i = tid; i < n; i += 256) {
                const unsigned long long key = make_key(i);
                if (key != 0ull && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFF], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                // wave 0 finds the digit that holds the remaining-th largest key: lane l owns digits 255-4l .. 252-4l; a
                // shuffle prefix gives the count of keys in larger digits (was: one thread walking 256 bins, 8 times)
                unsigned h[4], mine = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { h[j] = hist[255 - 4 * tid - j]; mine += h[j]; }
                unsigned incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
                    if (tid >= o) incl += up;
                }
                unsigned before = incl - mine;  // keys in digits above this lane's four
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
        kth = prefix;
    }

    // collect keys >= kth (exactly kk of them: keys are unique)
    int npow2 = 1;
    while (npow2 < (int)kk) npow2 <<= 1;
    for (int i = tid; i < npow2; i += 256) keys[i] = 0ull;
    __syncthreads();
    for (int64_t i = tid; i < n; i += 256) {
        const unsigned long long key = make_key(i);
        if (key != 0ull && key >= kth) {
            const unsigned pos = atomicAdd(&s_count, 1u);
            if (pos < (unsigned)TOPK_MAX) keys[pos] = key;
        }
    }
    __syncthreads();

    // bitonic sort, descending; stages of stride <= 64 stay inside the 128-key block a wave owns (no workgroup barrier:
    // the wave's LDS operations execute in order), see candidates_topk_kernel
    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 128) __syncthreads();
            for (int i = tid; i < npow2 / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = keys[lo], c = keys[hi];
                if ((a < c) == desc) { keys[lo] = c; keys[hi] = a; }
            }
            if (stride >= 128) {
                __syncthreads();
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();

    for (int r = tid; r < k; r += 256) {
        int64_t idx = -1;
        float val = -INFINITY;
        if (r < (int)kk) {
            const unsigned long long key = keys[r];
            idx = b + (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
            val = unordered_bits((uint32_t)(key >> 32));
        }
        out_idx[(int64_t)seg * k + r] = idx;
        if (out_val) out_val[(int64_t)seg * k + r] = val;
    }
    if (tid == 0) out_cnt[seg] = (int64_t)kk;
}

// rank-th largest of each row of S [nq][ns] (fp32; NaN-free), one workgroup per row.  Selection = bisection over the bits of the
// order-preserving integer image of the scores (the largest key K with #(key >= K) >= rank is the rank-th largest key), but not over the
// whole row: a thread keeps 16 keys in registers (ns <= 16 384); T0 = the rank-th largest of the 1 024 per-thread MAXIMA is a lower bound
// of the answer that only a few more than `rank` keys reach; wave 0 finds T0 (16 maxima per lane, 32 steps of 16 compares + five ballots
// over the bits of the lanes' counts, no barrier inside), every thread files its keys >= T0 in LDS, wave 0 bisects over those.  Longer
// rows, ranks beyond 1 024 or more than 1 024 survivors take the plain form: 32 steps over all keys with one LDS atomic per wave and a
// barrier per step.  The dense top-k's pruning threshold = the score at a small rank of a 16 k-row sample (swat_amd/retrieval/dense.py:
// _thresholds): 19-25 us against 62 for the two sorting launches it replaces (measured on the way: MSB radix select with LDS histograms
// 52 us — the scores share a few exponent bins and their atomics serialise; the plain form 28; wave-local selection 35 — all 16 waves of
// a row share a CU and the steps are VALU-issue and latency bound; what is left here is the two serial 32-step chains of wave 0).
// Rider (q != null): the workgroup of row r also leaves |q_r|^2 in norms2[r]; the LAST workgroup to finish (a self-resetting counter)
// writes qnorm_max[0] = max_r |q_r| rounded up by 2^-20 — the |q| factor of the top-k's error bound for a new query batch, without a
// launch of its own and without a host synchronisation.
constexpr int RS_THREADS = 1024, RS_KPT = 16;
__global__ __launch_bounds__(RS_THREADS) void rank_select_kernel(const float* __restrict__ S, int ns, int rank, float* __restrict__ out,
                                                                 const float* __restrict__ q, int d, float* __restrict__ norms2,
                                                                 unsigned* __restrict__ done_ctr, float* __restrict__ qnorm_max) {
    __shared__ unsigned total[32];
    __shared__ unsigned last_flag;
    const float* row = S + (int64_t)blockIdx.x * ns;
    const int tid = threadIdx.x, lane = tid & 63;
    auto to_key = [](float f) {
        const unsigned b = __float_as_uint(f);
        return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);   // larger float <-> larger unsigned
    };
    constexpr int KPT = RS_KPT;
    const bool in_regs = ns <= KPT * RS_THREADS;
    unsigned key[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = tid + RS_THREADS * u;
        key[u] = (in_regs && i < ns) ? to_key(row[i]) : 0u;   // (0 is below every candidate K >= 1: never counted)
    }
    if (tid < 32) total[tid] = 0;
    if (q && tid >= RS_THREADS - 64) {   // the last wave: |q_row|^2
        const float* qr = q + (int64_t)blockIdx.x * d;
        float ss = 0.f;
        for (int i = lane; i < d; i += 64) ss = fmaf(qr[i], qr[i], ss);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (lane == 0) norms2[blockIdx.x] = ss;
    }
    __syncthreads();
    unsigned prefix = 0;
    // Fast path (keys in registers, rank <= 1 024): the rank-th largest of the 1 024 per-thread maxima, T0, is a lower bound of the
    // answer that only a few more than `rank` keys reach (the row's best `rank` keys seldom share a thread); wave 0 finds T0 by
    // bisection over the maxima (16 per lane, no barrier inside), every thread files its keys >= T0 in LDS, wave 0 bisects over those.
    // More than 1 024 survivors (ties): the barrier form below answers.
    __shared__ unsigned pool[RS_THREADS];
    __shared__ unsigned pool_n, t0_key, result_key;
    bool done = false;
    auto wave0_select = [&](unsigned n) {   // rank-th largest of pool[0 .. n), n <= 1 024; by wave 0 only
        unsigned k2[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) k2[w] = (unsigned)(w * 64 + lane) < n ? pool[w * 64 + lane] : 0u;
        unsigned fp = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = fp | (1u << bit);
            unsigned c = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) c += k2[w] >= cand ? 1u : 0u;
            unsigned t = 0;
#pragma unroll
            for (int b = 0; b < 5; ++b) t += (unsigned)__popcll(__ballot((c >> b) & 1u)) << b;
            if (t >= (unsigned)rank) fp = cand;
        }
        return fp;
    };
    if (in_regs && rank <= RS_THREADS) {
        unsigned mx = key[0];
#pragma unroll
        for (int u = 1; u < KPT; ++u) mx = key[u] > mx ? key[u] : mx;
        pool[tid] = mx;
        if (tid == 0) pool_n = 0;
        __syncthreads();
        if (tid < 64) {
            const unsigned t0 = wave0_select(RS_THREADS);
            if (lane == 0) t0_key = t0;
        }
        __syncthreads();   // (also: every thread has read nothing of pool since the maxima went in; wave 0 is done with them)
        const unsigned t0 = t0_key;
        unsigned mine = 0;
#pragma unroll
        for (int u = 0; u < KPT; ++u) mine += key[u] >= t0 ? 1u : 0u;
        unsigned pos = 0;
        if (mine) pos = atomicAdd(&pool_n, mine);
        if (mine && pos + mine <= RS_THREADS) {
#pragma unroll
            for (int u = 0; u < KPT; ++u)
                if (key[u] >= t0) pool[pos++] = key[u];
        }
        __syncthreads();
        if (pool_n <= RS_THREADS) {
            if (tid < 64) {
                const unsigned r = wave0_select(pool_n);
                if (lane == 0) result_key = r;
            }
            __syncthreads();
            prefix = result_key;
            done = true;
        }
    }
    for (int bit = done ? -1 : 31; bit >= 0; --bit) {
        const unsigned cand = prefix | (1u << bit);
        unsigned c = 0;
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < KPT; ++u) c += key[u] >= cand ? 1u : 0u;
            unsigned t = 0;   // c <= 16: the wave's sum = sum over the count's bits of 2^b x (lanes with that bit set)
#pragma unroll
            for (int b = 0; b < 5; ++b) t += (unsigned)__popcll(__ballot((c >> b) & 1u)) << b;
            c = t;
        } else {
            for (int i = tid; i < ns; i += RS_THREADS) c += to_key(row[i]) >= cand ? 1u : 0u;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
        }
        if (lane == 0 && c) atomicAdd(&total[bit], c);
        __syncthreads();
        if (total[bit] >= (unsigned)rank) prefix = cand;
    }
    if (tid == 0) {
        const unsigned b = prefix ^ ((prefix >> 31) ? 0x80000000u : 0xFFFFFFFFu);
        out[blockIdx.x] = __uint_as_float(b);
    }
    if (q) {
        if (tid == 0) {
            __threadfence();   // norms2[row] is visible before the arrival
            last_flag = atomicAdd(done_ctr, 1u) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (last_flag) {   // every row's norm has been published
            float m = 0.f;
            for (int r = tid; r < (int)gridDim.x; r += RS_THREADS) m = fmaxf(m, __builtin_nontemporal_load(norms2 + r));
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            __shared__ float wm[RS_THREADS / 64];
            if (lane == 0) wm[tid >> 6] = m;
            __syncthreads();
            if (tid == 0) {
                float v = wm[0];
                for (int w = 1; w < RS_THREADS / 64; ++w) v = fmaxf(v, wm[w]);
                qnorm_max[0] = sqrtf(v) * (1.0f + 9.5367431640625e-07f);
                *done_ctr = 0u;   // ready for the next launch
            }
        }
    }
}

}  // namespace swat

SWAT_API int swat_sim_segmented(const float* feats, int64_t n_rows, int d, const float* queries,
                                const int32_t* row_seg, float* sims, swat_stream_t stream) {
    SWAT_REQUIRE(d > 0 && d % swat::KC == 0, "swat_sim_segmented: d=%d must be a positive multiple of %d", d, swat::KC);
    SWAT_REQUIRE(n_rows >= 0, "swat_sim_segmented: negative n_rows");
    if (n_rows == 0) return 0;
    SWAT_REQUIRE(feats && queries && sims, "swat_sim_segmented: null pointer");
    const int64_t blocks = (n_rows + 255) / 256;
    SWAT_REQUIRE(blocks < (1ll << 31), "swat_sim_segmented: too many rows for one launch");
    hipLaunchKernelGGL(swat::sim_segmented_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, feats,
                       n_rows, d, queries, row_seg, sims, (const int64_t*)nullptr, (unsigned long long*)nullptr, (int64_t)0);
    return swat::check_launch("swat_sim_segmented");
}

SWAT_API int swat_sim_gathered(const float* feats, const int64_t* row_index, int64_t n_idx, int d, const float* queries,
                               const int32_t* row_seg, float* sims, void* keys_out, int64_t row_base, swat_stream_t stream) {
    SWAT_REQUIRE(d > 0 && d % swat::KC == 0, "swat_sim_gathered: d=%d must be a positive multiple of %d", d, swat::KC);
    if (n_idx <= 0) return 0;
    SWAT_REQUIRE(feats && row_index && queries && (sims || keys_out), "swat_sim_gathered: null pointer");
    const int64_t blocks = (n_idx + 255) / 256;
    SWAT_REQUIRE(blocks < (1ll << 31), "swat_sim_gathered: too many rows for one launch");
    hipLaunchKernelGGL(swat::sim_segmented_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, feats,
                       n_idx, d, queries, row_seg, sims, row_index, (unsigned long long*)keys_out, row_base);
    return swat::check_launch("swat_sim_gathered");
}

SWAT_API int swat_dense_topk_proof(const void* cand_cnt, int cap, int K2, const float* approx_val, const float* thr_dev,
                                   const float* exact_val, int k, const int64_t* out_cnt, int need, float eps_qnorm,
                                   const float* max_norm2, const float* qnorm_max, int n_q, int* fails, swat_stream_t stream) {
    SWAT_REQUIRE(cand_cnt && approx_val && thr_dev && exact_val && out_cnt && max_norm2 && fails, "swat_dense_topk_proof: null pointer");
    SWAT_REQUIRE(n_q > 0 && need >= 1 && need <= k && K2 >= 1, "swat_dense_topk_proof: bad sizes");
    SWAT_HIP(hipMemsetAsync(fails, 0, sizeof(int), (hipStream_t)stream));
    hipLaunchKernelGGL(swat::dense_proof_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const unsigned*)cand_cnt, cap, K2,
                       approx_val, thr_dev, exact_val, k, out_cnt, need, eps_qnorm, max_norm2, qnorm_max, n_q, fails);
    return swat::check_launch("swat_dense_topk_proof");
}

SWAT_API int swat_segment_topk(const float* sims, const float* t2i, const uint8_t* excluded,
                               const int64_t* seg_offsets, int64_t n_seg, float thr, float t2i_thr, int k,
                               int64_t* out_idx, float* out_val, int64_t* out_cnt, swat_stream_t stream) {
    SWAT_REQUIRE(k >= 1 && k <= swat::TOPK_MAX, "swat_segment_topk: k=%d outside [1,%d]", k, swat::TOPK_MAX);
    SWAT_REQUIRE(n_seg >= 0 && n_seg < (1ll << 31), "swat_segment_topk: bad n_seg");
    if (n_seg == 0) return 0;
    SWAT_REQUIRE(sims && seg_offsets && out_idx && out_cnt, "swat_segment_topk: null pointer");
    hipLaunchKernelGGL(swat::segment_topk_kernel, dim3((unsigned)n_seg), dim3(256), 0, (hipStream_t)stream, sims, t2i,
                       excluded, seg_offsets, thr, t2i_thr, k, out_idx, out_val, out_cnt);
    return swat::check_launch("swat_segment_topk");
}

SWAT_API int swat_rank_select(const float* scores, int n_rows, int n_cols, int rank, float* out, const float* q_rows, int d,
                              void* norm_workspace, float* qnorm_max, swat_stream_t s) {
    SWAT_REQUIRE(scores && out && n_rows >= 0 && n_cols >= 1 && rank >= 1 && rank <= n_cols, "swat_rank_select: bad arguments");
    SWAT_REQUIRE(!q_rows || (d >= 1 && norm_workspace && qnorm_max), "swat_rank_select: the norm rider needs d, a workspace and an output");
    if (n_rows == 0) return 0;
    float* norms2 = q_rows ? (float*)norm_workspace + 1 : nullptr;   // [0]: the arrival counter (zero before the first use), [1 ..]: |q_r|^2
    hipLaunchKernelGGL(swat::rank_select_kernel, dim3(n_rows), dim3(swat::RS_THREADS), 0, (hipStream_t)s, scores, n_cols, rank, out, q_rows, d,
                       norms2, (unsigned*)norm_workspace, qnorm_max);
    return swat::check_launch("swat_rank_select");
}

