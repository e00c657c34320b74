// swat_amd/csrc/tail.hip — the fp32 "tail" of path A on gfx950: L2-normalise (A3), MyLinear head + logit
// scale (A4), (CutMix-)mixed cross entropy (A6) and their backward, plus the small strided fp32 GEMM they use.
//
// ref: utils/training.py:333-338,1475-1481 (normalise, head, scale, loss), :1157-1158 (mixup_criterion),
//      utils/models.py:47-72 (MyLinear), utils/losses.py:10-11 (nn.CrossEntropyLoss, mean reduction).
// M = batch (32..256) here, so these are latency-bound VALU kernels; everything stays fp32 (the master
// weights of proj / head are read directly).
#include "kernels.hpp"
#include "../../include/swat_hip.h"

namespace swat {

// C[m,n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]  (+ C if accumulate).
// Block = one output row m, 64 columns x 4 k-lanes: lane j runs an fma chain over k = j, j+4, ...; the four partial
// sums are added in lane order -> deterministic.
__global__ __launch_bounds__(256) void sgemm_small_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K, int64_t sam, int64_t sak,
                                                          int64_t sbk, int64_t sbn, int64_t ldc, float alpha, int accumulate) {
    __shared__ float sh[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int m = blockIdx.y;
    const int j = threadIdx.x >> 6;
    float acc = 0.f;
    if (n < N) {
        const float* a = A + (int64_t)m * sam;
        const float* b = B + (int64_t)n * sbn;
        for (int k = j; k < K; k += 4) acc = fmaf(a[(int64_t)k * sak], b[(int64_t)k * sbk], acc);
    }
    sh[j][threadIdx.x & 63] = acc;
    __syncthreads();
    if (j == 0 && n < N) {
        const int t = threadIdx.x;
        const float v = alpha * (((sh[0][t] + sh[1][t]) + sh[2][t]) + sh[3][t]);
        float* c = C + (int64_t)m * ldc + n;
        *c = accumulate ? *c + v : v;
    }
}

// y = x / ||x||_2 per row (no eps — ref utils/training.py:333); one wave per row; norm saved.
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ nrm, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) { const float v = x[(int64_t)row * d + c]; s += v * v; }
    const float n = sqrtf(wave_sum(s));
    if (lane == 0 && nrm) nrm[row] = n;
    for (int c = lane; c < d; c += 64) y[(int64_t)row * d + c] = x[(int64_t)row * d + c] / n;
}

// dx = (dy - y * <y, dy>) / ||x||
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ nrm,
                                                         const float* __restrict__ dy, float* __restrict__ dx, int rows,
                                                         int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += y[(int64_t)row * d + c] * dy[(int64_t)row * d + c];
    s = wave_sum(s);
    const float inv = 1.0f / nrm[row];
    for (int c = lane; c < d; c += 64)
        dx[(int64_t)row * d + c] = (dy[(int64_t)row * d + c] - y[(int64_t)row * d + c] * s) * inv;
}

// Mixed cross entropy on scaled logits.  z [B,C] = head output (un-scaled); logits = z * exp(ls).
//   loss_b = lam * CE(logits_b, ya_b) + (1-lam) * CE(logits_b, yb_b)      (ref :1157-1158)
//   dz_b   = exp(ls) * (softmax(logits_b) - lam*1[ya_b] - (1-lam)*1[yb_b]) / B        (mean reduction)
//   dls_b  = sum_c dlogits_bc * logits_bc
// One block (256 threads) per sample.  per_loss[b], per_dls[b] are reduced by ce_finish_kernel.
__global__ __launch_bounds__(256) void ce_mixed_kernel(const float* __restrict__ z, const float* __restrict__ ls_ptr,
                                                       const int64_t* __restrict__ ya, const int64_t* __restrict__ yb, float lam,
                                                       float* __restrict__ per_loss, float* __restrict__ per_dls,
                                                       float* __restrict__ dz, float* __restrict__ logits_out, int B, int C) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float sc = __expf(ls_ptr[0]);
    const float* zr = z + (int64_t)b * C;
    float mx = -INFINITY;
    for (int c = tid; c < C; c += 256) mx = fmaxf(mx, zr[c] * sc);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < C; c += 256) s += __expf(zr[c] * sc - mx);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    const float lse = mx + __logf(s);
    const int a_i = (int)ya[b], b_i = (int)yb[b];
    float dls = 0.f;
    const float invB = 1.0f / (float)B;
    for (int c = tid; c < C; c += 256) {
        const float l = zr[c] * sc;
        if (logits_out) logits_out[(int64_t)b * C + c] = l;
        float gl = __expf(l - lse);
        if (c == a_i) gl -= lam;
        if (c == b_i) gl -= (1.0f - lam);
        gl *= invB;
        dls += gl * l;
        if (dz) dz[(int64_t)b * C + c] = gl * sc;
    }
    dls = wave_sum(dls);
    if (lane == 0) red[wave] = dls;
    __syncthreads();
    if (tid == 0) {
        per_dls[b] = (red[0] + red[1]) + (red[2] + red[3]);
        per_loss[b] = lam * (lse - zr[a_i] * sc) + (1.0f - lam) * (lse - zr[b_i] * sc);
    }
}

// loss_accum += loss, unless a persistent launch of the step gave up (the host re-runs the step): for steps whose BACKWARD may
// void them — the addition is queued behind that backward instead of being made by ce_finish_kernel
__global__ void loss_accumulate_kernel(const float* __restrict__ loss, double* __restrict__ loss_accum, const unsigned* __restrict__ guard) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && !(guard && guard[0] != 0u)) loss_accum[0] += (double)loss[0];
}

__global__ void ce_finish_kernel(const float* __restrict__ per_loss, const float* __restrict__ per_dls, int B,
                                 float* __restrict__ loss, float* __restrict__ dls, double* __restrict__ loss_accum,
                                 const unsigned* __restrict__ guard) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float l = 0.f, d = 0.f;
        for (int i = 0; i < B; ++i) { l += per_loss[i]; d += per_dls[i]; }
        l /= (float)B;
        loss[0] = l;
        if (dls) dls[0] = d;
        if (loss_accum && !(guard && guard[0] != 0u)) loss_accum[0] += (double)l;  // (not for a step whose forward gave up)
        // running float64 sum for the per-epoch loss.csv row (ONE host sync per epoch)
    }
}

// validate's host tail on the device (ref testing.py:449-458: torch.max / argmax per row, torchmetrics ConfusionMatrix.update):
// one wave per row of z [B, C]; pred = FIRST index of the row maximum (torch.argmax's tie rule on a contiguous row), then
// confusion[label, pred] += 1 (rows = true class, integer atomics: exact in any order) and correct += (pred == label).
__global__ __launch_bounds__(256) void argmax_confusion_kernel(const float* __restrict__ z, const int64_t* __restrict__ labels,
                                                               int B, int C, int64_t* __restrict__ pred_out,
                                                               float* __restrict__ max_out, unsigned long long* __restrict__ confusion,
                                                               unsigned long long* __restrict__ correct) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* zr = z + (int64_t)row * C;
    // torch.argmax's order: a NaN is greater than everything, ties (and several NaNs) go to the lower index — diverged logits must
    // give the predictions and the confusion matrix the host path gave (round-3 advisor finding: the shuffle reduce dropped NaNs)
    auto better = [](float v, int c, float b, int a) {
        const bool vn = v != v, bn = b != b;
        if (vn != bn) return vn;
        if (vn) return c < a;
        return v > b || (v == b && c < a);
    };
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = zr[c];
        if (arg == 0x7fffffff || better(v, c, best, arg)) { best = v; arg = c; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        if (oa != 0x7fffffff && (arg == 0x7fffffff || better(ov, oa, best, arg))) { best = ov; arg = oa; }
    }
    if (lane == 0) {
        if (arg == 0x7fffffff) arg = 0;
        const int64_t y = labels[row];
        if (pred_out) pred_out[row] = arg;
        if (max_out) max_out[row] = best;
        if (confusion && y >= 0 && y < C) atomicAdd(&confusion[y * C + arg], 1ull);
        if (correct && y == arg) atomicAdd(correct, 1ull);
    }
}

}  // namespace swat
using namespace swat;

// ---- the two shapes the finetuning tail actually uses, each with the memory-contiguous dimension across the lanes ----
// (a) B n-contiguous (sbn == 1):  C[m, n] = sum_k A[m, k] B[k, n].  64 columns x 16 k-groups per workgroup, 4 rows of A
//     per workgroup so a B element is loaded once for 4 outputs; the A operands are wave-uniform (scalar loads).
__global__ __launch_bounds__(1024) void sgemm_small_ncontig_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                   float* __restrict__ C, int M, int N, int K, int64_t sam,
                                                                   int64_t sak, int64_t sbk, int64_t ldc, float alpha,
                                                                   int accumulate) {
    __shared__ float sh[16][4][64];
    const int t = threadIdx.x & 63;
    const int n = blockIdx.x * 64 + t;
    const int m0 = blockIdx.y * 4;
    const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // k-group, wave-uniform
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int nn = n < N ? n : N - 1;
    const float* a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = A + (int64_t)(m0 + r < M ? m0 + r : M - 1) * sam;
#pragma unroll 4
    for (int k = j; k < K; k += 16) {
        const float b = B[(int64_t)k * sbk + nn];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaf(a[r][(int64_t)k * sak], b, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[j][r][t] = acc[r];
    __syncthreads();
    if (j < 4 && n < N && m0 + j < M) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += sh[q][j][t];  // fixed order
        float* c = C + (int64_t)(m0 + j) * ldc + n;
        *c = accumulate ? *c + alpha * v : alpha * v;
    }
}

// (b) A and B k-contiguous (sak == sbk == 1):  C[m, n] = <A[m, :], B[n, :]>.  One wave per column n keeps B's row in
//     registers (K <= 1024) and walks the rows of A with the lanes along k; wave-level reduction per output.
__global__ __launch_bounds__(256) void sgemm_small_kcontig_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                  float* __restrict__ C, int M, int N, int K, int64_t sam,
                                                                  int64_t sbn, int64_t ldc, float alpha, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float b[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = lane + 64 * q;
        b[q] = k < K ? B[(int64_t)n * sbn + k] : 0.f;
    }
    for (int m0 = 0; m0 < M; m0 += 8) {  // 8 rows of A in flight: one memory latency per 8 outputs
        float av[8][16];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float* a = A + (int64_t)(m0 + r < M ? m0 + r : M - 1) * sam;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = lane + 64 * q;
                av[r][q] = k < K ? a[k] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = fmaf(av[r][q], b[q], acc);
            acc = wave_sum(acc);
            if (lane == 0 && m0 + r < M) {
                float* c = C + (int64_t)(m0 + r) * ldc + n;
                *c = accumulate ? *c + alpha * acc : alpha * acc;
            }
        }
    }
}

// (c) A and B k-contiguous, a handful of rows (M <= 32), K % 64 == 0, K <= 768:  C[m, n] = <A[m, :], B[n, :]> as ONE k-ordered fp32 fma
//     chain per output (acc = fma(a_k, b_k, acc), k ascending from a zero accumulator) — bit for bit what the f32 matrix cores give
//     (v_mfma_f32_32x32x2_f32 is such a chain: csrc/retrieval_mfma.hip), which served these two products of the head (logits z = fn W^T,
//     du = df proj^T) until round 3 at 14-18 us each: eight 64-deep chunks staged one after the other, 256 dependent MFMAs, on 4-12
//     workgroups.  Here a workgroup owns 32 rows x 8 columns, one output per thread, and stages its WHOLE operands once (A: 32 x K,
//     B: 8 x K, <= 123 KB of LDS): one memory latency, then 512 dependent fmas per lane fed by float4 LDS reads (rows 16-B aligned
//     and 4 banks apart).  Measured alone: see tools/perf_small_chain.py.
__global__ __launch_bounds__(256) void sgemm_small_kordered_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                   float* __restrict__ C, int M, int N, int K, int64_t ldc) {
    extern __shared__ __attribute__((aligned(16))) float sko[];
    const int pitch = K + 4;
    float* at = sko;                 // [32][pitch]
    float* bt = sko + 32 * pitch;    // [8][pitch]
    const int tid = threadIdx.x, n0 = blockIdx.x * 8;
    const int kq = K >> 2;           // float4 per row
    for (int i = tid; i < 40 * kq; i += 256) {
        const int row = i / kq, q = i - row * kq;
        if (row < 32) {
            const int m = row < M ? row : M - 1;
            *reinterpret_cast<float4*>(at + row * pitch + 4 * q) = *reinterpret_cast<const float4*>(A + (int64_t)m * K + 4 * q);
        } else {
            const int n = n0 + row - 32 < N ? n0 + row - 32 : N - 1;
            *reinterpret_cast<float4*>(bt + (row - 32) * pitch + 4 * q) = *reinterpret_cast<const float4*>(B + (int64_t)n * K + 4 * q);
        }
    }
    __syncthreads();
    const int m = tid >> 3, c = tid & 7;
    const float* ar = at + m * pitch;
    const float* br = bt + c * pitch;
    float acc = 0.f;
#pragma unroll 8
    for (int kk = 0; kk < K; kk += 4) {
        const float4 a = *reinterpret_cast<const float4*>(ar + kk), b = *reinterpret_cast<const float4*>(br + kk);
        acc = fmaf(a.x, b.x, acc);
        acc = fmaf(a.y, b.y, acc);
        acc = fmaf(a.z, b.z, acc);
        acc = fmaf(a.w, b.w, acc);
    }
    if (m < M && n0 + c < N) C[(int64_t)m * ldc + n0 + c] = acc;
}

SWAT_API int swat_sgemm_small(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak,
                              int64_t sbk, int64_t sbn, int64_t ldc, float alpha, int accumulate, swat_stream_t s) {
    if (M <= 0 || N <= 0) return 0;
    if (sbn == 1 && K >= 64) {
        hipLaunchKernelGGL(sgemm_small_ncontig_kernel, dim3((N + 63) / 64, (M + 3) / 4), dim3(1024), 0, (hipStream_t)s, A, B, C, M,
                           N, K, sam, sak, sbk, ldc, alpha, accumulate);
    } else if (sak == 1 && sbk == 1 && sam == K && sbn == K && alpha == 1.0f && !accumulate && K % 64 == 0 && M <= 32 && K <= 768) {
        // both operands k-contiguous, a handful of rows — head logits z = fn W^T and du = df proj^T: the k-ordered fp32 chain
        // (bit-identical to the f32-matrix-core product swat_sim_topk_candidates that computes the larger cases below)
        const int lds = 40 * (K + 4) * 4;
        static int lds_ok = 0;   // (the attribute is raised once to the largest size: 123 KB at K = 768)
        if (!lds_ok) {
            SWAT_HIP(hipFuncSetAttribute((const void*)sgemm_small_kordered_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * (768 + 4) * 4));
            lds_ok = 1;
        }
        hipLaunchKernelGGL(sgemm_small_kordered_kernel, dim3((N + 7) / 8), dim3(256), lds, (hipStream_t)s, A, B, C, M, N, K, ldc);
    } else if (sak == 1 && sbk == 1 && sam == K && sbn == K && alpha == 1.0f && !accumulate && K % 64 == 0 && M <= 64) {
        return swat_sim_topk_candidates(A, M, B, N, K, nullptr, 0, nullptr, 0, nullptr, C, ldc, s);
    } else if (sak == 1 && sbk == 1 && K <= 1024 && M <= 256) {
        hipLaunchKernelGGL(sgemm_small_kcontig_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)s, A, B, C, M, N, K, sam, sbn,
                           ldc, alpha, accumulate);
    } else {
        hipLaunchKernelGGL(sgemm_small_kernel, dim3((N + 63) / 64, M), dim3(256), 0, (hipStream_t)s, A, B, C, M, N, K, sam, sak,
                           sbk, sbn, ldc, alpha, accumulate);
    }
    return check_launch("swat_sgemm_small");
}

SWAT_API int swat_l2norm_fwd(const float* x, float* y, float* norms, int rows, int d, swat_stream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, x, y, norms, rows, d);
    return check_launch("swat_l2norm_fwd");
}

SWAT_API int swat_l2norm_bwd(const float* y, const float* norms, const float* dy, float* dx, int rows, int d,
                             swat_stream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, y, norms, dy, dx, rows, d);
    return check_launch("swat_l2norm_bwd");
}

namespace swat {
// The head's path from the features to their gradient as ONE launch, a workgroup per sample: L2 normalise -> z = fn W^T -> mixed CE on
// the scaled logits (ce_mixed_kernel's arithmetic) -> d fn = d z W -> back through the normalisation.  Every step is local to a sample
// (the batch means and d W_head are not: ce_finish_kernel and the weight-gradient product follow, off the critical path), and each of
// the five launches this replaces costs 5-14 us of launch latency for a few microseconds of work.  1 024 threads per sample: with a
// workgroup per sample only B CUs work, each pulling W (C x E fp32, L2-resident) twice, so the loads must be wide (float4) and many —
// z by a wave per class (lanes split the features), d fn by (class group, float4 of features) threads with eight partial sums per
// feature added in group order.  Fixed summation orders: repeat runs agree bit
// for bit.  Outputs as the separate launches leave them: fn, nrm, z, dz, per-sample loss / d logit_scale, d feats.
constexpr int HR_THREADS = 1024, HR_WAVES = HR_THREADS / 64;
__global__ __launch_bounds__(HR_THREADS) void head_rows_kernel(const float* __restrict__ feats, const float* __restrict__ W, const float* __restrict__ ls_ptr,
                                                               const int64_t* __restrict__ ya, const int64_t* __restrict__ yb, float lam,
                                                               float* __restrict__ fn_out, float* __restrict__ nrm_out, float* __restrict__ z_out,
                                                               float* __restrict__ dz_out, float* __restrict__ per_loss, float* __restrict__ per_dls,
                                                               float* __restrict__ dfeats, int B, int E, int C) {
    extern __shared__ float hr_smem[];   // fn [E] | z [C] | dz [C] | partial d fn [8][E]
    __shared__ float red[HR_WAVES];
    float* fn = hr_smem;
    float* zs = fn + E;
    float* dzs = zs + C;
    float* part = dzs + C;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto block_sum = [&](float v) {   // fixed order: lanes by the xor tree, waves 0 .. 15 in sequence
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        float t = red[0];
#pragma unroll
        for (int w = 1; w < HR_WAVES; ++w) t += red[w];
        return t;
    };
    const float* x = feats + (int64_t)b * E;
    // ---- L2 normalise (no eps, ref utils/training.py:333)
    float s = 0.f;
    for (int e = tid; e < E; e += HR_THREADS) { const float v = x[e]; s += v * v; }
    const float n = sqrtf(block_sum(s));
    for (int e = tid; e < E; e += HR_THREADS) {
        const float v = x[e] / n;
        fn[e] = v;
        fn_out[(int64_t)b * E + e] = v;
    }
    if (tid == 0) nrm_out[b] = n;
    __syncthreads();
    // ---- z[c] = sum_e fn[e] W[c][e]: a wave per class, lanes split the features by float4 (E % 4 == 0)
    const int e4n = E >> 2;
    if (e4n <= 128) {
        // E <= 512 (every CLIP head of the reference): a lane holds at most two float4 of a class row — the rows of SEVEN classes are
        // loaded before the first is used, so a wave waits for memory twice for its ~13 classes instead of once per class (each wait
        // is ~1 us of L2 latency: the class loop was 13 of the kernel's 24 us).  Same products in the same order per class.
        constexpr int ZU = 7;
        const bool two = lane + 64 < e4n, one = lane < e4n;
        const float4 f0 = one ? reinterpret_cast<const float4*>(fn)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 f1 = two ? reinterpret_cast<const float4*>(fn)[lane + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c0 = wave; c0 < C; c0 += HR_WAVES * ZU) {
            float4 w0[ZU], w1[ZU];
#pragma unroll
            for (int u = 0; u < ZU; ++u) {
                const int c = c0 + HR_WAVES * u;
                const float4* wr = reinterpret_cast<const float4*>(W + (int64_t)(c < C ? c : c0) * E);
                w0[u] = one ? wr[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                w1[u] = two ? wr[lane + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < ZU; ++u) {
                const int c = c0 + HR_WAVES * u;
                float a = 0.f;
                if (one) { a = fmaf(f0.x, w0[u].x, a); a = fmaf(f0.y, w0[u].y, a); a = fmaf(f0.z, w0[u].z, a); a = fmaf(f0.w, w0[u].w, a); }
                if (two) { a = fmaf(f1.x, w1[u].x, a); a = fmaf(f1.y, w1[u].y, a); a = fmaf(f1.z, w1[u].z, a); a = fmaf(f1.w, w1[u].w, a); }
                a = wave_sum(a);
                if (lane == 0 && c < C) {
                    zs[c] = a;
                    z_out[(int64_t)b * C + c] = a;
                }
            }
        }
    } else {
    for (int c = wave; c < C; c += HR_WAVES) {
        const float4* wr = reinterpret_cast<const float4*>(W + (int64_t)c * E);
        float a = 0.f;
        for (int q = lane; q < e4n; q += 64) {
            const float4 wv = wr[q];
            const float4 fv = reinterpret_cast<const float4*>(fn)[q];
            a = fmaf(fv.x, wv.x, a); a = fmaf(fv.y, wv.y, a); a = fmaf(fv.z, wv.z, a); a = fmaf(fv.w, wv.w, a);
        }
        a = wave_sum(a);
        if (lane == 0) {
            zs[c] = a;
            z_out[(int64_t)b * C + c] = a;
        }
    }
    }
    __syncthreads();
    // ---- mixed cross entropy on the scaled logits (the arithmetic of ce_mixed_kernel, sums over 16 waves)
    const float sc = __expf(ls_ptr[0]);
    float mx = -INFINITY;
    for (int c = tid; c < C; c += HR_THREADS) mx = fmaxf(mx, zs[c] * sc);
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < HR_WAVES; ++w) mx = fmaxf(mx, red[w]);
    float se = 0.f;
    for (int c = tid; c < C; c += HR_THREADS) se += __expf(zs[c] * sc - mx);
    se = block_sum(se);
    const float lse = mx + __logf(se);
    const int a_i = (int)ya[b], b_i = (int)yb[b];
    float dls = 0.f;
    const float invB = 1.0f / (float)B;
    for (int c = tid; c < C; c += HR_THREADS) {
        const float l = zs[c] * sc;
        float gl = __expf(l - lse);
        if (c == a_i) gl -= lam;
        if (c == b_i) gl -= (1.0f - lam);
        gl *= invB;
        dls += gl * l;
        const float g = gl * sc;
        dzs[c] = g;
        dz_out[(int64_t)b * C + c] = g;
    }
    dls = block_sum(dls);   // (its barriers also publish dzs)
    if (tid == 0) {
        per_dls[b] = dls;
        per_loss[b] = lam * (lse - zs[a_i] * sc) + (1.0f - lam) * (lse - zs[b_i] * sc);
    }
    // ---- d fn[e] = sum_c dz[c] W[c][e]: thread = (class group cg of 8, float4 of features q), classes cg, cg + 8, ..; the eight partial
    //      sums of a feature are added in group order; then back through the normalisation: d feats = (d fn - fn <fn, d fn>) / ||x||
    {
        const int groups = HR_THREADS / 128;   // 8
        for (int q0 = 0; q0 < e4n; q0 += 128) {
            const int q = q0 + (tid & 127), cg = tid >> 7;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < e4n) {
#pragma unroll 8
                for (int c = cg; c < C; c += groups) {
                    const float4 wv = reinterpret_cast<const float4*>(W + (int64_t)c * E)[q];
                    const float g = dzs[c];
                    a.x = fmaf(g, wv.x, a.x); a.y = fmaf(g, wv.y, a.y); a.z = fmaf(g, wv.z, a.z); a.w = fmaf(g, wv.w, a.w);
                }
                reinterpret_cast<float4*>(part + cg * E)[q] = a;
            }
        }
    }
    __syncthreads();
    float dot = 0.f;
    for (int e = tid; e < E; e += HR_THREADS) {
        float a = part[e];
#pragma unroll
        for (int gi = 1; gi < HR_THREADS / 128; ++gi) a += part[gi * E + e];
        part[e] = a;   // (own element: no other thread reads it before the barrier inside block_sum)
        dot += fn[e] * a;
    }
    dot = block_sum(dot);
    const float inv = 1.0f / n;
    for (int e = tid; e < E; e += HR_THREADS) dfeats[(int64_t)b * E + e] = (part[e] - fn[e] * dot) * inv;
}

// -> 0 launched, 1 shapes not served (the caller issues the separate launches)
int launch_head_rows(const float* feats, const float* W, const float* ls, const int64_t* ya, const int64_t* yb, float lam, float* fn, float* nrm,
                     float* z, float* dz, float* per_sample_ws, float* dfeats, int B, int E, int C, hipStream_t s) {
    if (B < 1 || E < 4 || E % 4 || E > 2048 || C < 1 || C > 4096 || ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(feats)) & 15)) return 1;
    const size_t lds = (size_t)(E + 2 * C + (HR_THREADS / 128) * E) * sizeof(float);   // <= 106 KB
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
        SWAT_HIP(hipFuncSetAttribute((const void*)head_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(head_rows_kernel, dim3(B), dim3(HR_THREADS), lds, s, feats, W, ls, ya, yb, lam, fn, nrm, z, dz, per_sample_ws, per_sample_ws + B,
                       dfeats, B, E, C);
    return check_launch("head_rows_kernel");
}

// the two halves of swat_ce_mixed as separate launches (vit.hip: swat_head_loss can leave the second — the batch means, which feed
// nothing on the way to the backward — for later)
int launch_ce_rows(const float* z, const float* logit_scale_dev, const int64_t* ya, const int64_t* yb, float lam, float* per_sample_ws, float* dz,
                   float* logits_out, int B, int C, hipStream_t s) {
    hipLaunchKernelGGL(ce_mixed_kernel, dim3(B), dim3(256), 0, s, z, logit_scale_dev, ya, yb, lam, per_sample_ws, per_sample_ws + B, dz, logits_out,
                       B, C);
    return check_launch("ce_mixed_kernel");
}
int launch_ce_finish(const float* per_sample_ws, int B, float* loss, float* dlogit_scale, double* loss_accum, hipStream_t s) {
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, s, per_sample_ws, per_sample_ws + B, B, loss, dlogit_scale, loss_accum,
                       mega_guard_word());
    return check_launch("ce_finish_kernel");
}
}  // namespace swat

SWAT_API int swat_ce_mixed(const float* z, const float* logit_scale_dev, const int64_t* ya, const int64_t* yb, float lam,
                           float* per_sample_ws /* [2*B] */, float* loss, float* dlogit_scale, float* dz,
                           float* logits_out, double* loss_accum, int B, int C, swat_stream_t s) {
    SWAT_REQUIRE(B > 0 && C > 0, "swat_ce_mixed: bad shape");
    hipLaunchKernelGGL(ce_mixed_kernel, dim3(B), dim3(256), 0, (hipStream_t)s, z, logit_scale_dev, ya, yb, lam,
                       per_sample_ws, per_sample_ws + B, dz, logits_out, B, C);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, (const float*)per_sample_ws,
                       (const float*)(per_sample_ws + B), B, loss, dlogit_scale, loss_accum, mega_guard_word());
    return check_launch("swat_ce_mixed");
}

SWAT_API int swat_argmax_confusion(const float* z, const int64_t* labels, int B, int C, int64_t* pred_out, float* max_out,
                                   int64_t* confusion, int64_t* correct, swat_stream_t s) {
    SWAT_REQUIRE(z && labels && B >= 0 && C >= 1, "swat_argmax_confusion: null pointer or bad shape");
    if (B == 0) return 0;
    hipLaunchKernelGGL(argmax_confusion_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)s, z, labels, B, C, pred_out, max_out,
                       (unsigned long long*)confusion, (unsigned long long*)correct);
    return check_launch("swat_argmax_confusion");
}

SWAT_API int swat_loss_accumulate(const float* loss, double* loss_accum, swat_stream_t s) {
    SWAT_REQUIRE(loss && loss_accum, "swat_loss_accumulate: null pointer");
    hipLaunchKernelGGL(swat::loss_accumulate_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, loss, loss_accum, swat::mega_guard_word());
    return swat::check_launch("swat_loss_accumulate");
}
