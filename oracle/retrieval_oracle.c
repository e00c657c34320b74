/*
 * oracle/retrieval_oracle.c — TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT.
 *
 * Plain-C CPU restatement of SWAT's retrieval arithmetic (path B).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (swat_amd/) never does.
 *
 * Parity status: PINNED.  tests/golden/retrieval_*.npz were produced by importing the
 * reference's own retrieval/sample_retrieval.py in the build container
 * (oracle/make_golden.py) and tests/test_oracle_retrieval.py checks every function here
 * against them.
 *
 * Reference (paths relative to /root/reference):
 *   similarity   : retrieval/sample_retrieval.py:397-416 (t2t_similarity),
 *                  :335-353 (cal_t2i_similarity), :369-394 (i2i_similarity_p2p)
 *   rank + filter: :754 / :807-808 (sorted(..., key=sim, reverse=True) — CPython's sort is
 *                  stable, reverse=True keeps the original order among equal keys),
 *                  :439-482 (add_to_split), :492-540 (add_t2t_ranked_t2i_tshd_to_split)
 *   group by cls : :1387-1415 (transform_extracted_fea), classes visited in ascending
 *                  int order (:738-739, :789-790)
 *   near-dup     : :237-275 (remove_near_duplicates2)
 *
 * Canonical floating-point order.  The reference computes `feats @ prompt.t()` in fp32
 * through whatever BLAS its device has, so its summation order is unspecified.  This
 * build pins it: a dot product is ONE fp32 fused-multiply-add chain over k = 0..d-1,
 *     acc = 0;  acc = fmaf(x[k], q[k], acc)  for k ascending,
 * which is bit-for-bit what gfx950's v_mfma_f32_*_f32 and a per-lane v_fma_f32 chain
 * compute, so the HIP path can be held to bit equality with this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline float dot_chain(const float* x, const float* q, int d) {
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) acc = fmaf(x[k], q[k], acc);
    return acc;
}

/* sims[i] = <feats[i,:], q>  (sample_retrieval.py:397-416 with a [1,d] prompt). */
ORC_API void orc_similarity(const float* feats, int64_t n, int d, const float* q, float* sims) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) sims[i] = dot_chain(feats + i * (int64_t)d, q, d);
}

/* Segmented form: row i is scored against queries[seg_of_row(i)], segments given as CSR
 * offsets over rows already grouped by class (orc_group_by_label). */
ORC_API void orc_similarity_segmented(const float* feats, int64_t n, int d, const float* queries,
                                      const int64_t* seg_offsets, int64_t n_seg, float* sims) {
    (void)n;
    for (int64_t s = 0; s < n_seg; ++s) {
        const float* q = queries + s * (int64_t)d;
#pragma omp parallel for schedule(static)
        for (int64_t i = seg_offsets[s]; i < seg_offsets[s + 1]; ++i)
            sims[i] = dot_chain(feats + i * (int64_t)d, q, d);
    }
}

/* sims[qi*n + i] = <db[i,:], queries[qi,:]>  — the (N_query x N_db) form of BASELINE config 4. */
ORC_API void orc_similarity_multi(const float* db, int64_t n, int d, const float* queries,
                                  int64_t nq, float* sims) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float* x = db + i * (int64_t)d;
        for (int64_t qi = 0; qi < nq; ++qi) sims[qi * n + i] = dot_chain(x, queries + qi * (int64_t)d, d);
    }
}

/* i2i_similarity_p2p (:369-394): per row, min / max / mean over m few-shot embeddings.
 * mode 0=min 1=max 2=mean.  mean = (sequential fp32 sum in column order) / m. */
ORC_API void orc_similarity_p2p(const float* feats, int64_t n, int d, const float* fs, int m,
                                int mode, float* sims) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float* x = feats + i * (int64_t)d;
        float r = 0.0f;
        for (int j = 0; j < m; ++j) {
            float s = dot_chain(x, fs + (int64_t)j * d, d);
            if (j == 0) r = s;
            else if (mode == 0) r = s < r ? s : r;
            else if (mode == 1) r = s > r ? s : r;
            else r = r + s;
        }
        if (mode == 2) r = r / (float)m;
        sims[i] = r;
    }
}

typedef struct { float s; int64_t i; } orc_pair;

/* (sim desc, original index asc): what CPython's stable sorted(..., reverse=True) yields. */
static int cmp_desc_stable(const void* a, const void* b) {
    const orc_pair* x = (const orc_pair*)a;
    const orc_pair* y = (const orc_pair*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/* Full ranking of one class segment: order[r] = local row index at rank r. */
ORC_API void orc_rank(const float* sims, int64_t n, int64_t* order) {
    orc_pair* p = (orc_pair*)malloc(sizeof(orc_pair) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { p[i].s = sims[i]; p[i].i = i; }
    qsort(p, (size_t)n, sizeof(orc_pair), cmp_desc_stable);
    for (int64_t i = 0; i < n; ++i) order[i] = p[i].i;
    free(p);
}

/* Rank then walk (add_to_split :439-482; with t2i != NULL add_t2t_ranked_t2i_tshd_to_split
 * :492-540): keep row iff sim >= thr [and t2i >= t2i_thr] and !excluded[row]; stop after k.
 * `excluded` folds the reference's duplicates_dict / filtered_images_dict membership tests.
 * Returns the number kept; out_idx[0..ret) are local row indices in rank order. */
ORC_API int64_t orc_rank_filter(const float* sims, const float* t2i, const uint8_t* excluded,
                                int64_t n, float thr, float t2i_thr, int64_t k, int64_t* out_idx) {
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    orc_rank(sims, n, order);
    int64_t ct = 0;
    for (int64_t r = 0; r < n; ++r) {
        if (ct == k) break;
        int64_t i = order[r];
        if (sims[i] >= thr && (t2i == NULL || t2i[i] >= t2i_thr) && (excluded == NULL || !excluded[i]))
            out_idx[ct++] = i;
    }
    free(order);
    return ct;
}

/* Whole-index form: for every class segment s (rows [off[s], off[s+1]) of class-grouped
 * matrices) score caption_feats against queries[s] (t2t) and, when img_feats != NULL,
 * img_feats against queries[s] (t2i threshold); write up to k kept GROUPED row indices per
 * segment to out_idx[s*k ...] (padded with -1) and the count to out_cnt[s].
 * rank_by: 0 = rank by caption sim (T2T-rank / T2T-rank-T2I-tshd), 1 = rank by image sim
 * (T2I-rank, :1195-1243; img_feats required, no extra threshold). */
ORC_API void orc_sample_index(const float* caption_feats, const float* img_feats, int64_t n, int d,
                              const float* queries, const int64_t* off, int64_t n_seg,
                              const uint8_t* excluded, float thr, float t2i_thr, int use_t2i_thr,
                              int rank_by, int64_t k, int64_t* out_idx, int64_t* out_cnt,
                              float* out_sims /* nullable [n] */, float* out_t2i /* nullable [n] */) {
    (void)n;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t s = 0; s < n_seg; ++s) {
        int64_t b = off[s], e = off[s + 1], m = e - b;
        const float* q = queries + s * (int64_t)d;
        float* sim = (float*)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
        float* ti = NULL;
        const float* rank_src = rank_by == 0 ? caption_feats : img_feats;
        for (int64_t i = 0; i < m; ++i) sim[i] = dot_chain(rank_src + (b + i) * (int64_t)d, q, d);
        if (rank_by == 0 && use_t2i_thr) {
            ti = (float*)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
            for (int64_t i = 0; i < m; ++i) ti[i] = dot_chain(img_feats + (b + i) * (int64_t)d, q, d);
        }
        int64_t* loc = (int64_t*)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
        int64_t c = orc_rank_filter(sim, ti, excluded ? excluded + b : NULL, m, thr, t2i_thr, k, loc);
        for (int64_t r = 0; r < k; ++r) out_idx[s * k + r] = r < c ? b + loc[r] : -1;
        out_cnt[s] = c;
        if (out_sims) memcpy(out_sims + b, sim, sizeof(float) * (size_t)m);
        if (out_t2i && ti) memcpy(out_t2i + b, ti, sizeof(float) * (size_t)m);
        free(loc); free(sim); if (ti) free(ti);
    }
}

/* Global top-k per query over an un-segmented DB (BASELINE config 4): (value desc, row asc).
 * out_idx/out_val are [nq, k]; n >= k assumed by callers (shorter lists are padded -1/-inf). */
ORC_API void orc_topk_multi(const float* db, int64_t n, int d, const float* queries, int64_t nq,
                            int64_t k, int64_t* out_idx, float* out_val) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t qi = 0; qi < nq; ++qi) {
        orc_pair* p = (orc_pair*)malloc(sizeof(orc_pair) * (size_t)(n > 0 ? n : 1));
        const float* q = queries + qi * (int64_t)d;
        for (int64_t i = 0; i < n; ++i) { p[i].s = dot_chain(db + i * (int64_t)d, q, d); p[i].i = i; }
        qsort(p, (size_t)n, sizeof(orc_pair), cmp_desc_stable);
        for (int64_t r = 0; r < k; ++r) {
            out_idx[qi * k + r] = r < n ? p[r].i : -1;
            out_val[qi * k + r] = r < n ? p[r].s : -INFINITY;
        }
        free(p);
    }
}

/* transform_extracted_fea (:1387-1415) as a stable group-by: order[] lists original rows
 * class by class (classes ascending by integer id, rows in file order inside a class);
 * class_ids[0..n_cls) and offsets[0..n_cls] describe the groups.  Returns n_cls. */
static int cmp_label_stable(const void* a, const void* b) {
    const int64_t* x = (const int64_t*)a;
    const int64_t* y = (const int64_t*)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return (x[1] > y[1]) - (x[1] < y[1]);
}
ORC_API int64_t orc_group_by_label(const int64_t* labels, int64_t n, int64_t* order,
                                   int64_t* class_ids, int64_t* offsets) {
    int64_t* p = (int64_t*)malloc(sizeof(int64_t) * 2 * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { p[2 * i] = labels[i]; p[2 * i + 1] = i; }
    qsort(p, (size_t)n, 2 * sizeof(int64_t), cmp_label_stable);
    int64_t nc = 0;
    for (int64_t i = 0; i < n; ++i) {
        order[i] = p[2 * i + 1];
        if (i == 0 || p[2 * i] != p[2 * (i - 1)]) { class_ids[nc] = p[2 * i]; offsets[nc] = i; ++nc; }
    }
    offsets[nc] = n;
    free(p);
    return nc;
}

/* remove_near_duplicates2 (:237-275): marked[j] = 1 iff exists i < j with <f_i, f_j> > thr.
 * (The reference then matches these POSITIONAL indices against int(basename(file)) — that
 * quirk lives in the Python wrapper, not here.) */
ORC_API void orc_near_dup(const float* feats, int64_t n, int d, float thr, uint8_t* marked) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t j = 0; j < n; ++j) {
        uint8_t m = 0;
        for (int64_t i = 0; i < j && !m; ++i)
            if (dot_chain(feats + i * (int64_t)d, feats + j * (int64_t)d, d) > thr) m = 1;
        marked[j] = m;
    }
}
