"""oracle/synth.py — TEST INFRASTRUCTURE ONLY.

Seeded synthetic inputs shared by oracle/make_golden.py (which feeds them to the reference)
and tests/ (which feed them to the oracle and to the HIP path).  Everything is drawn from
numpy's legacy `RandomState`, whose stream is frozen across numpy versions, so a fixture only
needs to store its spec.
"""
import numpy as np

# mined.pth-shaped cases (retrieval/extract_mined_feature.py:270-279).
#  grid : every element is j/64 with integer |j| <= 8, so every dot product is exact in fp32 in
#         ANY summation order -> the reference's BLAS, the C oracle and the GPU agree bit for bit
#         and the many exact ties pin the stable-sort tie rule.
#  float: unit-norm gaussian rows, the realistic case.
RETRIEVAL_CASES = {
    "grid": dict(kind="grid", seed=20240131, d=512, k=40,
                 class_ids=[0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144],
                 sizes=[300, 1, 39, 40, 41, 700, 2, 250, 64, 65, 1000, 128], n_dups=60),
    "float": dict(kind="float", seed=777, d=512, k=50,
                  class_ids=[0, 1, 2, 3, 4, 5, 6, 7],
                  sizes=[500, 37, 800, 64, 1, 333, 1200, 90], n_dups=40),
}


def retrieval_case(kind, seed, d, k, class_ids, sizes, n_dups):
    rs = np.random.RandomState(seed)
    n = int(sum(sizes))
    labels = np.repeat(np.asarray(class_ids, np.int64), sizes)
    labels = labels[rs.permutation(n)]  # interleave classes: the group-by has work to do
    C = len(class_ids)
    cls_pos = {c: i for i, c in enumerate(class_ids)}
    if kind == "grid":
        q_int = rs.randint(-8, 9, size=(C, d))
        w = rs.randint(0, 3, size=(n, 2)) / 8.0
        qi = q_int[[cls_pos[c] for c in labels.tolist()]]
        cap = np.clip(np.rint(w[:, :1] * qi) + rs.randint(-4, 5, size=(n, d)), -8, 8)
        img = np.clip(np.rint(w[:, 1:] * qi) + rs.randint(-4, 5, size=(n, d)), -8, 8)
        queries = (q_int / 64.0).astype(np.float32)
        cap = (cap / 64.0).astype(np.float32)
        img = (img / 64.0).astype(np.float32)
    else:
        def unit(x):
            x = x.astype(np.float32)
            return x / np.linalg.norm(x, axis=-1, keepdims=True).astype(np.float32)
        centres = rs.randn(C, d)
        qi = centres[[cls_pos[c] for c in labels.tolist()]]
        w = rs.rand(n, 2) * np.array([0.12, 0.5])
        cap = unit(w[:, :1] * qi + rs.randn(n, d))
        img = unit(w[:, 1:] * qi + rs.randn(n, d))
        queries = unit(centres)
    # exact duplicates inside a class (ties for the sort, > 0.9 hits for the dedup)
    for _ in range(n_dups):
        a = rs.randint(n)
        same = np.nonzero(labels == labels[a])[0]
        b = same[rs.randint(len(same))]
        cap[a], img[a] = cap[b], img[b]
    # file paths '<root>/<cls>/<id>.jpg' (sample_retrieval.py:262, :485-489); ids are a permutation
    # of the positions inside the class so the reference's positional-vs-basename dedup quirk bites.
    paths = [None] * n
    for c in class_ids:
        rows = np.nonzero(labels == c)[0]
        ids = rs.permutation(len(rows))
        for r, i in zip(rows.tolist(), ids.tolist()):
            paths[r] = f"/synthetic/retrieved/{c}/{i}.jpg"
    excl_dup = rs.rand(n) < 0.05
    excl_filt = rs.rand(n) < 0.03
    return dict(labels=labels, cap=np.ascontiguousarray(cap), img=np.ascontiguousarray(img),
                queries=np.ascontiguousarray(queries), class_ids=list(class_ids), paths=paths,
                excl_dup=excl_dup, excl_filt=excl_filt)


def unit_rows(rs, n, d):
    x = rs.randn(n, d).astype(np.float32)
    return x / np.linalg.norm(x, axis=-1, keepdims=True).astype(np.float32)


def hash_tokenize(ctx=16, vocab=128):
    """A deterministic stand-in for open_clip's BPE tokenizer (test infrastructure: the B8 golden needs SOME tokenizer both
    the imported reference and the GPU test can call): lower-cased whitespace words -> 1 + crc32(word) % (vocab - 3),
    <start> = vocab - 2 in front, <end> = vocab - 1 behind (the largest id, so argmax finds it like CLIP's EOT), zero padded,
    truncated to ctx like open_clip's tokenizer."""
    import zlib

    import torch

    def tok(texts):
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), ctx, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [vocab - 2] + [1 + zlib.crc32(w.encode()) % (vocab - 3) for w in t.lower().split()]
            ids = ids[:ctx - 1] + [vocab - 1]
            out[i, :len(ids)] = torch.tensor(ids)
        return out
    return tok
