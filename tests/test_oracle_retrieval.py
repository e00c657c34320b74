"""CPU: the C oracle of path B against the golden vectors recorded from the reference
(oracle/make_golden.py ran retrieval/sample_retrieval.py's own samplers on these inputs)."""
import json
import os

import numpy as np
import pytest

from oracle import retrieval_oracle as orc
from oracle import synth


def _load(golden_dir, kind):
    g = np.load(os.path.join(golden_dir, f"retrieval_{kind}.npz"), allow_pickle=False)
    spec = json.loads(str(g["spec"]))
    return g, spec, synth.retrieval_case(**spec)


@pytest.mark.parametrize("kind", ["grid", "float"])
def test_group_by_label_matches_reference(golden_dir, kind):
    g, spec, case = _load(golden_dir, kind)
    order, cls, off = orc.group_by_label(case["labels"])
    assert np.array_equal(order, g["group_order"])
    assert np.array_equal(cls, g["group_classes"])
    assert np.array_equal(np.diff(off), g["group_sizes"])


@pytest.mark.parametrize("kind", ["grid", "float"])
def test_similarity_matches_reference(golden_dir, kind):
    g, spec, case = _load(golden_dir, kind)
    for i, c in enumerate(case["class_ids"]):
        rows = np.nonzero(case["labels"] == c)[0]
        s = orc.similarity(case["cap"][rows], case["queries"][i])
        t = orc.similarity(case["img"][rows], case["queries"][i])
        if kind == "grid":  # exact arithmetic: bit equality with the reference's BLAS
            assert np.array_equal(s, g["sims_t2t"][rows]) and np.array_equal(t, g["sims_t2i"][rows])
        else:  # unspecified BLAS summation order in the reference: 512-term fp32 dot, unit rows
            assert np.abs(s - g["sims_t2t"][rows]).max() <= 1e-6
            assert np.abs(t - g["sims_t2i"][rows]).max() <= 1e-6


@pytest.mark.parametrize("kind", ["grid", "float"])
@pytest.mark.parametrize("name,kw", [("t2t", dict(rank_by="t2t", use_t2i_thr=False)),
                                     ("t2t_t2i", dict(rank_by="t2t", use_t2i_thr=True)),
                                     ("t2i", dict(rank_by="t2i"))])
def test_samplers_match_reference(golden_dir, kind, name, kw):
    g, spec, case = _load(golden_dir, kind)
    order, cls, off = orc.group_by_label(case["labels"])
    excl = (case["excl_dup"] | case["excl_filt"])[order].astype(np.uint8)
    idx, cnt = orc.sample_index(case["cap"][order], case["img"][order], case["queries"], off, spec["k"],
                                excluded=excl, **kw)
    rows = np.concatenate([order[idx[s, :cnt[s]]] for s in range(len(cls))])
    assert np.array_equal(rows, g[f"{name}_rows"])
    assert {str(int(c)): int(n) for c, n in zip(cls, cnt)} == json.loads(str(g[f"{name}_counts"]))
    paths_g = [case["paths"][i] for i in order]
    assert orc.sample_file_list_text(paths_g, cls, idx, cnt) == str(g[f"{name}_txt"])
    assert (idx[np.arange(idx.shape[1])[None, :] >= cnt[:, None]] == -1).all()


@pytest.mark.parametrize("kind", ["grid", "float"])
def test_near_dup_matches_reference(golden_dir, kind):
    g, spec, case = _load(golden_dir, kind)
    got = []
    fracs = []
    for c in case["class_ids"]:
        rows = np.nonzero(case["labels"] == c)[0]
        files, n_marks = orc.near_dup_files(case["img"][rows], [case["paths"][r] for r in rows])
        got += [r for r in rows if case["paths"][r] in files]
        fracs.append(n_marks / len(rows))
    assert np.array_equal(np.array(sorted(got)), g["dedup_rows"])
    assert np.allclose(fracs, g["dedup_frac"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("kind", ["grid", "float"])
@pytest.mark.parametrize("mode", ["min", "max", "mean"])
def test_p2p_matches_reference(golden_dir, kind, mode):
    g, spec, case = _load(golden_dir, kind)
    c0 = min(case["class_ids"])
    rows = np.nonzero(case["labels"] == c0)[0]
    got = orc.similarity_p2p(case["img"][rows], case["img"][rows[:16]], mode)
    if kind == "grid" and mode != "mean":
        assert np.array_equal(got, g[f"p2p_{mode}"])
    else:
        assert np.abs(got - g[f"p2p_{mode}"]).max() <= 1e-6


def test_rank_filter_edge_cases():
    # empty segment, k larger than the segment, everything filtered, scalar segment, -0.0 ties
    assert len(orc.rank_filter(np.zeros(0, np.float32), 5)) == 0
    assert orc.rank_filter(np.array([0.5], np.float32), 5).tolist() == [0]
    assert len(orc.rank_filter(np.array([-0.5, -0.1], np.float32), 5)) == 0
    s = np.array([0.0, -0.0, 0.0, 1.0, -0.0], np.float32)
    assert orc.rank_filter(s, 5).tolist() == [3, 0, 1, 2, 4]  # -0.0 == 0.0: ties keep row order
    assert orc.rank_filter(s, 2).tolist() == [3, 0]
    ex = np.array([0, 1, 0, 1, 0], np.uint8)
    assert orc.rank_filter(s, 5, excluded=ex).tolist() == [0, 2, 4]
    t2i = np.array([0.25, 0.3, 0.2499, 0.9, 0.25], np.float32)
    assert orc.rank_filter(s, 5, t2i=t2i).tolist() == [3, 0, 1, 4]


def test_topk_multi_is_rank_prefix():
    rs = np.random.RandomState(3)
    db = synth.unit_rows(rs, 3000, 64)
    db[100] = db[7]
    db[2000] = db[7]
    q = synth.unit_rows(rs, 5, 64)
    idx, val = orc.topk_multi(db, q, 17)
    sims = orc.similarity_multi(db, q)
    for i in range(5):
        assert np.array_equal(idx[i], orc.rank(sims[i])[:17])
        assert np.array_equal(val[i], sims[i][idx[i]])


def test_c_oracle_is_clean_under_address_and_ub_sanitizers():
    """`make -C oracle asan`: the golden-vector checks of this file once more against the -fsanitize=address,undefined build
    of the C restatement (CPU only; GPU sanitizers are not available on the pool)."""
    import shutil
    import subprocess
    import sys
    if os.environ.get("SWAT_ORACLE_LIB"):
        pytest.skip("already running against a substitute oracle build")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not shutil.which("make") or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this image")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"])
    env = dict(os.environ, SWAT_ORACLE_LIB=os.path.join(root, "oracle", "_build", "libswat_oracle_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "not sanitizers"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
