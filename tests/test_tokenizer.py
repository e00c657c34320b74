"""CPU: swat_amd.tokenizer (CLIP's SimpleTokenizer, which the reference obtains through open_clip.get_tokenizer at
utils/extras.py:157) on a small synthetic merges file, against an INDEPENDENT byte-level BPE built from the same merges with
the HF `tokenizers` library (the construction transformers' CLIPTokenizer uses).  The real 49 152-entry vocabulary is an
open_clip asset that is not reachable here; the algorithm is what is pinned."""
import collections
import gzip
import os

import pytest
import torch

CORPUS = ("a photo of a dog. a photo of a cat, a type of pet. the bird's nest isn't empty! 12 red-winged blackbirds "
          "an image of the scarlet tanager (piranga olivacea); a close-up photo of a northern cardinal. don't they're "
          "photo photos photograph photography birds bird birding low lower lowest new newer newest wide wider widest").split()


def learn_merges(words, n_merges, byte_encoder):
    """Plain BPE training on byte-level symbols with CLIP's end-of-word marker."""
    vocab = collections.Counter()
    for w in words:
        sym = [byte_encoder[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        vocab[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for sym, c in vocab.items():
            for a, b in zip(sym[:-1], sym[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        new = collections.Counter()
        for sym, c in vocab.items():
            out, i = [], 0
            while i < len(sym):
                if i + 1 < len(sym) and (sym[i], sym[i + 1]) == best:
                    out.append(sym[i] + sym[i + 1]); i += 2
                else:
                    out.append(sym[i]); i += 1
            new[tuple(out)] += c
        vocab = new
    return merges


@pytest.fixture(scope="module")
def bpe_file(tmp_path_factory):
    from swat_amd import tokenizer as tk
    import re
    pat = re.compile(r"'s|'t|'re|'ve|'m|'ll|'d|[a-zA-Z]+|[0-9]|[^\sa-zA-Z0-9]+", re.IGNORECASE)
    words = [w for t in CORPUS for w in re.findall(pat, t.lower())]
    merges = learn_merges(words, 120, tk._bytes_to_unicode())
    path = tmp_path_factory.mktemp("bpe") / "bpe_tiny.txt.gz"
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: synthetic\n")
        for a, b in merges:
            f.write(f"{a} {b}\n")
    return str(path), merges


TEXTS = ["a photo of a dog.", "A  Photo\tof the  BIRD'S nest!", "photography isn't birding", "12 red-winged blackbirds",
         "the newest, widest &amp; lowest", "café naïve — test", "", "they're don't we'll i'm"]


def test_simple_tokenizer_matches_an_independent_bpe(bpe_file):
    from tokenizers import Regex, Tokenizer, pre_tokenizers
    from tokenizers.models import BPE
    from swat_amd import tokenizer as tk
    path, merges = bpe_file
    mine = tk.SimpleTokenizer(path)
    assert len(mine.encoder) == 512 + len(merges) + 2
    hf = Tokenizer(BPE(vocab=dict(mine.encoder), merges=list(merges), end_of_word_suffix="</w>", fuse_unk=False))
    hf.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(r"'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"), behavior="removed", invert=True),
        pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    import html, re
    for text in TEXTS:
        clean = re.sub(r"\s+", " ", html.unescape(html.unescape(text))).strip().lower()
        want = hf.encode(clean).ids
        assert mine.encode(text) == want, (text, mine.encode(text), want)
    # every id decodes back to its symbol, words end in the </w> form
    inv = {v: k for k, v in mine.encoder.items()}
    ids = mine.encode("a photo of a dog.")
    assert "".join(inv[i] for i in ids).replace("</w>", " ").strip() == "a photo of a dog ."


def test_tokenize_pads_truncates_and_marks_sot_eot(bpe_file, monkeypatch):
    from swat_amd import tokenizer as tk
    path, merges = bpe_file
    monkeypatch.setenv("SWAT_BPE_VOCAB", path)
    monkeypatch.setattr(tk, "_tok", None)
    out = tk.tokenize(["a photo of a dog.", "photo " * 40], context_length=16)
    sot, eot = tk._tok.encoder["<start_of_text>"], tk._tok.encoder["<end_of_text>"]
    assert out.shape == (2, 16) and out.dtype == torch.long
    assert out[0, 0] == sot and out[1, 0] == sot
    n = len(tk._tok.encode("a photo of a dog."))
    assert out[0, n + 1] == eot and (out[0, n + 2:] == 0).all()
    assert out[1, -1] == eot                       # truncated rows still end in EOT (open_clip's rule)
    assert int(out[0].argmax()) == n + 1           # EOT is the highest id: what encode_text pools on
    monkeypatch.setattr(tk, "_tok", None)
    monkeypatch.delenv("SWAT_BPE_VOCAB")
    with pytest.raises(NotImplementedError):
        tk.tokenize("x")
