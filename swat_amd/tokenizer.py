"""CLIP BPE tokenizer hook (ref: open_clip.get_tokenizer, utils/extras.py:157).  The BPE merges file is an asset of
open_clip that is not reachable here; point SWAT_BPE_VOCAB at `bpe_simple_vocab_16e6.txt.gz` to enable it."""
import gzip
import html
import os
import re
from functools import lru_cache

import regex  # \p{L} / \p{N} classes, as open_clip's tokenizer uses

import torch


@lru_cache()
def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return dict(zip(bs, [chr(n) for n in cs]))


class SimpleTokenizer:
    def __init__(self, bpe_path):
        self.byte_encoder = _bytes_to_unicode()
        merges = gzip.open(bpe_path).read().decode("utf-8").split('\n')[1:49152 - 256 - 2 + 1]
        merges = [tuple(m.split()) for m in merges if m.strip()]  # (a shorter, e.g. synthetic, file ends in an empty line)
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + '</w>' for v in vocab]
        for m in merges:
            vocab.append(''.join(m))
        vocab.extend(['<start_of_text>', '<end_of_text>'])
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {'<start_of_text>': '<start_of_text>', '<end_of_text>': '<end_of_text>'}
        self.pat = regex.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                 regex.IGNORECASE)

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + '</w>',)
        pairs = set(zip(word[:-1], word[1:]))
        if not pairs:
            return token + '</w>'
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float('inf')))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new_word, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                    new_word.extend(word[i:j])
                    i = j
                except ValueError:
                    new_word.extend(word[i:])
                    break
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new_word.append(first + second)
                    i += 2
                else:
                    new_word.append(word[i])
                    i += 1
            word = tuple(new_word)
            if len(word) == 1:
                break
            pairs = set(zip(word[:-1], word[1:]))
        word = ' '.join(word)
        self.cache[token] = word
        return word

    def encode(self, text):
        text = re.sub(r'\s+', ' ', html.unescape(html.unescape(text))).strip().lower()
        out = []
        for token in regex.findall(self.pat, text):
            token = ''.join(self.byte_encoder[b] for b in token.encode('utf-8'))
            out.extend(self.encoder[t] for t in self.bpe(token).split(' '))
        return out


_tok = None


def tokenize(texts, context_length=77):
    global _tok
    if _tok is None:
        path = os.environ.get('SWAT_BPE_VOCAB')
        if not path or not os.path.exists(path):
            raise NotImplementedError("CLIP BPE vocabulary not available: set SWAT_BPE_VOCAB to bpe_simple_vocab_16e6.txt.gz")
        _tok = SimpleTokenizer(path)
    if isinstance(texts, str):
        texts = [texts]
    sot, eot = _tok.encoder['<start_of_text>'], _tok.encoder['<end_of_text>']
    result = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        toks = [sot] + _tok.encode(t) + [eot]
        if len(toks) > context_length:
            toks = toks[:context_length]
            toks[-1] = eot
        result[i, :len(toks)] = torch.tensor(toks)
    return result
