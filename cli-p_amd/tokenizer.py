"""CLIP byte-pair tokenizer: `clip.tokenize([in_text])` (reference query-index.py:107).

The upstream implementation (openai/CLIP clip/simple_tokenizer.py, un-vendored and unpinned:
reference setup.sh:22-24) is restated from its published algorithm: lower-case, whitespace
clean-up, HTML unescape, the split pattern, byte -> printable-unicode mapping, greedy lowest-rank
pair merging with "</w>" word ends, <|startoftext|> = vocab-2 and <|endoftext|> = vocab-1,
zero padding to the context length, and an error for prompts that do not fit.

The merge table (bpe_simple_vocab_16e6.txt.gz upstream) is NOT on this machine and cannot be
fetched: pass its path, or set $CLIPMI_BPE_PATH. PARITY UNPINNED against the upstream vocabulary;
the algorithm itself is tested on a small synthetic merge table (tests/test_tokenizer.py).
`ftfy.fix_text` (mojibake repair) is not applied: ftfy is not a dependency here.
"""
import gzip
import html
import os
from functools import lru_cache

import numpy as np

try:
    import regex as re
    _PAT = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""
except ImportError:          # pragma: no cover - regex is present in the target image
    import re
    _PAT = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[^\W\d_]+|\d|[^\s\w]+"""


@lru_cache()
def bytes_to_unicode():
    """Reversible map from the 256 byte values to printable unicode characters (printable bytes map
    to themselves, the rest to code points from 256 upwards)."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    chars = keep[:]
    n = 0
    for b in range(256):
        if b not in keep:
            keep.append(b)
            chars.append(256 + n)
            n += 1
    return dict(zip(keep, (chr(c) for c in chars)))


def _pairs(word):
    return {(word[i], word[i + 1]) for i in range(len(word) - 1)}


def _clean(text):
    text = html.unescape(html.unescape(text)).strip()
    return " ".join(text.split()).strip()


class SimpleTokenizer:
    def __init__(self, bpe_path=None, merges=None, n_merges=49152 - 256 - 2):
        if merges is None:
            bpe_path = bpe_path or os.environ.get("CLIPMI_BPE_PATH")
            if not bpe_path or not os.path.exists(bpe_path):
                raise FileNotFoundError(
                    "CLIP BPE merge table not found: pass bpe_path or set CLIPMI_BPE_PATH to a local copy of "
                    "bpe_simple_vocab_16e6.txt.gz (it is not bundled and nothing is downloaded)")
            opener = gzip.open if bpe_path.endswith(".gz") else open
            with opener(bpe_path, "rt", encoding="utf-8") as f:
                lines = f.read().split("\n")
            merges = [tuple(m.split()) for m in lines[1:1 + n_merges] if m.strip()]
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {t: i for i, t in enumerate(vocab)}
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(_PAT, re.IGNORECASE)
        self.sot = self.encoder["<|startoftext|>"]
        self.eot = self.encoder["<|endoftext|>"]

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            best = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if best not in self.bpe_ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        res = " ".join(word)
        self.cache[token] = res
        return res

    def encode(self, text):
        ids = []
        for tok in self.pat.findall(_clean(text).lower()):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def decode(self, ids):
        text = "".join(self.decoder[i] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


_default = None


def tokenize(texts, context_length=77, truncate=False, tokenizer=None):
    """clip.tokenize: str or list[str] -> int64 [n, context_length] (numpy -> torch on demand by the
    caller: the model accepts either). Raises RuntimeError when a prompt is too long, as upstream."""
    global _default
    if isinstance(texts, str):
        texts = [texts]
    if tokenizer is None:
        if _default is None:
            _default = SimpleTokenizer()
        tokenizer = _default
    out = np.zeros((len(texts), context_length), dtype=np.int64)
    for i, t in enumerate(texts):
        ids = [tokenizer.sot] + tokenizer.encode(t) + [tokenizer.eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = tokenizer.eot
        out[i, :len(ids)] = ids
    import torch
    return torch.from_numpy(out)
