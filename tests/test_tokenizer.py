"""CPU tests of the BPE tokenizer's ALGORITHM on a small synthetic merge table (the upstream
vocabulary file is not available offline: parity with it is unpinned, see cli-p_amd/tokenizer.py)."""
import pytest
import torch


def _tok(clipmi):
    merges = [("a", "b"), ("ab", "c</w>"), ("d", "e</w>"), ("x", "y"), ("xy", "z"), ("xyz", "w</w>"), ("h", "i</w>")]
    return clipmi.tokenizer.SimpleTokenizer(merges=merges)


def test_vocab_layout_and_specials(clipmi):
    t = _tok(clipmi)
    assert len(t.encoder) == 256 + 256 + 7 + 2
    assert t.sot == len(t.encoder) - 2 and t.eot == len(t.encoder) - 1
    assert len(set(clipmi.tokenizer.bytes_to_unicode().values())) == 256


def test_bpe_merges_greedy_by_rank(clipmi):
    t = _tok(clipmi)
    assert t.bpe("abc") == "abc</w>"
    assert t.bpe("abd") == "ab d</w>"
    assert t.bpe("xyzw") == "xyzw</w>"
    assert t.bpe("xyw") == "xy w</w>"
    ids = t.encode("ABC  abc &amp; de")          # lower-cased, whitespace collapsed, html unescaped
    assert [t.decoder[i] for i in ids] == ["abc</w>", "abc</w>", "&</w>", "de</w>"]
    assert t.decode(ids) == "abc abc & de "


def test_tokenize_shapes_padding_and_errors(clipmi):
    t = _tok(clipmi)
    out = clipmi.tokenize(["hi", "abc de hi"], context_length=8, tokenizer=t)
    assert out.dtype == torch.int64 and out.shape == (2, 8)
    assert out[0, 0] == t.sot and out[0, 2] == t.eot and (out[0, 3:] == 0).all()
    assert out[0].argmax() == 2 and out[1].argmax() == 4        # EOT has the highest id: the pooled row
    with pytest.raises(RuntimeError, match="too long"):
        clipmi.tokenize(["hi hi hi hi hi hi hi hi"], context_length=8, tokenizer=t)
    tr = clipmi.tokenize(["hi hi hi hi hi hi hi hi"], context_length=8, tokenizer=t, truncate=True)
    assert tr[0, -1] == t.eot


def test_missing_vocab_is_a_clear_error(clipmi, monkeypatch):
    monkeypatch.delenv("CLIPMI_BPE_PATH", raising=False)
    monkeypatch.setattr(clipmi.tokenizer, "_default", None)
    with pytest.raises(FileNotFoundError, match="CLIPMI_BPE_PATH"):
        clipmi.tokenize(["a photo"])
