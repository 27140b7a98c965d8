"""CPU checks that pin oracle/clip_oracle.py: its outputs on seeded weights equal the committed
golden vectors, which an independent implementation (transformers' CLIP classes) produced —
tests/golden/make_clip_golden.py. Tolerance 2e-5 absolute on outputs of magnitude ~10 (fp32,
different op order)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLD)
sys.path.insert(0, ROOT)
import clip_case  # noqa: E402
from oracle import clip_oracle  # noqa: E402


@pytest.mark.parametrize("name", list(clip_case.CASES))
def test_oracle_matches_independent_golden(name):
    g = np.load(os.path.join(GOLD, f"clip_{name}.npz"))
    assert str(g["torch_version"]) == torch.__version__, "goldens were made with another torch: regenerate"
    sd = clip_case.state_dict(name)
    images, ids = clip_case.inputs(name)
    assert float(images.double().sum().item()) == float(g["image_checksum"])
    assert np.array_equal(ids.numpy(), g["ids"])
    img = clip_oracle.encode_image(sd, images).numpy()
    txt = clip_oracle.encode_text(sd, ids).numpy()
    assert np.abs(img).max() > 1.0 and np.abs(txt).max() > 1.0         # not vacuous
    assert np.abs(img - g["image_embeds"]).max() < 2e-5 * max(1.0, np.abs(img).max())
    assert np.abs(txt - g["text_embeds"]).max() < 2e-5 * max(1.0, np.abs(txt).max())


def test_text_rows_after_eot_do_not_matter():
    """Causal mask: tokens after EOT cannot influence the pooled row."""
    sd = clip_case.state_dict("toy_seed0")
    _, ids = clip_case.inputs("toy_seed0")
    a = clip_oracle.encode_text(sd, ids)
    ids2 = ids.clone()
    ids2[0, 6:] = 3          # junk after EOT at position 5 (all < EOT id)
    b = clip_oracle.encode_text(sd, ids2)
    assert torch.equal(a[0], b[0])


def test_normalize_helpers():
    x = torch.tensor([[3.0, 4.0], [0.0, 2.0]])
    assert torch.allclose(clip_oracle.normalize_rows(x), torch.tensor([[0.6, 0.8], [0.0, 1.0]]))
    z = np.zeros((1, 4), np.float32)
    assert clip_oracle.normalize_query(z) is z
