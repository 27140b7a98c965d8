"""smoke() helper: one tiny encode_image + encode_text on cuda:0 checked against the oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def smoke_encode(dev):
    import clipmi
    import clip_case
    from oracle import clip_oracle
    sd = clip_case.state_dict("vitb32_seed0")
    images, ids = clip_case.inputs("vitb32_seed0")
    model = clipmi.CLIP(sd, device=dev)
    sdr = clipmi.weights.bf16_round_state_dict(sd)
    got = model.encode_image(images[:2]).cpu()
    ref = clip_oracle.encode_image(sdr, images[:2])
    cos = torch.nn.functional.cosine_similarity(got.double(), ref.double(), dim=-1).min().item()
    assert cos >= 0.9995, f"encode_image parity: cosine {cos}"
    got = model.encode_text(ids[:1]).cpu()
    ref = clip_oracle.encode_text(sdr, ids[:1])
    cos = torch.nn.functional.cosine_similarity(got.double(), ref.double(), dim=-1).min().item()
    assert cos >= 0.9995, f"encode_text parity: cosine {cos}"
