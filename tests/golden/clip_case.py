"""Deterministic inputs of the CLIP golden cases (SURVEY.md §8c fixtures (i)-(iii),(v)): weights
and inputs are rebuilt from seeds (torch CPU generator, torch version pinned by the image); only
expected outputs are stored in clip_*.npz by make_clip_golden.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASES = {
    # name: (arch, weight seed, outlier channels?)
    "vitb32_seed0": ("ViT-B/32", 0, False),
    "vitb32_outlier": ("ViT-B/32", 1, True),
    "toy_seed0": ("toy", 0, False),
    # round 2: the geometry paths that had no independent golden — 14-px patches (K = 588 padded to 640) with
    # 101 tokens (flash attention), and ViT-B/16 (16-px patches, 197 tokens)
    "toyl14_seed3": ("toy-l14", 3, False),
    "vitb16_seed2": ("ViT-B/16", 2, False),
    # ViT-L/14 (width 1024, 16 heads, 24 layers, 257 tokens; text width 768, 12 heads): two images, three prompts
    "vitl14_seed4": ("ViT-L/14", 4, False),
}


def state_dict(name):
    import clipmi
    arch, seed, outlier = CASES[name]
    sd = clipmi.weights.random_state_dict(arch, seed=seed)
    if outlier:
        # a few large residual-stream channels, as real CLIP checkpoints have (bf16 range test)
        for tower, W in (("visual.transformer", sd["visual.ln_pre.weight"].shape[0]),
                         ("transformer", sd["ln_final.weight"].shape[0])):
            b = sd[f"{tower}.resblocks.0.attn.out_proj.bias"]
            for c in (5, W // 3, W - 7):
                b[c] += 40.0
    return sd


def inputs(name):
    import clipmi
    arch = CASES[name][0]
    a = clipmi.weights.ARCHS[arch]
    g = torch.Generator(device="cpu")
    g.manual_seed(1234)
    images = torch.randn(2 if arch.startswith("ViT-L") else 4, 3, a["res"], a["res"], generator=g, dtype=torch.float32)
    ctx, vocab = a["ctx"], a["vocab"]
    ids = torch.zeros(3, ctx, dtype=torch.int64)
    for r, eot in enumerate((5, min(20, ctx - 2), ctx - 1)):
        ids[r, 0] = vocab - 2                                   # <|startoftext|>
        ids[r, 1:eot] = torch.randint(1, vocab - 2, (eot - 1,), generator=g)
        ids[r, eot] = vocab - 1                                 # <|endoftext|>, the highest id
    return images, ids
