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
    # round 5: the statistics real CLIP residual streams show, which seeded weights do not - a few
    # channels 50-200 x the median magnitude at a FEW TOKENS only (massive activations: CLS / a couple of patch positions, SOT
    # on the text side), LayerNorm gains that suppress exactly those channels, and a residual norm that grows ~6 x over the
    # 12 layers - through both towers' LN-folded un-normalised-bf16(x) operand path and the FP8 tower
    "vitb32_realstats": ("ViT-B/32", 5, "realstats"),
}


def state_dict(name):
    import clipmi
    arch, seed, outlier = CASES[name]
    sd = clipmi.weights.random_state_dict(arch, seed=seed)
    if outlier == "realstats":
        _realstats(sd)
    elif outlier:
        # a few large residual-stream channels, as real CLIP checkpoints have (bf16 range test)
        for tower, W in (("visual.transformer", sd["visual.ln_pre.weight"].shape[0]),
                         ("transformer", sd["ln_final.weight"].shape[0])):
            b = sd[f"{tower}.resblocks.0.attn.out_proj.bias"]
            for c in (5, W // 3, W - 7):
                b[c] += 40.0
    return sd


def _realstats(sd):
    """In place: see CASES["vitb32_realstats"]. Measured on the fp32 oracle (tests/test_oracle_clip.py pins the numbers):
    largest |x| / median |x| of the residual stream 60-230 over the layers, on 3 of the 50 image tokens (1 of 77 text
    positions); mean row norm x 6 from the first block's output to the last's."""
    for tower, pos, ln0, toks in (("visual.transformer", "visual.positional_embedding", "visual.ln_pre", (0, 7, 23)),
                                  ("transformer", "positional_embedding", None, (0,))):
        W = sd[f"{tower}.resblocks.0.ln_1.weight"].shape[0]
        chans = (5, W // 3, W - 7)
        n = 0
        while f"{tower}.resblocks.{n}.ln_1.weight" in sd:
            n += 1
        for t in toks:                                   # token-specific massive channels, present from the embedding on
            for j, c in enumerate(chans):
                sd[pos][t, c] += (30.0, -24.0, 18.0)[j] * (1.0 if ln0 else 3.0)
        if ln0:
            for c in chans:
                sd[ln0 + ".weight"][c] = 6.0
        for l in range(n):
            p = f"{tower}.resblocks.{l}"
            grow = 1.2 ** l                              # the branches write more and more into the stream
            for k in ("attn.out_proj", "mlp.c_proj"):
                sd[f"{p}.{k}.weight"] *= grow
                sd[f"{p}.{k}.bias"] *= grow
            sd[f"{p}.mlp.c_proj.bias"][chans[0]] += 6.0 * grow       # one channel large at EVERY token as well, and growing
            for ln in ("ln_1", "ln_2"):
                for c in chans:
                    sd[f"{p}.{ln}.weight"][c] = 0.08 if l % 2 == 0 else 0.3     # learned suppression of the massive channels


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
