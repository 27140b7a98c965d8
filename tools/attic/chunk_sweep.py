"""encode_image over batch sizes with and without whole-round chunking (development aid)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
dev = torch.device("cuda:0")
def t_ms(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
g = torch.Generator(device=dev); g.manual_seed(0)
for name, res, Bs in (("ViT-B/32", 224, (300, 436, 512, 600, 700, 800, 871, 1000, 1024, 1305)), ("ViT-L/14@336px", 336, (255, 266, 283, 300))):
    model = clipmi.CLIP(clipmi.weights.random_state_dict(name, seed=0), device=dev)
    for B in Bs:
        x = torch.randint(0, 256, (B, 3, res, res), generator=g, device=dev, dtype=torch.uint8)
        row = []
        for rc in (False, True):
            model.round_chunks = rc
            ms = t_ms(lambda: model.encode_image(x, normalize=True), 6 if res == 224 else 2)
            row.append(f"{'rounds' if rc else 'single'} {model.image_chunks(B)} {ms:.2f} ms {B / ms * 1e3:,.0f}/s")
        print(f"{name} B={B}: " + " | ".join(row), flush=True)
    del model
    torch.cuda.empty_cache()
