import sys, os, time
import torch
sys.path.insert(0, os.getcwd())
import clipmi
dev = torch.device("cuda:0")
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-L/14@336px", seed=0), device=dev)
print("one-round chunk:", model.image_chunk(limit=512), model.image_chunk(limit=256), flush=True)
side = torch.cuda.Stream(device=dev)
def run(B, parts, reps=3):
    x = torch.randint(0, 256, (B, 3, 336, 336), device=dev, dtype=torch.uint8)
    out = torch.empty((B, model.embed_dim), dtype=torch.float32, device=dev)
    model.chunks_in_flight = 1
    def step():
        if parts == 1:
            model.encode_image(x, normalize=True, out=out)
        else:
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            h = B // 2
            model.encode_image(x[:h], normalize=True, out=out[:h])
            with torch.cuda.stream(side):
                model.encode_image(x[h:], normalize=True, out=out[h:])
            cur.wait_stream(side)
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return B * reps / (time.perf_counter() - t0)
for B in (266, 256, 510):
    for rnd in range(2):
        print(f"ViT-L/14@336 B={B}: one sequence {run(B,1):.0f} img/s, two halves in flight {run(B,2):.0f} img/s", flush=True)
