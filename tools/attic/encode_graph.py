"""Does replaying the encode kernel sequence as one HIP graph beat eager launches? (development aid)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
dev = torch.device("cuda:0")
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 435
x = torch.randint(0, 256, (B, 3, 224, 224), device=dev, dtype=torch.uint8)
for _ in range(3): ref = model.encode_image(x, normalize=True)
torch.cuda.synchronize()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timeit(lambda: model.encode_image(x, normalize=True))
print(f"eager: {ms:.3f} ms  {B/ms*1e3:.0f} img/s", flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    model.encode_image(x, normalize=True)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = model.encode_image(x, normalize=True)
ms = timeit(lambda: g.replay())
print(f"graph: {ms:.3f} ms  {B/ms*1e3:.0f} img/s  equal={torch.equal(out, ref)}", flush=True)
