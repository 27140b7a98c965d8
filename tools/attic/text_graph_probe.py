"""encode_text latency at Q = 1: eager launches vs one HIP graph replay (development aid). Measured r02: 0.878 ms eager,
0.885 ms replayed - the ~100 dependent kernels of a one-prompt text tower are bound by kernel boundaries on the GPU, not by
host launch cost, so graph capture buys nothing here."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
dev = torch.device("cuda:0")
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
def timeit(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for Q in (1,):
    ids = torch.zeros(Q, 77, dtype=torch.int64)
    ids[:, 0] = 49406; ids[:, 1:9] = torch.randint(1, 40000, (Q, 8)); ids[:, 9] = 49407
    ids = ids.to(dev)
    ref = model.encode_text(ids, normalize=True)
    ms = timeit(lambda: model.encode_text(ids, normalize=True))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model.encode_text(ids, normalize=True)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = model.encode_text(ids, normalize=True)
    msg = timeit(lambda: g.replay())
    print(f"encode_text Q={Q}: eager {ms:.3f} ms, graph {msg:.3f} ms, equal={torch.equal(out, ref)}", flush=True)
