"""Two encoders on two HIP streams pinned to disjoint halves of the CUs (hipExtStreamCreateWithCUMask):
does one half's HBM-bound kernels / epilogue bursts overlap the other half's MFMA mainloops? (development aid)"""
import sys, os, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
sd = clipmi.weights.random_state_dict("ViT-B/32", seed=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 435
models = [clipmi.CLIP(sd, device=dev) for _ in range(2)]
xs = [torch.randint(0, 256, (B, 3, 224, 224), device=dev, dtype=torch.uint8) for _ in range(2)]
outs = [torch.empty((B, 512), dtype=torch.float32, device=dev) for _ in range(2)]
for m, x, o in zip(models, xs, outs):
    m.encode_image(x, normalize=True, out=o)
torch.cuda.synchronize()

def masked_streams(words):
    ss = []
    for w in words:
        s = C.c_void_p()
        arr = (C.c_uint32 * 8)(*w)
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
        assert rc == 0, rc
        ss.append(s)
    return ss

def run(streams, reps):
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for r in range(reps):
        for i, s in enumerate(streams):
            models[i].encode_image(xs[i], normalize=True, out=outs[i], stream=s.value)
    for s in streams:
        hip.hipStreamSynchronize(s)
    return time.perf_counter() - t0

patterns = {
    "all|all": ([0xFFFFFFFF] * 8, [0xFFFFFFFF] * 8),
    "low16|high16 of each word": ([0x0000FFFF] * 8, [0xFFFF0000] * 8),
    "even|odd bits": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
    "words 0-3|4-7": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
    "even|odd words": ([0xFFFFFFFF, 0] * 4, [0, 0xFFFFFFFF] * 4),
}
for name, (a, b) in patterns.items():
    ss = masked_streams([a, b])
    run(ss, 2)
    dt = run(ss, 8)
    print(f"B={B} {name}: {2*8*B/dt:.0f} img/s", flush=True)
one = masked_streams([[0xFFFFFFFF] * 8])
run(one, 2); dt = run(one, 16)
print(f"B={B} single stream: {16*B/dt:.0f} img/s", flush=True)
