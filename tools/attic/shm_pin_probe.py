"""Write rate into a shared-memory segment before / after hipHostRegister, and the H2D rate out of it (development aid)."""
import ctypes, time, numpy as np, torch
from multiprocessing import shared_memory
n = 700 << 20
seg = shared_memory.SharedMemory(create=True, size=n)
a = np.frombuffer(seg.buf, np.uint8)
src = np.random.default_rng(0).integers(0, 256, 1 << 20, dtype=np.uint8)
def fill():
    t0 = time.time()
    for k in range(0, n, 1 << 20): a[k:k + (1 << 20)] = src
    return n / (time.time() - t0) / 1e9
print("first touch GB/s %.2f" % fill()); print("second pass GB/s %.2f" % fill())
torch.cuda.init(); dev = torch.device("cuda:0")
addr = ctypes.addressof(ctypes.c_char.from_buffer(seg.buf))
t0 = time.time(); rc = torch.cuda.cudart().cudaHostRegister(addr, n, 0); print("register rc", int(rc), "%.0f ms" % ((time.time() - t0) * 1e3))
print("write after register GB/s %.2f" % fill())
t = torch.from_numpy(a)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.time(); d = t.to(dev, non_blocking=True); torch.cuda.synchronize(); print("H2D GB/s %.1f" % (n / (time.time() - t0) / 1e9))
p = torch.empty(n, dtype=torch.uint8).pin_memory()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time(); d = p.to(dev, non_blocking=True); torch.cuda.synchronize(); print("torch-pinned H2D GB/s %.1f" % (n / (time.time() - t0) / 1e9))
torch.cuda.cudart().cudaHostUnregister(addr); del a, t; seg.close(); seg.unlink()
