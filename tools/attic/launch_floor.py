"""Per-kernel cost of a chain of tiny dependent kernels on one stream (development aid): the floor a one-prompt tower pays."""
import sys, os, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
dev = torch.device("cuda:0")
L = clipmi._lib.lib()
x = torch.randn(4, 512, device=dev)
sp = clipmi._lib.stream_ptr(dev)
def chain(n):
    for _ in range(n): L.clipmi_l2_normalize_rows(x.data_ptr(), 4, 512, sp)
for n in (100,):
    chain(20); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); chain(n); e1.record(); torch.cuda.synchronize()
    print(f"eager chain of {n} l2_normalize_rows(4 x 512): {e0.elapsed_time(e1) / n * 1e3:.2f} us per kernel")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): chain(n)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"graph replay of the same chain: {e0.elapsed_time(e1) / n * 1e3:.2f} us per kernel")
# torch's own tiny kernel for comparison
y = torch.zeros(64, device=dev)
for _ in range(20): y.add_(1.0)
torch.cuda.synchronize()
e0.record()
for _ in range(100): y.add_(1.0)
e1.record(); torch.cuda.synchronize()
print(f"eager chain of 100 torch add_: {e0.elapsed_time(e1) / 100 * 1e3:.2f} us per kernel")
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    for _ in range(100): y.add_(1.0)
g2.replay(); torch.cuda.synchronize()
e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
print(f"graph replay of 100 torch add_: {e0.elapsed_time(e1) / 100 * 1e3:.2f} us per kernel")
