import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan2shape_amd
from gan2shape_amd.lpips import PerceptualLoss
mode = sys.argv[1]
if "bench" in mode:
    torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
p = PerceptualLoss().to(dev)
if "nhwc" in mode:
    p = p.to(memory_format=torch.channels_last)
for B in (1, 8):
    a = torch.rand(B, 3, 128, 128, device=dev, requires_grad=True)
    b = torch.rand(B, 3, 128, 128, device=dev)
    if "nhwc" in mode:
        b = b.contiguous(memory_format=torch.channels_last)
    def f():
        a.grad = None
        x = a.contiguous(memory_format=torch.channels_last) if "nhwc" in mode else a
        p(x, b).mean().backward()
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / 10
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 10
    print(f"{mode:12s} B={B} eager {te*1e3:7.2f} ms  graph {tg*1e3:7.2f} ms", flush=True)
