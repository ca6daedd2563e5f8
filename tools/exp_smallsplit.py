import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan2shape_amd  # noqa
from gan2shape_amd import lib
from gan2shape_amd.op.conv import _conv2d_raw
torch.cuda.set_stream(torch.cuda.Stream())


def timeit(fn, n=20):
    """GPU time per call with the calls replayed from a HIP graph (no host launch cost)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
L = lib.load()
LAYERS = [(1, 32, 64, 64, 4, 2, 1, 0), (1, 64, 128, 32, 4, 2, 1, 0), (1, 128, 256, 16, 4, 2, 1, 0), (1, 256, 256, 4, 3, 1, 1, 0),
          (1, 128, 128, 8, 3, 1, 1, 0), (1, 32, 32, 32, 3, 1, 1, 0), (1, 32, 32, 128, 3, 1, 1, 0), (1, 256, 256, 4, 4, 1, 0, 0),
          (9, 64, 128, 32, 4, 2, 1, 0), (9, 256, 512, 8, 4, 2, 1, 0)]
e = torch.empty(1 << 20, device="cuda")
print("memset-like fill of 1 MiB:", timeit(lambda: e.zero_(), 50), "us; empty launch floor")
for B, cin, cout, h, k, s, p, tr in LAYERS:
    w = torch.randn((cout, cin, k, k), device="cuda")
    x = torch.randn(B, cin, h, h, device="cuda")
    f = lambda: _conv2d_raw(x, w, None, cin, cout, k, s, p, False, True, None, False, 0.0)
    L.g2s_modconv_tune(-1, -1)
    base = timeit(f, 30)
    out = []
    for tile in (2, 1):
        for sk in (1, 2, 4, 8, 16, 32, 64):
            L.g2s_modconv_tune(tile, sk)
            out.append(f"t{tile}sk{sk}:{timeit(f, 30):5.1f}")
    L.g2s_modconv_tune(-1, -1)
    print(f"B={B} {cin}->{cout} {h}^2 k{k}s{s}: default {base:5.1f} | " + " ".join(out), flush=True)
