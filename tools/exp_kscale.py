"""Experiment: time vs reduction length for 4-tap and 16-tap kernels (fixed M = 256, N = 8*32*32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan2shape_amd  # noqa
from gan2shape_amd.op.conv import _conv2d_raw
from tools.bench_modconv import timeit
B, m, h = 8, 256, 32
for k, pad in [(1, 0), (2, 1), (3, 1), (4, 1)]:
    for cr in (64, 128, 256, 512, 1024, 2048):
        w = torch.randn(m, cr, k, k, device="cuda")
        x = torch.randn(B, cr, h, h, device="cuda")
        f = lambda: _conv2d_raw(x, w, None, cr, m, k, 1, pad, False, True, None, False, 0.0)
        y = f()
        t = timeit(f, 20)
        fl = 2.0 * B * cr * m * k * k * y.shape[2] * y.shape[3]
        print(f"k={k} Cr={cr:5d} K={cr*k*k:6d}: {t:7.1f} us {fl / t / 1e6:6.1f} TF/s", flush=True)
