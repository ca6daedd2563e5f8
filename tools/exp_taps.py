"""Experiment: MFMA conv throughput by tap count with dense weights (k=1,2,3,4,5) at one size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan2shape_amd  # noqa
from gan2shape_amd.op.conv import _conv2d_raw
from tools.bench_modconv import timeit
for (B, cr, m, h) in [(8, 512, 256, 32), (8, 256, 128, 64), (8, 512, 512, 16)]:
    for k, pad in [(1, 0), (2, 1), (3, 1), (4, 1), (5, 2)]:
        for mm in (1, 0):
            w = torch.randn(m, cr, k, k, device="cuda") if mm else torch.randn(cr, m, k, k, device="cuda")
            x = torch.randn(B, cr, h, h, device="cuda")
            f = lambda: _conv2d_raw(x, w, None, cr, m, k, 1, pad, False, mm, None, False, 0.0)
            y = f()
            t = timeit(f, 20)
            fl = 2.0 * B * cr * m * k * k * y.shape[2] * y.shape[3]
            print(f"B={B} {cr}->{m} {h}^2 k={k} m_major={mm}: {t:7.1f} us {fl / t / 1e6:6.1f} TF/s", flush=True)
