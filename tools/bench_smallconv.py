"""Latency of the small trained nets' convolution launches (B = 1 depth/albedo nets, B = 9
viewpoint/lighting nets): forward, data-gradient, weight-gradient; back-to-back launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan2shape_amd  # noqa
from gan2shape_amd.op.conv import _conv2d_raw, _wgrad
from tools.bench_modconv import timeit
# B, Cin, Cout, H, k, stride, pad, transposed
LAYERS = [(1, 3, 32, 128, 4, 2, 1, 0), (1, 32, 64, 64, 4, 2, 1, 0), (1, 64, 128, 32, 4, 2, 1, 0), (1, 128, 256, 16, 4, 2, 1, 0),
          (1, 256, 256, 4, 4, 1, 0, 0), (1, 256, 256, 1, 4, 1, 0, 1), (1, 256, 256, 4, 3, 1, 1, 0), (1, 256, 128, 4, 4, 2, 1, 1),
          (1, 128, 128, 8, 3, 1, 1, 0), (1, 128, 64, 8, 4, 2, 1, 1), (1, 64, 64, 16, 3, 1, 1, 0), (1, 64, 32, 16, 4, 2, 1, 1),
          (1, 32, 32, 32, 3, 1, 1, 0), (1, 32, 32, 128, 3, 1, 1, 0), (1, 32, 32, 128, 5, 1, 2, 0), (1, 32, 3, 128, 5, 1, 2, 0),
          (9, 3, 32, 128, 4, 2, 1, 0), (9, 32, 64, 64, 4, 2, 1, 0), (9, 64, 128, 32, 4, 2, 1, 0), (9, 128, 256, 16, 4, 2, 1, 0),
          (9, 256, 512, 8, 4, 2, 1, 0), (9, 512, 512, 4, 4, 1, 0, 0), (9, 512, 6, 1, 1, 1, 0, 0)]
tot = [0, 0, 0]
for B, cin, cout, h, k, s, p, tr in LAYERS:
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device="cuda")
    x = torch.randn(B, cin, h, h, device="cuda")
    fwd = lambda: _conv2d_raw(x, w, None, cin, cout, k, s, p, bool(tr), not tr, None, False, 0.0)
    y = fwd()
    gy = torch.randn_like(y)
    dg = lambda: _conv2d_raw(gy, w, None, cout, cin, k, s, p, not tr, bool(tr), (h, h), False, 0.0)
    wg = (lambda: _wgrad(x, gy, k, s, p)) if tr else (lambda: _wgrad(gy, x, k, s, p))
    t = [timeit(f, 30) for f in (fwd, dg, wg)]
    for i in range(3):
        tot[i] += t[i]
    fl = 2.0 * B * cin * cout * k * k * (h * h if tr else y.shape[2] * y.shape[3])
    print(f"B={B} {cin:3d}->{cout:3d} {h:3d}^2 k{k} s{s} p{p} {'T' if tr else ' '} {fl / 1e6:8.1f} MFLOP: fwd {t[0]:6.1f}  dgrad {t[1]:6.1f}  wgrad {t[2]:6.1f} us", flush=True)
print(f"total fwd {tot[0]:.0f} dgrad {tot[1]:.0f} wgrad {tot[2]:.0f} us")
