"""MIOpen F.conv2d vs g2s_modconv on the VGG16 (LPIPS) convolution shapes at 128x128 input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import gan2shape_amd
from gan2shape_amd.modconv import modconv_raw
from tools.bench_modconv import timeit
for B in (1, 9):
    tm = tg = 0
    for cin, cout, h, reps in [(3, 64, 128, 1), (64, 64, 128, 1), (64, 128, 64, 1), (128, 128, 64, 1), (128, 256, 32, 1),
                               (256, 256, 32, 2), (256, 512, 16, 1), (512, 512, 16, 2), (512, 512, 8, 3)]:
        x = torch.randn(B, cin, h, h, device="cuda", requires_grad=True)
        w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
        y = F.conv2d(x, w, padding=1); g = torch.randn_like(y)
        t_f = timeit(lambda: F.conv2d(x, w, padding=1))
        t_b = timeit(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
        m_f = timeit(lambda: modconv_raw(x.detach(), w, None, None, 0, 0))
        m_b = timeit(lambda: modconv_raw(g, w, None, None, 0, 1))
        tm += reps * (2 * t_f + t_b); tg += reps * (2 * m_f + m_b)
        print(f"B={B} {cin:3d}->{cout:3d}@{h:3d} | MIOpen fwd {t_f:7.1f} bwd {t_b:7.1f} | g2s fwd {m_f:7.1f} bwd {m_b:7.1f} us")
    print(f"B={B}: 2 fwd + 1 bwd over VGG16: MIOpen {tm/1e3:.2f} ms, g2s {tg/1e3:.2f} ms")
