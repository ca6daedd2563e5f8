"""MIOpen F.conv2d vs g2s_modconv on the discriminator's convolution shapes (B = 8), fwd and bwd-data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import gan2shape_amd
from gan2shape_amd.modconv import modconv_raw
from tools.bench_modconv import timeit
B = 8
for name, cin, cout, h, k, stride in [("conv1 128@128", 128, 128, 128, 3, 1), ("conv2 s2 128->256", 128, 256, 131, 3, 2),
                                      ("conv1 256@64", 256, 256, 64, 3, 1), ("conv2 s2 256->512", 256, 512, 67, 3, 2),
                                      ("conv1 512@32", 512, 512, 32, 3, 1), ("conv2 s2 512@35", 512, 512, 35, 3, 2),
                                      ("conv1 512@16", 512, 512, 16, 3, 1), ("conv2 s2 512@19", 512, 512, 19, 3, 2),
                                      ("1x1 3->128", 3, 128, 128, 1, 1)]:
    x = torch.randn(B, cin, h, h, device="cuda", requires_grad=True)
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    pad = k // 2 if stride == 1 else 0
    y = F.conv2d(x, w, stride=stride, padding=pad)
    g = torch.randn_like(y)
    t_f = timeit(lambda: F.conv2d(x, w, stride=stride, padding=pad))
    t_b = timeit(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]))
    mode = 0 if stride == 1 else 2
    m_f = timeit(lambda: modconv_raw(x.detach(), w, None, None, mode, 0))
    m_b = timeit(lambda: modconv_raw(g, w, None, None, mode, 1))
    flop = 2.0 * B * cin * cout * k * k * y.shape[2] * y.shape[3]
    print(f"{name:20s} {flop/1e9:6.1f} GF | MIOpen fwd {t_f:7.1f} bwd {t_b:7.1f} us | g2s fwd {m_f:7.1f} bwd {m_b:7.1f} us")
