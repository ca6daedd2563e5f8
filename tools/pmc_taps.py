"""For rocprofv3 --pmc: the MFMA conv kernel on one size with 1, 4, 9, 16 taps (dense weights), 3 launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan2shape_amd  # noqa
from gan2shape_amd.op.conv import _conv2d_raw
B, cr, m, h = 8, 512, 256, 32
for k, pad in [(1, 0), (2, 1), (3, 1), (4, 1)]:
    w = torch.randn(m, cr, k, k, device="cuda")
    x = torch.randn(B, cr, h, h, device="cuda")
    for _ in range(3):
        _conv2d_raw(x, w, None, cr, m, k, 1, pad, False, True, None, False, 0.0)
    torch.cuda.synchronize()
