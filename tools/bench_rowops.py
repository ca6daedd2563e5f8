import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan2shape_amd
from gan2shape_amd.modconv import rows_dot_scale, demodulation
from tools.bench_modconv import timeit
B = 8
for C, H in [(512, 4), (512, 9), (512, 16), (512, 33), (512, 32), (256, 65), (256, 64), (128, 129), (128, 128)]:
    a = torch.randn(B, C, H, H, device="cuda"); b = torch.randn_like(a)
    s = torch.randn(B, C, device="cuda")
    t1 = timeit(lambda: rows_dot_scale(a, b, s, None))
    t2 = timeit(lambda: ((a * b).sum((2, 3)), b * s[:, :, None, None]))
    print(f"rows_dot_scale C={C} H={H}: fused {t1:7.1f} us  torch {t2:7.1f} us  ({3*a.numel()*4/t1/1e6:.2f} TB/s)")
for cin, cout in [(512, 512), (512, 256), (256, 256), (256, 128), (128, 128)]:
    s = (torch.randn(B, cin, device="cuda") + 1).requires_grad_(True)
    wsq = torch.rand(cout, cin, device="cuda")
    d = demodulation(s, wsq); g = torch.randn_like(d)
    t1 = timeit(lambda: demodulation(s, wsq))
    t2 = timeit(lambda: torch.autograd.grad(d, s, g, retain_graph=True))
    print(f"demod {cin}->{cout}: fwd {t1:6.1f} us  bwd {t2:6.1f} us")
