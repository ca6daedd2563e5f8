"""One signature, N launches of one Winograd kernel (profiling aid).  python tools/w4_one.py f4|f2 B C M H [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gan2shape_amd  # noqa
from gan2shape_amd import modconv as mc

kind, B, cin, cout, H = sys.argv[1], *[int(v) for v in sys.argv[2:6]]
N = int(sys.argv[6]) if len(sys.argv) > 6 else 20
torch.manual_seed(0)
x = torch.randn(B, cin, H, H, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
s = torch.rand(B, cin, device="cuda") + 0.5
d = torch.rand(B, cout, device="cuda") + 0.5
mc.WINO_FORCE = "w4:0" if kind == "f4" else 0
for _ in range(N):
    mc.modconv_raw(x, w, s, d, mc.PLAIN, 0)
torch.cuda.synchronize()
print("done")
