"""Known-traffic launches for calibrating FETCH_SIZE / WRITE_SIZE on the access widths libg2s uses.
Run under rocprofv3 --pmc.  (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a 16 B/lane stream; other
widths are uncalibrated -> calibrate on a known byte count in your own access pattern.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan2shape_amd
from gan2shape_amd.plugins import fused
from gan2shape_amd.modconv import modconv_raw
dev = torch.device("cuda:0")
e = torch.empty(0, device=dev)
n = 96 * 1024 * 1024   # 384 MiB per tensor: larger than the 256 MiB Infinity Cache
x = torch.randn(n, device=dev)
for _ in range(3):
    fused.fused_bias_act(x.view(1, 1, -1), e, e, 3, 0, 0.2, 1.0)         # fba_f32_vec4: 16 B/lane
xo = torch.randn(n + 1, device=dev)
for _ in range(3):
    fused.fused_bias_act(xo.view(1, 1, -1), e, e, 3, 0, 0.2, 1.0)        # fba_scalar: 4 B/lane
# the three 38.65-GFLOP generator layers, B = 8 (algorithmic bytes: x + y + w)
for cin, h in ((512, 32), (256, 64), (128, 128)):
    xx = torch.randn(8, cin, h, h, device=dev)
    w = torch.randn(cin, cin, 3, 3, device=dev)
    s = torch.rand(8, cin, device=dev) + 0.5
    for _ in range(3):
        modconv_raw(xx, w, s, s, 0, 0)
torch.cuda.synchronize()
