"""Winograd F(2x2,3x3) vs the direct implicit GEMM on the workload's stride-1 3x3 signatures.
python tools/bench_wino.py  -> us per launch, TFLOP/s against the DIRECT algorithmic FLOP."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gan2shape_amd  # noqa
from gan2shape_amd import modconv as mc

SIGS = [  # B, Cin, Cout, H  (G plain layers, D conv1, VGG at B = 9 and B = 2)
    (8, 512, 512, 8), (8, 512, 512, 16), (8, 512, 512, 32), (8, 256, 256, 64), (8, 128, 128, 128),
    (9, 64, 64, 128), (9, 64, 128, 64), (9, 128, 128, 64), (9, 128, 256, 32), (9, 256, 256, 32),
    (9, 256, 512, 16), (9, 512, 512, 16), (9, 512, 512, 8), (2, 64, 64, 128), (2, 128, 128, 64),
    (2, 256, 256, 32), (2, 512, 512, 16),
    (18, 64, 64, 128), (18, 64, 128, 64), (18, 128, 128, 64), (18, 128, 256, 32), (18, 256, 256, 32),
    (18, 256, 512, 16), (18, 512, 512, 16), (16, 128, 128, 128), (16, 256, 256, 64),
]
if os.environ.get("G2S_WINO_SIGS") == "vgg":
    SIGS = [s_ for s_ in SIGS if s_[0] in (9, 18, 16)]


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


splits = [int(v) for v in sys.argv[1:]] or [0]
print(f"{'signature':28s} {'direct us':>10s} {'TF/s':>7s} | " + " | ".join(f"wino sk={s:<2d} us   TF/s" for s in splits))
for B, cin, cout, H in SIGS:
    x = torch.randn(B, cin, H, H, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    s = torch.rand(B, cin, device="cuda") + 0.5
    d = torch.rand(B, cout, device="cuda") + 0.5
    flop = 2.0 * B * cout * cin * 9 * H * H
    mc.WINO_FORCE = "direct"
    t_d = timeit(lambda: mc.modconv_raw(x, w, s, d, mc.PLAIN, 0))
    row = f"{str((B, cin, cout, H)):28s} {t_d:10.1f} {flop / t_d / 1e6:7.1f} | "
    for sk in splits:
        mc.WINO_FORCE = sk
        t_w = timeit(lambda: mc.modconv_raw(x, w, s, d, mc.PLAIN, 0))
        row += f"{t_w:13.1f} {flop / t_w / 1e6:6.1f} | "
    print(row, flush=True)
