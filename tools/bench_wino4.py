"""Winograd F(4x4,3x3) (csrc/winograd4.hip) against F(2x2,3x3) and the direct implicit GEMM on the workload's large
stride-1 3x3 signatures: us per launch, TFLOP/s of DIRECT algorithmic FLOP, and the error of each kernel against a
float64 convolution of sample 0 (CPU).
    python tools/bench_wino4.py [splitk ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import gan2shape_amd  # noqa
from gan2shape_amd import modconv as mc

SIGS = [  # B, Cin, Cout, H
    (8, 128, 128, 128), (16, 128, 128, 128), (8, 256, 256, 64), (16, 256, 256, 64), (8, 512, 512, 32),
    (16, 512, 512, 32), (8, 512, 512, 16), (16, 512, 512, 16),
    (18, 64, 64, 128), (18, 64, 128, 64), (18, 128, 128, 64), (18, 128, 256, 32), (18, 256, 256, 32),
    (9, 64, 64, 128), (9, 128, 128, 64), (9, 256, 256, 32), (2, 64, 64, 128), (2, 128, 128, 64),
]
if os.environ.get("G2S_W4_SIGS"):
    keep = [int(v) for v in os.environ["G2S_W4_SIGS"].split(",")]
    SIGS = [SIGS[i] for i in keep]


def timeit(fn, reps=20):
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
print(f"{'signature':24s} {'direct':>8s} {'F2 us':>8s} {'TF/s':>6s} | " +
      " | ".join(f"F4 sk={s:<2d} us   TF/s  x F2" for s in splits) + " | err/max|y|: direct      F2       F4")
torch.manual_seed(0)
for B, cin, cout, H in SIGS:
    x = torch.randn(B, cin, H, H, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    s = torch.rand(B, cin, device="cuda") + 0.5
    d = torch.rand(B, cout, device="cuda") + 0.5
    flop = 2.0 * B * cout * cin * 9 * H * H
    ref = (F.conv2d((x[:1] * s[:1, :, None, None]).double().cpu(), w.double().cpu(), padding=1) *
           d[:1, :, None, None].double().cpu())
    scale = ref.abs().max().item()

    def err(y):
        return (y[:1].double().cpu() - ref).abs().max().item() / scale

    def run():
        return mc.modconv_raw(x, w, s, d, mc.PLAIN, 0)

    mc.WINO_FORCE = "direct"
    t_d, e_d = timeit(run), err(run())
    mc.WINO_FORCE = 0
    t_2, e_2 = timeit(run), err(run())
    row = f"{str((B, cin, cout, H)):24s} {t_d:8.1f} {t_2:8.1f} {flop / t_2 / 1e6:6.1f} | "
    e_4 = float("nan")
    for sk in splits:
        mc.WINO_FORCE = f"w4:{sk}"
        if not mc.wino4_supported(B, cin, cout, H, H):
            row += f"{'unsupported':>32s} | "
            continue
        t_4, e_4 = timeit(run), err(run())
        row += f"{t_4:13.1f} {flop / t_4 / 1e6:6.1f} {t_2 / t_4:5.2f} | "
    print(row + f"          {e_d:8.1e} {e_2:8.1e} {e_4:8.1e}", flush=True)
