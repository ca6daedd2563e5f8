"""Per-layer timing of g2s_modconv on the StyleGAN2-128 generator shapes (B = 8), forward and
data-gradient.  python tools/bench_modconv.py [B] [f16|f32] [direct]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan2shape_amd  # noqa
from gan2shape_amd.modconv import modconv_raw

LAYERS = [  # name, Cin, Cout, H, k, mode
    ("conv1   4x4", 512, 512, 4, 3, 0), ("convs0 up 4", 512, 512, 4, 3, 1), ("convs1  8x8", 512, 512, 8, 3, 0),
    ("convs2 up 8", 512, 512, 8, 3, 1), ("convs3 16", 512, 512, 16, 3, 0), ("convs4 up16", 512, 512, 16, 3, 1),
    ("convs5 32", 512, 512, 32, 3, 0), ("convs6 up32", 512, 256, 32, 3, 1), ("convs7 64", 256, 256, 64, 3, 0),
    ("convs8 up64", 256, 128, 64, 3, 1), ("convs9 128", 128, 128, 128, 3, 0), ("torgb 128", 128, 3, 128, 1, 0),
]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from gan2shape_amd import modconv as mc
    if "f16" in sys.argv[2:]:
        mc.OPERANDS = "f16"       # fp16 MFMA operands (BASELINE config 5)
    if "direct" in sys.argv[2:]:
        mc.WINOGRAD = False       # the direct implicit-GEMM kernel everywhere
    print(f"B = {B}, operands {mc.OPERANDS}, winograd {mc.WINOGRAD}")
    tot_t = tot_f = 0
    for name, cin, cout, h, k, mode in LAYERS:
        x = torch.randn(B, cin, h, h, device="cuda")
        w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
        s = torch.rand(B, cin, device="cuda") + 0.5
        dm = torch.rand(B, cout, device="cuda") + 0.5
        y = modconv_raw(x, w, s, dm, mode, 0)
        g = torch.randn_like(y)
        flop = 2.0 * B * cin * cout * k * k * h * h
        tf = timeit(lambda: modconv_raw(x, w, s, dm, mode, 0))
        tb = timeit(lambda: modconv_raw(g, w, dm, None, mode, 1))
        tot_t += tf + tb
        tot_f += 2 * flop
        print(f"{name:12s} {flop/1e9:7.2f} GF  fwd {tf:8.1f} us {flop/tf/1e6:6.1f} TF/s   bwd {tb:8.1f} us {flop/tb/1e6:6.1f} TF/s")
    print(f"total {tot_t/1e3:.2f} ms  {tot_f/tot_t/1e6:.1f} TF/s")


if __name__ == "__main__":
    main()
