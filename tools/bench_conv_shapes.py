"""Time g2s_modconv on an explicit list of shapes (the slow entries of tools/conv_census.py).
Each spec: B Cin Cout H k mode transpose   (H = input side of the call)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan2shape_amd  # noqa
from gan2shape_amd.modconv import modconv_raw
from tools.bench_modconv import timeit

SHAPES = [
    (8, 512, 512, 16, 3, 2, 1), (8, 256, 512, 32, 3, 2, 1), (8, 128, 256, 64, 3, 2, 1), (8, 512, 512, 8, 3, 2, 1),
    (8, 256, 128, 64, 3, 1, 0), (8, 512, 256, 32, 3, 1, 0), (8, 512, 512, 16, 3, 1, 0), (8, 512, 512, 8, 3, 1, 0),
    (8, 512, 512, 33, 3, 2, 0), (8, 512, 512, 17, 3, 2, 0), (8, 512, 512, 16, 3, 0, 0), (8, 512, 512, 8, 3, 0, 0),
    (8, 128, 256, 127, 1, 2, 0), (8, 256, 512, 63, 1, 2, 0), (8, 512, 512, 31, 1, 2, 0), (8, 128, 256, 64, 1, 2, 1),
    (9, 256, 256, 32, 3, 0, 0), (9, 512, 512, 16, 3, 0, 0), (9, 512, 512, 8, 3, 0, 0), (9, 64, 64, 128, 3, 0, 0),
]


def main():
    for B, cin, cout, h, k, mode, tr in SHAPES:
        w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
        x = torch.randn(B, cout if tr else cin, h, h, device="cuda")
        y = modconv_raw(x, w, None, None, mode, tr)
        oh = y.shape[-1]
        sp = h * h if mode == 0 else min(h * h, oh * oh)
        flop = 2.0 * B * cin * cout * k * k * sp
        t = timeit(lambda: modconv_raw(x, w, None, None, mode, tr), 20)
        print(f"B={B} {cin}->{cout} {h}x{h}->{oh} k={k} mode={mode} T={tr}: {t:8.1f} us {flop / t / 1e6:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
