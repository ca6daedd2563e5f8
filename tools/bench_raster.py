"""Rasterizer micro-benchmark at the BASELINE size (S = 128, ssaa 2) through the C ABI: forward
with saved maps and backward, B = 1 and 8, on GAN2Shape-like scenes (tests/raster_cases.scene):
"hard" = +-60 deg poses with a depth step and per-vertex noise (folded mesh, heavy overdraw),
"easy" = +-5 deg.  Times are HIP-event averages over 50 back-to-back calls."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import gan2shape_amd  # noqa
from gan2shape_amd import lib
from raster_cases import scene

S = 128
N, F = S * S, 2 * (S - 1) ** 2


def timed(fn, n=50):
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
    L = lib.load()
    st = lib.stream()
    for B in (1, 2, 4, 8):
        for name, seed, rot in (("hard", 1, 60.0), ("hard", 2, 60.0), ("easy", 1, 5.0)):
            geo, verts, _ = scene(S, B=B, seed=seed, rot=rot)
            K = (C.c_float * 9)(*np.asarray(geo.K[0], np.float32).reshape(9).tolist())
            v = torch.tensor(verts, device="cuda").contiguous()
            wsb = L.g2s_raster_workspace_bytes(B, N, F, S)
            ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            d = torch.empty(B, S, S, device="cuda")
            fi = torch.empty(B, 2 * S, 2 * S, dtype=torch.int32, device="cuda")
            ba = torch.empty(B, 2 * S, 2 * S, 3, device="cuda")
            gv = torch.empty(B, N, 3, device="cuda")

            def fwd():
                lib.check(L.g2s_raster_depth_fwd(lib.ptr(v), None, B, N, F, S, K, float(S), 2, 1, 0.1, 100.0,
                                                 lib.ptr(d), lib.ptr(fi), lib.ptr(ba), lib.ptr(ws), wsb, st))

            t_f = timed(fwd)
            g = (torch.randn_like(d) * (d < 50)).contiguous()

            def bwd():
                lib.check(L.g2s_raster_depth_bwd(lib.ptr(v), None, lib.ptr(g), lib.ptr(fi), lib.ptr(ba), B, N, F, S,
                                                 K, float(S), 2, lib.ptr(gv), st))

            t_b = timed(bwd)
            cov = float((fi >= 0).float().mean())
            tests = B * (2 * S) ** 2 * 2 * F
            print(f"B={B} {name} seed {seed}: fwd {t_f:7.1f} us  bwd {t_b:7.1f} us | coverage {cov:.2f} | "
                  f"brute-force equivalent {tests / t_f / 1e6:.1f} T tests/s", flush=True)


if __name__ == "__main__":
    main()
