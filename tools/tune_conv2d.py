"""Sweep tile configuration x split-K of g2s_conv2d over the convolution calls of the trained nets
(depth / albedo B = 1, viewpoint / lighting B = 9, offset encoder B = 8; forward and data-gradient),
timed as HIP-graph replays (the eager loop is host-bound at these sizes).  Writes the table
csrc/conv2d_tuned.inc is made of.   python tools/tune_conv2d.py gpurun_out/conv2d_tuned.inc"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan2shape_amd  # noqa
from gan2shape_amd import lib
from gan2shape_amd.op.conv import _conv2d_raw

torch.cuda.set_stream(torch.cuda.Stream())
L = lib.load()


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * n) * 1e3


# layers: B, Cin, Cout, H, k, stride, pad, transposed, fused activation in forward, needs dgrad
ED = [(3, 32, 128, 4, 2, 1, 0, 0, 0), (32, 64, 64, 4, 2, 1, 0, 0, 1), (64, 128, 32, 4, 2, 1, 0, 0, 1),
      (128, 256, 16, 4, 2, 1, 0, 1, 1), (256, 256, 4, 4, 1, 0, 0, 1, 1), (256, 256, 1, 4, 1, 0, 1, 1, 1),
      (256, 256, 4, 3, 1, 1, 0, 1, 1), (256, 128, 4, 4, 2, 1, 1, 0, 1), (128, 128, 8, 3, 1, 1, 0, 0, 1),
      (128, 64, 8, 4, 2, 1, 1, 0, 1), (64, 64, 16, 3, 1, 1, 0, 0, 1), (64, 32, 16, 4, 2, 1, 1, 0, 1),
      (32, 32, 32, 3, 1, 1, 0, 0, 1), (32, 32, 128, 3, 1, 1, 0, 0, 1), (32, 32, 128, 5, 1, 2, 0, 0, 1),
      (32, 1, 128, 5, 1, 2, 0, 0, 1), (32, 3, 128, 5, 1, 2, 0, 0, 1)]
ENC = [(3, 32, 128, 4, 2, 1, 0, 1, 0), (32, 64, 64, 4, 2, 1, 0, 1, 1), (64, 128, 32, 4, 2, 1, 0, 1, 1),
       (128, 256, 16, 4, 2, 1, 0, 1, 1), (256, 512, 8, 4, 2, 1, 0, 1, 1), (512, 512, 4, 4, 1, 0, 0, 1, 1),
       (512, 6, 1, 1, 1, 0, 0, 0, 1), (512, 4, 1, 1, 1, 0, 0, 0, 1)]
OFF = [(3, 32, 128, 4, 2, 1, 0, 0, 0)]
for cin, cout, h in [(32, 64, 64), (64, 128, 32), (128, 256, 16), (256, 512, 8)]:
    OFF += [(cin, cout, h, 3, 2, 1, 0, 0, 1), (cout, cout, h // 2, 3, 1, 1, 0, 0, 1), (cin, cout, h // 2, 1, 1, 0, 0, 0, 1)]
OFF += [(512, 1024, 4, 4, 1, 0, 0, 0, 1), (1024, 512, 1, 1, 1, 0, 0, 0, 1)]
# (B, groups) + layer.  groups = 2: depth + albedo / viewpoint + lighting as one pass (networks.forward_pair);
# their first layer is one plain convolution with both nets' filters (M doubled).
PAIR_FIRST = [(3, 64, 128, 4, 2, 1, 0, 0, 0)]
JOBS = [(1, 1) + l for l in ED] + [(9, 1) + l for l in ENC] + [(8, 1) + l for l in OFF] + \
       [(1, 2) + l for l in ED[1:-2]] + [(b, 2) + l for b in (1, 9) for l in ENC[1:-2]] + \
       [(b, 1) + l for b in (1, 9) for l in PAIR_FIRST]
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64]


def main():
    emit, seen = [], set()
    for B, G, cin, cout, h, k, s, p, tr, act, dgrad in JOBS:
        w = torch.randn((G * cin, cout, k, k) if tr else (G * cout, cin, k, k), device="cuda")
        bias = torch.randn(cout, device="cuda")
        x = torch.randn(B, G * cin, h, h, device="cuda")
        y = _conv2d_raw(x, w, None, cin, cout, k, s, p, bool(tr), not tr, None, False, 0.0, groups=G)
        gy = torch.randn_like(y)
        has_bias = B == 8  # the offset encoder's convolutions carry a bias (networks.py:178-241)
        fused = int(bool(act) or has_bias)
        calls = [("fwd", (B, cin, cout, h, k, s, p, tr, int(not tr), fused, G),
                  lambda: _conv2d_raw(x, w, bias if has_bias else None, cin, cout, k, s, p, bool(tr), not tr, None,
                                      bool(act), 0.0, groups=G))]
        if dgrad:
            calls.append(("dgrad", (B, cout, cin, y.shape[2], k, s, p, int(not tr), int(bool(tr)), 0, G),
                          lambda: _conv2d_raw(gy, w, None, cout, cin, k, s, p, not tr, bool(tr), (h, h), False, 0.0,
                                              groups=G)))
        for name, key, f in calls:
            if key in seen or y.shape[2] != y.shape[3]:
                continue
            seen.add(key)
            L.g2s_modconv_tune(-2, -1)
            t0 = min(timeit(f), timeit(f))
            res = []
            for tile in (2, 3, 4, 1, 0):
                for sk in SPLITS:
                    L.g2s_modconv_tune(tile, sk)
                    res.append((timeit(f, 10), tile, sk))
            res.sort()
            L.g2s_modconv_tune(res[0][1], res[0][2])
            tb = min(timeit(f), timeit(f))
            L.g2s_modconv_tune(-1, -1)
            print(f"{name:5s} B={key[0]} g{key[10]} {key[1]}->{key[2]} {key[3]}^2 k{key[4]} s{key[5]} p{key[6]} adj{key[7]} mm{key[8]} f{key[9]}: "
                  f"heuristic {t0:6.1f} us | best {tb:6.1f} us ({t0 / tb:.2f}x) t{res[0][1]} sk{res[0][2]}", flush=True)
            if t0 / tb >= 1.05 and t0 - tb >= 0.5:
                emit.append("{%d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d},  // %.1f -> %.1f us" % (key + (res[0][1], res[0][2], t0, tb)))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("// Generated by tools/tune_conv2d.py on MI355X (HIP-graph replay timing): g2s_conv2d call signatures of\n"
                    "// the trained nets whose best (tile, split-K) beats the built-in heuristic by >= 5 %.\n"
                    "// {B, Cr, M, H, k, stride, pad, adjoint, w_m_major, fused_epilogue, groups, tile, splitk},  // heuristic -> tuned\n")
            f.write("\n".join(emit) + "\n")


if __name__ == "__main__":
    main()
