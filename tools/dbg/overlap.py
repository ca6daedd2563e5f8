"""Would dgrad and wgrad of one small-net layer overlap if they ran concurrently? (two eager streams)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gan2shape_amd  # noqa
from gan2shape_amd.op.conv import _conv2d_raw, _wgrad
dev = torch.device("cuda:0")
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
torch.cuda.set_stream(s0)
for (B, G, cin, cout, h, k, s, p) in [(1, 2, 128, 128, 8, 3, 1, 1), (1, 2, 64, 128, 32, 4, 2, 1), (1, 2, 32, 32, 128, 3, 1, 1),
                                      (9, 2, 128, 256, 16, 4, 2, 1), (8, 1, 128, 128, 16, 3, 1, 1)]:
    x = torch.randn(B, G * cin, h, h, device=dev)
    w = torch.randn(G * cout, cin, k, k, device=dev)
    y = _conv2d_raw(x, w, None, cin, cout, k, s, p, False, True, None, False, 0.0, groups=G)
    gy = torch.randn_like(y)
    d = lambda: _conv2d_raw(gy, w, None, cout, cin, k, s, p, True, False, (h, h), False, 0.0, groups=G)
    g = lambda: _wgrad(gy, x, k, s, p, None, G)
    def seq():
        d(); g()
    def par():
        e = torch.cuda.Event(); e.record(s0); s1.wait_event(e)
        with torch.cuda.stream(s1):
            g()
            e2 = torch.cuda.Event(); e2.record(s1)
        d()
        s0.wait_event(e2)
    res = {}
    for name, f in (("dgrad", d), ("wgrad", g), ("seq", seq), ("par", par)):
        for _ in range(5): f()
        torch.cuda.synchronize()
        # host-bound in eager: queue behind a long kernel so launches are back to back
        big = torch.randn(8192, 8192, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): big @ big
        e0.record()
        for _ in range(20): f()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    print((B, G, cin, cout, h, k, s), {k_: round(v, 1) for k_, v in res.items()}, flush=True)
