"""Per-launch census of the MFMA convolution kernel (g2s_modconv / g2s_conv_bias_act) in one eager
iteration of each step kind: shape, time (HIP events), TFLOP/s.  python tools/conv_census.py [kind...]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import gan2shape_amd  # noqa
from gan2shape_amd import lib
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
torch.manual_seed(0)
tr = Trainer(GAN2Shape, bench.face_config(8), device=dev)
image, latent = bench.synthetic_sample(tr.model, 1234, dev)
r = bench.StepRunner(tr, image, latent)
for k in (1, 2, 3):
    r.run(k)
L = lib.load()
REC = None


def wrap(name, shape_of):
    f = getattr(L, name)

    def g(*a):
        if REC is None:
            return f(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = f(*a)
        e1.record()
        REC.append((name, shape_of(a), e0, e1))
        return rc
    setattr(L, name, g)


# g2s_modconv(x,w,in_scale,out_scale,y,B,Cin,Cout,H,W,k,mode,transpose,stream)
wrap("g2s_modconv", lambda a: (a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[2] is not None, a[3] is not None))
# g2s_conv_bias_act(x,w,bias,y,B,Cin,Cout,H,W,k,mode,act,alpha,gain,stream)
wrap("g2s_conv_bias_act", lambda a: (a[4], a[5], a[6], a[7], a[8], a[9], a[10], 0, False, False))

# g2s_modconv_ex(x,w,in_scale,out_scale,bias,y,B,Cin,Cout,H,W,k,mode,transpose,act,alpha,gain,y_is_zero,stream)
wrap("g2s_modconv_ex", lambda a: (a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[2] is not None, a[3] is not None))

# g2s_conv3x3_wino(x,U,in_scale,out_scale,bias,y,B,Cr,M,H,W,act,alpha,gain,splitk,ws,ws_floats,stream)
wrap("g2s_conv3x3_wino", lambda a: (a[6], a[7], a[8], a[9], a[10], 3, 0, 0, a[2] is not None, a[3] is not None))

for kind in [int(x) for x in sys.argv[1:]] or [1, 2, 3]:
    r.run(kind)
    REC = []
    r.run(kind)
    torch.cuda.synchronize()
    rec, REC = REC, None
    agg = collections.OrderedDict()
    for name, sh, e0, e1 in rec:
        B, Cin, Cout, H, W, k, mode, tr_, si, so = sh
        if mode == 0:
            sp = H * W
        elif tr_:
            sp = min(H * W, ((H - k) // 2 + 1) ** 2 if mode == 1 else ((H - 1) * 2 + k) ** 2)
        else:
            sp = min(H * W, ((H - 1) * 2 + k) ** 2 if mode == 1 else ((H - k) // 2 + 1) ** 2)
        fl = 2.0 * B * Cout * Cin * k * k * sp
        key = (name.replace("g2s_", ""), sh)
        v = agg.setdefault(key, [0, 0.0, fl])
        v[0] += 1
        v[1] += e0.elapsed_time(e1) * 1e3
    T = sum(v[1] for v in agg.values())
    FL = sum(v[0] * v[2] for v in agg.values())
    print(f"== step {kind}: {len(rec)} conv launches, {T / 1e3:.3f} ms, {FL / 1e9:.1f} GFLOP, {FL / T / 1e6:.1f} TFLOP/s")
    for (name, sh), v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        B, Cin, Cout, H, W, k, mode, tr_, si, so = sh
        print(f"  {v[1]:8.1f} us n={v[0]:2d} avg={v[1] / v[0]:7.1f} us {v[2] * v[0] / v[1] / 1e6:6.1f} TF/s  {name:14s} B={B} {Cin}->{Cout} {H}x{W} k={k} mode={mode} T={tr_} scales={int(si)}{int(so)}")
