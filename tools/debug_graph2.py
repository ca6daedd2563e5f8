import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan2shape_amd
from gan2shape_amd.modconv import modconv_raw
from gan2shape_amd.op import fused_leaky_relu, upfirdn2d
from gan2shape_amd.plugins import neural_renderer as nr
import numpy as np
case = sys.argv[1]
dev = torch.device("cuda:0")

def cap(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print(case, "OK", flush=True)

if case == "fused":
    x = torch.randn(2, 8, 16, 16, device=dev); b = torch.randn(8, device=dev)
    cap(lambda: fused_leaky_relu(x, b))
elif case == "upfirdn":
    x = torch.randn(2, 8, 16, 16, device=dev); k = torch.ones(4, 4, device=dev)
    cap(lambda: upfirdn2d(x, k, pad=(1, 1)))
elif case == "modconv":
    x = torch.randn(2, 64, 16, 16, device=dev); w = torch.randn(64, 64, 3, 3, device=dev)
    cap(lambda: modconv_raw(x, w, None, None, 0, 0))
elif case == "modconv_splitk":
    x = torch.randn(2, 512, 4, 4, device=dev); w = torch.randn(512, 512, 3, 3, device=dev)
    cap(lambda: modconv_raw(x, w, None, None, 0, 0))
elif case == "raster":
    S = 32
    r = nr.Renderer(camera_mode='projection', K=torch.tensor([[[100., 0, 15.5], [0, 100., 15.5], [0, 0, 1]]], device=dev), image_size=S, orig_size=S)
    v = torch.randn(1, S * S, 3, device=dev) * 0.05 + torch.tensor([0, 0, 1.0], device=dev)
    r.render_depth(v, None)
    cap(lambda: r.render_depth(v, None))
elif case == "raster_bwd":
    S = 32
    r = nr.Renderer(camera_mode='projection', K=torch.tensor([[[100., 0, 15.5], [0, 100., 15.5], [0, 0, 1]]], device=dev), image_size=S, orig_size=S)
    v = (torch.randn(1, S * S, 3, device=dev) * 0.05 + torch.tensor([0, 0, 1.0], device=dev)).requires_grad_(True)
    def f():
        v.grad = None
        d = r.render_depth(v, None)
        d.clamp(max=1.2).sum().backward()
    cap(f)
elif case == "conv":
    m = torch.nn.Conv2d(8, 8, 3, padding=1).to(dev); x = torch.randn(2, 8, 32, 32, device=dev)
    cap(lambda: m(x))
elif case == "conv_bwd":
    m = torch.nn.Conv2d(8, 8, 3, padding=1).to(dev); x = torch.randn(2, 8, 32, 32, device=dev)
    def f():
        m.zero_grad(set_to_none=True)
        m(x).sum().backward()
    cap(f)
elif case == "adam":
    m = torch.nn.Conv2d(8, 8, 3, padding=1).to(dev); x = torch.randn(2, 8, 32, 32, device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=5e-4, capturable=True)
    def f():
        opt.zero_grad(set_to_none=True)
        m(x).sum().backward()
        opt.step()
    cap(f)
elif case == "mvn":
    from torch.distributions.multivariate_normal import MultivariateNormal
    d = MultivariateNormal(torch.zeros(6, device=dev), torch.eye(6, device=dev))
    cap(lambda: d.sample())
