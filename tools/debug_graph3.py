import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import bench
import gan2shape_amd
from gan2shape_amd.model import GAN2Shape
case = sys.argv[1]
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = GAN2Shape(bench.face_config(2), device=dev)
image, latent = bench.synthetic_sample(m, 1234, dev)

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

def bw(net, x):
    def f():
        net.zero_grad(set_to_none=True)
        net(x).sum().backward()
    return f

if case == "depth": cap(bw(m.depth_net, image))
elif case == "depth_fwd":
    with torch.no_grad(): cap(lambda: m.depth_net(image))
elif case == "view": cap(bw(m.viewpoint_net, image))
elif case == "offset": cap(bw(m.offset_encoder_net, image))
elif case == "lpips":
    x = image.clone().requires_grad_(True)
    def f():
        x.grad = None
        m.perceptual_loss(x, image * 0.5).mean().backward()
    cap(f)
elif case == "geom":
    d = (torch.rand(1, 128, 128, device=dev) * 0.2 + 0.9).requires_grad_(True)
    v = torch.zeros(1, 6, device=dev)
    def f():
        d.grad = None
        m.renderer.set_transform_matrices(v)
        n = m.renderer.get_normal_from_depth(d)
        r = m.renderer.warp_canon_depth(d)
        g = m.renderer.get_inv_warped_2d_grid(r)
        out = F.grid_sample(image, g, mode='bilinear', align_corners=True)
        (out.sum() + n.sum()).backward()
    cap(f)
elif case == "step1_fwd":
    with torch.no_grad(): cap(lambda: m.forward_step1(image, latent, None))
elif case == "step1":
    def f():
        m.zero_grad(set_to_none=True)
        l, c = m.forward_step1(image, latent, None)
        l.backward()
    cap(f)
elif case == "smooth":
    d = (torch.rand(1, 128, 128, device=dev)).requires_grad_(True)
    def f():
        d.grad = None
        m.smooth_loss(d).backward()
    cap(f)
elif case.startswith("opt"):
    from gan2shape_amd.trainer import Trainer
    opt = Trainer.default_optimizer([m.albedo_net], capturable=True)
    if case == "opt_pre":   # an eager step first, as bench does
        opt.zero_grad()
        l, c = m.forward_step1(image, latent, None); l.backward(); opt.step()
        torch.cuda.synchronize()
    if case == "opt_pre_nostep":
        opt.zero_grad()
        l, c = m.forward_step1(image, latent, None); l.backward()
        torch.cuda.synchronize()
    if case == "opt_pre_side":
        s0 = torch.cuda.Stream(); s0.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s0):
            opt.zero_grad()
            l, c = m.forward_step1(image, latent, None); l.backward(); opt.step()
        torch.cuda.current_stream().wait_stream(s0)
        torch.cuda.synchronize()
    if case == "opt_pre_fwdonly":
        with torch.no_grad():
            l, c = m.forward_step1(image, latent, None)
        torch.cuda.synchronize()
    def f():
        opt.zero_grad(set_to_none=True)
        l, c = m.forward_step1(image, latent, None)
        l.backward()
        opt.step()
    if case == "opt_nozero":
        def f():
            l, c = m.forward_step1(image, latent, None)
            l.backward()
            opt.step()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                opt.zero_grad(set_to_none=True); f()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            f()
        torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize(); print(case, "OK")
    else:
        cap(f)
