import sys, torch
sys.path.insert(0, "/root/repo")
import gan2shape_amd
from gan2shape_amd import networks
torch.manual_seed(0)
for name in ("DepthNet", "ViewpointNet"):
    net = getattr(networks, name)(128).cuda()
    x = torch.randn(1 if name == "DepthNet" else 9, 3, 128, 128, device="cuda")
    outs = []
    for r in range(6):
        y = net(x)
        gy = torch.ones_like(y) * 0.37 + torch.arange(y.numel(), device="cuda").reshape(y.shape) % 7 * 0.1
        g = torch.autograd.grad(y, list(net.parameters()), gy)
        outs.append((y.detach().clone(), [t.clone() for t in g]))
    y0, g0 = outs[0]
    for r in range(1, 6):
        y, g = outs[r]
        dy = float((y - y0).norm() / y0.norm())
        dg = max(float((a - b).norm() / (b.norm() + 1e-12)) for a, b in zip(g, g0))
        tot = (sum(float((a - b).norm()) ** 2 for a, b in zip(g, g0)) / sum(float(b.norm()) ** 2 for b in g0)) ** 0.5
        print(name, "run", r, "rel diff y %.2e  max per-tensor grad %.2e  all grads %.2e" % (dy, dg, tot))
