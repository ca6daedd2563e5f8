"""Kernel census of one step-3 iteration: forward regions by monkeypatched record_function;
backward kernels attributed to the autograd node names."""
import os, sys, collections, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
import bench, gan2shape_amd
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = Trainer(GAN2Shape, bench.face_config(8), device=dev)
m = tr.model
image, latent = bench.synthetic_sample(m, 1234, dev)
r = bench.StepRunner(tr, image, latent)
for k in (1, 2, 3):
    r.run(k)

def wrap(obj, name, label):
    f = getattr(obj, name)
    @functools.wraps(f)
    def g(*a, **k):
        with record_function("R:" + label):
            return f(*a, **k)
    setattr(obj, name, g)

for n in ["get_normal_from_depth", "warp_canon_depth", "get_inv_warped_2d_grid", "set_transform_matrices"]:
    wrap(m.renderer, n, n)
for n in ["get_clamped_depth", "get_view_transformation", "get_lighting_directions", "get_shading",
          "photometric_loss", "smooth_loss"]:
    wrap(m, n, n)
for n in ["perceptual_loss", "depth_net", "albedo_net", "viewpoint_net", "lighting_net"]:
    wrap(getattr(m, n), "forward", n)
import torch.nn.functional as F
_gs = F.grid_sample
def gs(*a, **k):
    with record_function("R:grid_sample"):
        return _gs(*a, **k)
F.grid_sample = gs
import gan2shape_amd.model as mm
mm.F.grid_sample = gs

for _ in range(2):
    r.run(3)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    r.run(3)
    torch.cuda.synchronize()
ev = prof.events()
regions = [e for e in ev if e.name.startswith("R:")]
cnt = collections.Counter(); tim = collections.Counter()
bw = collections.Counter(); bwt = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    t = e.time_range.start
    reg = None
    for rg in regions:
        if rg.time_range.start <= t <= rg.time_range.end:
            reg = rg.name; break
    if reg is None:
        reg = "fwd-glue/bwd/optim"
        # find enclosing autograd node
    for k in e.kernels:
        cnt[reg] += 1; tim[reg] += k.duration
print("total kernels", sum(cnt.values()))
for rg, c in cnt.most_common():
    print(f"{rg:28s} kernels={c:5d} device_us={tim[rg]:9.1f}")
# backward by autograd node
nodes = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU and (e.name.startswith("autograd::engine::evaluate_function") )]
bc = collections.Counter(); bt = collections.Counter()
for nd in nodes:
    name = nd.name.split(": ")[-1]
    for e in ev:
        pass
import re
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and e.cpu_parent is not None:
        p = e
        top = None
        while p is not None:
            if p.name.startswith("autograd::engine::evaluate_function"):
                top = p.name.split(": ")[-1]
            p = p.cpu_parent
        if top:
            for k in e.kernels:
                bc[top] += 1; bt[top] += k.duration
print("---- backward by autograd node (top 25)")
for n, c in bc.most_common(25):
    print(f"{n:40s} kernels={c:5d} device_us={bt[n]:9.1f}")
