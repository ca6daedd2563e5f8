"""Kernel census of one eager iteration of a step kind: forward kernels by model region
(record_function around the model's sub-calls), backward kernels by autograd node.
python tools/census_step.py <kind>"""
import collections
import functools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile, record_function

import bench
import gan2shape_amd  # noqa
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer

kind = int(sys.argv[1])
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
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


for n in ["get_normal_from_depth", "warp_canon_depth", "get_inv_warped_2d_grid", "set_transform_matrices", "render_given_view"]:
    if hasattr(m.renderer, n):
        wrap(m.renderer, n, "renderer." + n)
for n in ["get_clamped_depth", "get_view_transformation", "get_lighting_directions", "get_shading", "photometric_loss",
          "smooth_loss", "sample_pseudo_imgs", "discriminator_loss", "_latent_centers"]:
    if hasattr(m, n):
        wrap(m, n, n)
for n in ["perceptual_loss", "depth_net", "albedo_net", "viewpoint_net", "lighting_net", "offset_encoder_net", "discriminator"]:
    if hasattr(m, n):
        wrap(getattr(m, n), "forward", n)
wrap(m.generator, "style_forward", "G.style_forward")
wrap(m.generator, "invert", "G.invert(total)")
for i, c in enumerate(m.generator.convs):
    wrap(c, "forward", "G.styledconv")
wrap(m.generator.conv1, "forward", "G.styledconv")
for t in list(m.generator.to_rgbs) + [m.generator.to_rgb1]:
    wrap(t, "forward", "G.to_rgb")
wrap(tr, "optim_step" if hasattr(tr, "optim_step") else "fit", "optimizer")

for _ in range(2):
    r.run(kind)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    r.run(kind)
    torch.cuda.synchronize()
ev = prof.events()
regions = sorted([e for e in ev if e.name.startswith("R:")], key=lambda e: e.time_range.end - e.time_range.start)
cnt, tim = collections.Counter(), collections.Counter()
bc, bt = collections.Counter(), collections.Counter()
kn = collections.defaultdict(collections.Counter)
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    p, top = e, None
    while p is not None:
        if p.name.startswith("autograd::engine::evaluate_function"):
            top = p.name.split(": ")[-1]
        p = p.cpu_parent
    if top:
        for k in e.kernels:
            bc[top] += 1
            bt[top] += k.duration
            kn["B:" + top][k.name[:70]] += 1
        continue
    t = e.time_range.start
    reg = "(glue / optimizer)"
    for rg in regions:  # innermost (shortest) region containing the op
        if rg.time_range.start <= t <= rg.time_range.end:
            reg = rg.name[2:]
            break
    for k in e.kernels:
        cnt[reg] += 1
        tim[reg] += k.duration
        kn[reg][k.name[:70]] += 1
print(f"step {kind}: forward+glue kernels {sum(cnt.values())} ({sum(tim.values()) / 1e3:.2f} ms), backward kernels {sum(bc.values())} ({sum(bt.values()) / 1e3:.2f} ms)")
for rg, c in sorted(cnt.items(), key=lambda kv: -tim[kv[0]]):
    print(f"F {rg:34s} kernels={c:5d} device_us={tim[rg]:9.1f}")
print("---- backward by autograd node")
for n, c in sorted(bc.items(), key=lambda kv: -bt[kv[0]])[:40]:
    print(f"B {n:44s} kernels={c:5d} device_us={bt[n]:9.1f}")
if len(sys.argv) > 2:
    for reg in sys.argv[2:]:
        print("----", reg)
        for n, c in kn[reg].most_common(30):
            print(f"   {c:4d}  {n}")
