"""Where do the small aten launches of a step come from?  Profiles one eager iteration of a step kind
with Python stacks and lists every aten op that launches a kernel, grouped by the innermost frame inside
this repository (forward ops) or by the autograd node (backward ops).
python tools/census_glue.py <kind> [top]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
import gan2shape_amd  # noqa
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer

kind = int(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
torch.manual_seed(0)
tr = Trainer(GAN2Shape, bench.face_config(8), device=dev)
image, latent = bench.synthetic_sample(tr.model, 1234, dev)
r = bench.StepRunner(tr, image, latent)
for k in (1, 2, 3):
    r.run(k)
for _ in range(2):
    r.run(kind)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    r.run(kind)
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels or not e.name.startswith("aten::"):
        continue
    if e.cpu_children and any(c.kernels for c in e.cpu_children):
        continue
    where = "?"
    for fr in (e.stack or []):
        if "gan-2d-to-3d_amd" in fr or "gan2shape_amd" in fr or "/bench.py" in fr:
            where = fr.replace(ROOT, "").strip()
            break
    else:
        # backward: name the autograd node from the parent chain
        p = e.cpu_parent
        while p is not None and not (p.name.endswith("Backward") or "Backward" in p.name or p.name.startswith("autograd::")):
            p = p.cpu_parent
        where = f"<backward of {p.name}>" if p is not None else "<no repo frame>"
    shapes = str([s for s in (e.input_shapes or []) if s])[:60]
    key = (e.name, where[:110], shapes)
    agg[key][0] += len(e.kernels)
    agg[key][1] += sum(k.duration for k in e.kernels)
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
print(f"step {kind}: {sum(v[0] for v in agg.values())} aten kernels, {sum(v[1] for v in agg.values()) / 1e3:.2f} ms")
for (name, where, shapes), (n, us) in rows[:top]:
    print(f"{n:3d} x {us:7.1f} us  {name:28s} {shapes:60s} {where}")
