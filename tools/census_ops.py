"""aten-level census of one eager iteration of a step kind: op name x input shapes -> calls, kernels,
device time (forward and backward).  python tools/census_ops.py <kind> [top]"""
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
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    r.run(kind)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0, 0.0])
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if e.cpu_children and any(c.kernels for c in e.cpu_children):
        continue  # count leaf ops only
    shapes = str([s for s in (e.input_shapes or []) if s])[:90]
    key = (e.name, shapes)
    agg[key][0] += 1
    agg[key][1] += len(e.kernels)
    agg[key][2] += sum(k.duration for k in e.kernels)
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
tot = sum(v[2] for v in agg.values())
print(f"step {kind}: {sum(v[1] for v in agg.values())} kernels, {tot / 1e3:.2f} ms device time")
for (name, shapes), (calls, kernels, us) in rows[:top]:
    print(f"{us:9.1f} us {calls:4d} calls {kernels:4d} kernels  {name[:40]:40s} {shapes}")
