"""Which Python lines launch the remaining aten kernels of a step kind (one eager step under torch.profiler with
stacks): what is left to fuse.    python tools/aten_sources.py KIND"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
import gan2shape_amd  # noqa
from gan2shape_amd import lib
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer

kind = int(sys.argv[1]) if len(sys.argv) > 1 else 3
device = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device))
lib.load()
torch.manual_seed(0)
trainer = Trainer(GAN2Shape, bench.face_config(8, "face128_n8"), device=device, capturable=False)
image, latent = bench.synthetic_sample(trainer.model, 1234, device)
runner = bench.StepRunner(trainer, image, latent)
for k in (1, 2, 3, 1, 2, 3):
    runner.run(k)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    runner.run(kind)
    torch.cuda.synchronize()
rows = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    n = len(getattr(ev, "kernels", []) or [])
    if not n:
        continue
    stack = ev.stack or []
    where = next((s for s in stack if "gan-2d-to-3d_amd" in s or "gan2shape_amd" in s or "bench.py" in s), stack[0] if stack else "(autograd engine)")
    rows[(ev.name, where.strip()[-100:])] += n
for (name, where), n in rows.most_common(70):
    print(f"{n:3d}  {name:30s} {where}")
