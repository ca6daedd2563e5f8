import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench, gan2shape_amd
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = Trainer(GAN2Shape, bench.face_config(8), device=dev)
image, latent = bench.synthetic_sample(tr.model, 1234, dev)
r = bench.StepRunner(tr, image, latent)
for _ in range(2):
    for k in (1, 2, 3):
        r.run(k)
torch.cuda.synchronize()
for kind in (1, 2, 3):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        r.run(kind)
        torch.cuda.synchronize()
    seen = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
            for k in e.kernels:
                if "naive_conv" in k.name:
                    p = e
                    chain = []
                    while p is not None and len(chain) < 4:
                        chain.append(p.name); p = p.cpu_parent
                    seen[(k.name[:45], str(e.input_shapes)[:150], " < ".join(chain)[:120], round(k.duration))] += 1
    print("== step", kind)
    for k, c in seen.most_common(12):
        print(c, k)
