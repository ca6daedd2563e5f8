"""The 20-step 7:7:6 cycle of bench.py as eager training iterations (after one warm-up of each kind),
for the rocprofv3 --pmc passes behind bench.py's roofline.traffic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, gan2shape_amd
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = Trainer(GAN2Shape, bench.face_config(8), device=dev)
image, latent = bench.synthetic_sample(tr.model, 1234, dev)
r = bench.StepRunner(tr, image, latent)
for k in (1, 2, 3):
    r.run(k)
torch.cuda.synchronize()
torch.ones(3, dtype=torch.int32, device=dev).bitwise_not()  # marker: the counted cycle starts here
torch.cuda.synchronize()
for i, k in enumerate(bench.PATTERN):
    r.run(k)
    torch.cuda.synchronize()
    print("done", i, k, flush=True)
