"""One eager training iteration of each kind (after one warm-up of each) for rocprofv3 --pmc passes."""
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
for rep in range(2):
    for k in (1, 2, 3):
        r.run(k)
        torch.cuda.synchronize()
        print("done", rep, k, flush=True)
