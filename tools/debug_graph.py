import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gan2shape_amd
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer
from gan2shape_amd.graphs import GraphedSteps
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
torch.manual_seed(0)
tr = Trainer(GAN2Shape, bench.face_config(int(sys.argv[2]) if len(sys.argv) > 2 else 8), device=dev, capturable=True)
image, latent = bench.synthetic_sample(tr.model, 1234, dev)
r = bench.StepRunner(tr, image, latent)
for k in (1, 2, 3):
    r.run(k)
torch.cuda.synchronize()
print("eager ok", flush=True)
g = GraphedSteps(tr, image, latent)
g.collected = dict(r.collected)
kinds = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "123")]
for k in kinds:
    print("capturing", k, flush=True)
    g.capture(k)
    torch.cuda.synchronize()
    print("captured", k, flush=True)
    for _ in range(3):
        g.run(k)
    torch.cuda.synchronize()
    print("replayed", k, float(g.loss[k]), flush=True)
