"""Two eager iterations of every step kind at image / GAN sizes other than the benchmark's (64 / 64, 64 / 128,
128 / 256): finite losses — a quick check that the fused kernels take every supported shape.
python tools/smoke_sizes.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
import gan2shape_amd
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer
for size, gan in ((64, 64), (64, 128), (128, 256)):
    torch.manual_seed(0)
    cfg = bench._face_config(4)
    cfg.update(image_size=size, gan_size=gan)
    t = Trainer(GAN2Shape, cfg, device="cuda")
    image, latent = bench.synthetic_sample(t.model, 5, torch.device("cuda"))
    r = bench.StepRunner(t, image, latent)
    out = []
    for it in range(2):
        for k in (1, 2, 3):
            out.append(float(r.run(k)))
    torch.cuda.synchronize()
    assert all(v == v and abs(v) < 1e6 for v in out), out
    print("image", size, "gan", gan, "losses", [round(v, 4) for v in out], flush=True)
print("ok")
