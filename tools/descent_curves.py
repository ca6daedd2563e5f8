"""The body of tests/test_gpu_round3.py::test_every_step_kind_actually_optimises as of round 3, printing the
whole loss curve of step 2 (fixed objective: torch.manual_seed(7) before every iteration; the trainer's own
optimiser at lr 1e-3, 25 iterations) for fresh models, with the discriminator-feature loss as one autograd
node and op by op.  Runs from any checkout of this repository (ROOT = the tree the file sits in, or
--root): used to re-run the red run of gpurun_out/r3_gpu_tests_11_1.log on the tree of that commit.

    python tools/descent_curves.py [--root DIR] [--repeats 3] [--iters 25]
"""
import argparse
import os
import sys

import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--root", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap.add_argument("--repeats", type=int, default=3)
ap.add_argument("--iters", type=int, default=25)
ap.add_argument("--lr", type=float, default=1e-3)
args = ap.parse_args()
sys.path.insert(0, os.path.abspath(args.root))
import bench                                           # noqa: E402
from gan2shape_amd import losses                       # noqa: E402
from gan2shape_amd.model import GAN2Shape              # noqa: E402
from gan2shape_amd.trainer import Trainer              # noqa: E402

print("tree:", os.path.abspath(args.root), " bench:", bench.__file__)
for one in (True, False):
    for rep in range(args.repeats):
        losses.DiscriminatorLoss.ONE_NODE = one
        torch.manual_seed(0)
        cfg = bench.face_config(n_proj=4)
        cfg["n_epochs_prior"] = 60
        cfg["learning_rate"] = args.lr
        t = Trainer(GAN2Shape, cfg, device="cuda")
        image, latent = bench.synthetic_sample(t.model, 4321, torch.device("cuda"))
        t.pretrain_on_prior(image, 0)
        m = t.model
        collected = None
        curves = {}
        for step, n_it in ((1, 25), (2, args.iters)):
            optim = getattr(t, f"optim_step{step}")
            forward = getattr(m, f"forward_step{step}")
            torch.manual_seed(100 + step)
            curve, out = [], None
            for it in range(n_it):
                if step == 2:
                    torch.manual_seed(7)
                optim.zero_grad()
                loss, out = forward(image, latent, collected, n_proj_samples=4)
                loss.backward()
                optim.step()
                curve.append(float(loss.detach()))
            collected = out
            curves[step] = curve
        c = curves[2]
        print(f"ONE_NODE={one} run {rep}: step 2 head {np.mean(c[:3]):.4f} tail {np.mean(c[-3:]):.4f} | " + " ".join(f"{v:.3f}" for v in c))
        del t, m
