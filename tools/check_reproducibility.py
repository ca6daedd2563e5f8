"""Which gradients are bit-identical from run to run?  Runs each step kind twice from the same state in
g2s_set_deterministic(1) (and once more in the default mode) and reports, per trained net, how many
parameter-gradient tensors differ and by how much.  python tools/check_reproducibility.py"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import gan2shape_amd  # noqa
from gan2shape_amd import lib
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer

torch.manual_seed(0)
t = Trainer(GAN2Shape, bench.face_config(n_proj=8), device="cuda")
m = t.model
image, latent = bench.synthetic_sample(m, 1234, torch.device("cuda"))


def grads_of(step, collected):
    for p in m.parameters():
        p.grad = None
    torch.manual_seed(5)
    loss, out = getattr(m, f"forward_step{step}")(image, latent, collected, n_proj_samples=8)
    loss.backward()
    g = {f"{n}.{k}": p.grad.clone() for n in m.NETS for k, p in getattr(m, n + "_net").named_parameters() if p.grad is not None}
    return float(loss.detach()), g, out


for mode in (1, 0):
    lib.set_deterministic(bool(mode))
    collected = None
    print(f"---- deterministic = {mode}")
    for step in (1, 2, 3):
        l1, g1, out = grads_of(step, collected)
        l2, g2, _ = grads_of(step, collected)
        per_net = {}
        for k in g1:
            net = k.split(".")[0]
            same = torch.equal(g1[k], g2[k])
            d = float((g1[k] - g2[k]).norm() / (g1[k].norm() + 1e-30))
            a = per_net.setdefault(net, [0, 0, 0.0])
            a[0] += 1
            a[1] += int(not same)
            a[2] = max(a[2], d)
        print(f"step {step}: loss {l1!r} vs {l2!r} ({'equal' if l1 == l2 else 'DIFFERENT'}); " +
              "; ".join(f"{n}: {v[1]}/{v[0]} tensors differ (max rel {v[2]:.1e})" for n, v in per_net.items()))
        collected = out
lib.set_deterministic(False)

if "--torch-ops" in sys.argv:
    import warnings
    torch.use_deterministic_algorithms(True, warn_only=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        collected = None
        for step in (1, 2, 3):
            _, _, collected = grads_of(step, collected)
    seen = sorted({str(x.message).split(".")[0] for x in w if "deterministic" in str(x.message)})
    print("torch ops without a deterministic implementation on the path:")
    for s in seen:
        print("  ", s)
