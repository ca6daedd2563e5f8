"""Directional-derivative check of whole training steps on the GPU.

For steps 2 and 3 of a fresh model (the set-up of tests/test_gpu_round3.py::
test_every_step_kind_actually_optimises: bench.face_config(n_proj=4), seed 0, prior pre-training, step 1
run 25 times first) this prints

  * the analytic gradient g of the step's trained parameters through the PRODUCT path (one-node
    discriminator-feature loss / LPIPS, one-node demodulated convolution) and through the op-by-op
    autograd path: norms, cosine, relative difference;
  * central differences (L(theta + eps d) - L(theta - eps d)) / (2 eps) along d = g / |g| and along seeded
    random unit directions r_k, against <g, d>, over a range of eps (fp32: the window between rounding
    noise and curvature is read off the table);
  * a line search L(theta - t g) and the loss curve of Adam at lr 1e-3 (the test's) and 1e-4 (the
    reference's) on the FIXED objective (torch.manual_seed(7) before every evaluation).

    python tools/dirderiv.py [--steps 2 3] [--adam-iters 60]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def params_of(t, step):
    optim = getattr(t, f"optim_step{step}")
    return [p for group in optim.param_groups for p in group["params"]]


def flat(ts):
    return torch.cat([t.reshape(-1) for t in ts])


def set_flat(ps, vec):
    off = 0
    with torch.no_grad():
        for p in ps:
            n = p.numel()
            p.copy_(vec[off:off + n].view_as(p))
            off += n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, nargs="+", default=[2, 3])
    ap.add_argument("--adam-iters", type=int, default=60)
    ap.add_argument("--n-proj", type=int, default=4)
    args = ap.parse_args()
    import bench
    from gan2shape_amd import losses, lpips
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import Trainer

    torch.manual_seed(0)
    cfg = bench.face_config(n_proj=args.n_proj)
    cfg["n_epochs_prior"] = 60
    cfg["learning_rate"] = 1e-3
    t = Trainer(GAN2Shape, cfg, device="cuda")
    image, latent = bench.synthetic_sample(t.model, 4321, torch.device("cuda"))
    t.pretrain_on_prior(image, 0)
    m = t.model
    torch.manual_seed(101)
    collected = {1: None}
    for it in range(25):
        t.optim_step1.zero_grad()
        loss, out = m.forward_step1(image, latent, None, n_proj_samples=args.n_proj)
        loss.backward()
        t.optim_step1.step()
    collected[2] = out
    torch.manual_seed(7)
    with torch.no_grad():
        _, collected[3] = m.forward_step2(image, latent, out, n_proj_samples=args.n_proj)

    def evaluate(step, grad):
        torch.manual_seed(7)
        ps = params_of(t, step)
        for p in ps:
            p.grad = None
        with torch.enable_grad() if grad else torch.no_grad():
            loss, _ = getattr(m, f"forward_step{step}")(image, latent, collected[step], n_proj_samples=args.n_proj)
            if grad:
                loss.backward()
                return float(loss.detach()), flat([p.grad if p.grad is not None else torch.zeros_like(p) for p in ps]).double()
        return float(loss)

    for step in args.steps:
        ps = params_of(t, step)
        theta0 = flat([p.detach() for p in ps]).clone()
        print(f"\n===== step {step}: {len(ps)} tensors, {theta0.numel()} parameters, |theta| = {float(theta0.norm()):.3f}")
        grads = {}
        for name, one in (("product (one-node losses)", True), ("op-by-op autograd", False)):
            losses.DiscriminatorLoss.ONE_NODE = one
            lpips.PNetLin.ONE_NODE = one
            try:
                l0, g = evaluate(step, True)
            finally:
                losses.DiscriminatorLoss.ONE_NODE = True
                lpips.PNetLin.ONE_NODE = True
            grads[name] = g
            print(f"  {name:28s} loss {l0:.6f}  |g| {float(g.norm()):.6e}")
        a, b = grads["product (one-node losses)"], grads["op-by-op autograd"]
        print(f"  product vs op-by-op: cosine {float(a @ b / (a.norm() * b.norm())):.8f}, "
              f"|a - b| / |b| = {float((a - b).norm() / b.norm()):.3e}")
        # repeat: run-to-run scatter of the product gradient itself (float atomics)
        _, a2 = evaluate(step, True)
        print(f"  product, second run : |a - a'| / |a| = {float((a - a2).norm() / a.norm()):.3e}")
        l_rep = [evaluate(step, False) for _ in range(4)]
        print(f"  loss under no_grad, 4 runs: {l_rep}  (noise {np.ptp(l_rep):.2e})")

        g = a
        gen = torch.Generator(device="cuda").manual_seed(1000 + step)
        dirs = [("g/|g|", (g / g.norm()))]
        for k in range(4):
            r = torch.randn(g.numel(), generator=gen, device="cuda", dtype=torch.float64)
            dirs.append((f"r{k}", r / r.norm()))
        print("  central differences  (L(+eps d) - L(-eps d)) / 2 eps   vs   <g, d>")
        for name, d in dirs:
            want = float(g @ d)
            row = []
            for eps in (3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1):
                set_flat(ps, (theta0.double() + eps * d).float())
                lp_ = evaluate(step, False)
                set_flat(ps, (theta0.double() - eps * d).float())
                lm_ = evaluate(step, False)
                row.append((eps, (lp_ - lm_) / (2 * eps)))
            set_flat(ps, theta0)
            print(f"    {name:6s} <g,d> = {want:+.5e} | " + "  ".join(f"eps {e:.0e}: {v:+.5e}" for e, v in row))
        print("  line search L(theta - t g):")
        base = evaluate(step, False)
        row = []
        for tt in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1):
            set_flat(ps, (theta0.double() - tt * g).float())
            row.append((tt, evaluate(step, False) - base, -tt * float(g @ g)))
        set_flat(ps, theta0)
        print("    " + "  ".join(f"t {tt:.0e}: dL {dl:+.3e} (first order {fo:+.3e})" for tt, dl, fo in row))

        nets = {2: [m.offset_encoder_net], 3: [m.lighting_net, m.viewpoint_net, m.depth_net, m.albedo_net]}[step]
        for lr, which in ((1e-3, "torch"), (1e-4, "torch"), (1e-3, "g2s"), (1e-4, "g2s")):
            set_flat(ps, theta0)
            if which == "torch":
                optim = torch.optim.Adam(ps, lr=lr, betas=(0.9, 0.999), weight_decay=5e-4)
            else:
                optim = Trainer.default_optimizer(nets, lr=lr)      # the product's: optim.Adam on g2s_adam_step
                assert type(optim).__module__.endswith("gan2shape_amd.optim"), type(optim)
            curve = []
            for it in range(args.adam_iters):
                torch.manual_seed(7)
                optim.zero_grad()
                loss, _ = getattr(m, f"forward_step{step}")(image, latent, collected[step], n_proj_samples=args.n_proj)
                loss.backward()
                optim.step()
                curve.append(float(loss.detach()))
            print(f"  Adam lr {lr:g} ({which} Adam, fixed objective), {args.adam_iters} iterations: "
                  f"first 3 mean {np.mean(curve[:3]):.4f}, last 3 mean {np.mean(curve[-3:]):.4f}, min {min(curve):.4f}, max {max(curve):.4f}")
            print("    " + " ".join(f"{v:.3f}" for v in curve))
        set_flat(ps, theta0)


if __name__ == "__main__":
    main()
