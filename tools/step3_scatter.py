"""Where does the run-to-run scatter of step 3's pose / depth gradients come from?  

    python tools/step3_scatter.py [runs]

Runs forward_step3 + backward of the steps.npz fixture model `runs` times in the default launch
mode and in g2s_set_deterministic(1), and prints per run: the loss, the gradient norms, and against
run 0 of the same mode: flipped mask pixels of both rasterizer calls, the relative L2 distance of
d loss / d view, d loss / d depth, d loss / d warped depth."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gan2shape_amd  # noqa: E402,F401
from gan2shape_amd import lib  # noqa: E402
from model_cases import capture_step_tensors  # noqa: E402
from test_gpu_golden import _gnorm, _zero_grads, build_step_model, dev  # noqa: E402


def one_run(m, g):
    image, latent = dev(g["image"]), dev(g["latent"])
    c2 = (dev(g["s2.projected"]), dev(g["s2.mask"]))
    _zero_grads(m)
    with capture_step_tensors(m) as cap:
        loss, _ = m.forward_step3(image, latent, c2)
        loss.backward()
    out = {"loss": loss.item()}
    for name in ("viewpoint", "depth", "lighting", "albedo"):
        out["gnorm." + name] = _gnorm(m, name)
        out["grad." + name] = torch.cat([p.grad.reshape(-1) for p in getattr(m, name + "_net").parameters()]).double().cpu().numpy()
    for i in range(2):
        out[f"view{i}"], out[f"gview{i}"] = cap.view[i].detach().cpu().numpy(), cap.view[i].grad.cpu().numpy()
        out[f"recon_depth{i}"], out[f"grecon{i}"] = cap.recon_depth[i].detach().cpu().numpy(), cap.recon_depth[i].grad.cpu().numpy()
        out[f"gverts{i}"] = cap.verts[i].grad.cpu().numpy()
    out["depth"], out["gdepth"] = cap.depth[0].detach().cpu().numpy(), cap.depth[0].grad.cpu().numpy()
    return out


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    g = dict(np.load(os.path.join(ROOT, "tests/golden/steps.npz")))
    m = build_step_model(g)
    ref = {k[3:]: v for k, v in g.items() if k.startswith("s3.")}
    for mode in (0, 1):
        lib.set_deterministic(bool(mode))
        res = [one_run(m, g) for _ in range(runs)]
        print(f"---- deterministic = {mode}")
        for r, o in enumerate(res):
            base = res[0]
            flips = [int(((o[f"recon_depth{i}"] < 1.2) != (base[f"recon_depth{i}"] < 1.2)).sum()) for i in range(2)]
            flips_ref = [int(((o[f"recon_depth{i}"] < 1.2) != (ref[f"recon_depth{i}"] < 1.2)).sum()) for i in range(2)]
            print(f"run {r}: loss {o['loss']:.7f} (ref {float(g['s3.loss']):.7f})  |g| view {o['gnorm.viewpoint']:.4f} "
                  f"(ref {float(g['s3.gnorm.viewpoint']):.4f}) depth {o['gnorm.depth']:.4f} (ref {float(g['s3.gnorm.depth']):.4f})")
            print(f"    vs run 0: mask flips {flips}  bitwise-equal depth {np.array_equal(o['depth'], base['depth'])} "
                  f"view {np.array_equal(o['view1'], base['view1'])} recon {np.array_equal(o['recon_depth1'], base['recon_depth1'])}  "
                  f"rel gview1 {rel(o['gview1'], base['gview1']):.2e} gdepth {rel(o['gdepth'], base['gdepth']):.2e} "
                  f"grecon1 {rel(o['grecon1'], base['grecon1']):.2e} net-grad view {rel(o['grad.viewpoint'], base['grad.viewpoint']):.2e} "
                  f"depth {rel(o['grad.depth'], base['grad.depth']):.2e}")
            print(f"    vs reference run: mask flips {flips_ref}  rel view1 {rel(o['view1'], ref['view1']):.2e} depth {rel(o['depth'], ref['depth']):.2e} "
                  f"gview0 {rel(o['gview0'], ref['gview0']):.2e} gview1 {rel(o['gview1'], ref['gview1']):.2e} gdepth {rel(o['gdepth'], ref['gdepth']):.2e} "
                  f"grecon0 {rel(o['grecon0'], ref['grecon0']):.2e} grecon1 {rel(o['grecon1'], ref['grecon1']):.2e}")
            print("    gview1 build", np.round(o["gview1"][0], 4), " ref", np.round(ref["gview1"][0], 4))
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"step3_scatter_mode{mode}.npz"),
                            **{f"r{r}.{k}": v for r, o in enumerate(res) for k, v in o.items() if not k.startswith("grad.")})
    lib.set_deterministic(False)


if __name__ == "__main__":
    main()
