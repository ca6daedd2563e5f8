"""Generate golden vectors from the reference's own importable Python (run in the build container).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (/root/reference) is imported read-only; only DATA (inputs, parameters, outputs) is
written.  Nothing here runs on the GPU box (the reference does not travel).  What is importable is
recorded in SURVEY.md §8c: the vendored StyleGAN2 `op/` package (its pure-PyTorch fallback),
`model.py` of the vendored StyleGAN2, GAN2Shape/renderer/utils.py (by file path),
GAN2Shape/renderer/renderer.py geometry methods (with an import-only placeholder for the absent
`neural_renderer` module — no rasterizer behaviour is involved), GAN2Shape/utils.py.

Seed convention: torch.manual_seed(0) immediately before each case; fp32; CPU.
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
SG2 = os.path.join(REF, "GAN2Shape/stylegan2/stylegan2-pytorch")
OUT = os.path.dirname(os.path.abspath(__file__))


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def np_(t):
    return t.detach().cpu().numpy()


def ops_golden():
    sys.path.insert(0, SG2)
    from op import fused_leaky_relu, upfirdn2d  # noqa: native fallback on CPU
    import model as sg2
    out = {}
    # ---- upfirdn2d: the four hot signatures + a down=2 case
    k1 = sg2.make_kernel([1, 3, 3, 1])
    cases = {
        "blur_up": dict(shape=(2, 3, 9, 9), k=k1 * 4, up=1, down=1, pad=(1, 1)),
        "rgb_up": dict(shape=(2, 3, 8, 8), k=k1 * 4, up=2, down=1, pad=(2, 1)),
        "d_blur3": dict(shape=(2, 3, 8, 8), k=k1, up=1, down=1, pad=(2, 2)),
        "d_blur1": dict(shape=(2, 3, 8, 8), k=k1, up=1, down=1, pad=(1, 1)),
        "down2": dict(shape=(2, 3, 8, 8), k=k1, up=1, down=2, pad=(1, 1)),
        "crop": dict(shape=(1, 2, 10, 10), k=k1, up=1, down=1, pad=(-1, 2)),
    }
    for name, c in cases.items():
        torch.manual_seed(0)
        x = torch.randn(*c["shape"], requires_grad=True)
        y = upfirdn2d(x, c["k"], up=c["up"], down=c["down"], pad=c["pad"])
        gy = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, x, gy)
        out[f"upfirdn2d.{name}.x"] = np_(x)
        out[f"upfirdn2d.{name}.k"] = np_(c["k"])
        out[f"upfirdn2d.{name}.args"] = np.array([c["up"], c["down"], c["pad"][0], c["pad"][1]])
        out[f"upfirdn2d.{name}.y"] = np_(y)
        out[f"upfirdn2d.{name}.gy"] = np_(gy)
        out[f"upfirdn2d.{name}.gx"] = np_(gx)
    # ---- fused_leaky_relu 4-D and 2-D
    for name, shape in {"4d": (2, 4, 5, 5), "2d": (3, 8)}.items():
        torch.manual_seed(0)
        x = torch.randn(*shape, requires_grad=True)
        b = torch.randn(shape[1], requires_grad=True)
        y = fused_leaky_relu(x, b)
        gy = torch.randn_like(y)
        gx, gb = torch.autograd.grad(y, (x, b), gy)
        for k_, v in dict(x=x, b=b, y=y, gy=gy, gx=gx, gb=gb).items():
            out[f"fused.{name}.{k_}"] = np_(v)
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)

    # ---- ModulatedConv2d: plain / upsample / 1x1 no-demod / downsample
    out = {}
    cfgs = {
        "plain": dict(cin=8, cout=6, k=3, kw={}),
        "up": dict(cin=8, cout=6, k=3, kw=dict(upsample=True)),
        "rgb": dict(cin=8, cout=3, k=1, kw=dict(demodulate=False)),
        "down": dict(cin=8, cout=6, k=3, kw=dict(downsample=True)),
    }
    for name, c in cfgs.items():
        torch.manual_seed(0)
        m = sg2.ModulatedConv2d(c["cin"], c["cout"], c["k"], 16, **c["kw"])
        with torch.no_grad():
            m.modulation.bias.add_(0.1 * torch.randn_like(m.modulation.bias))
        h = 6 if name == "down" else 5
        x = torch.randn(2, c["cin"], h, h, requires_grad=True)
        s = torch.randn(2, 16, requires_grad=True)
        y = m(x, s)
        gy = torch.randn_like(y)
        gx, gs, gw = torch.autograd.grad(y, (x, s, m.weight), gy)
        out[f"{name}.weight"] = np_(m.weight)
        out[f"{name}.mod_weight"] = np_(m.modulation.weight)
        out[f"{name}.mod_bias"] = np_(m.modulation.bias)
        for k_, v in dict(x=x, s=s, y=y, gy=gy, gx=gx, gs=gs, gw=gw).items():
            out[f"{name}.{k_}"] = np_(v)
        # intermediate: the conv output before the blur (upsample only) — model.py:272-274
        if name == "up":
            style = m.modulation(s)
            out["up.style_mod"] = np_(style)
    np.savez_compressed(os.path.join(OUT, "modconv.npz"), **out)

    # ---- mapping network slice semantics (style_forward skip/depth), Generator.forward tiny
    out = {}
    torch.manual_seed(0)
    g = sg2.Generator(8, 32, 4, channel_multiplier=1)
    # too big to commit (512-ch convs); store only the mapping network + PixelNorm behaviour
    z = torch.randn(3, 32)
    for i, layer in enumerate(g.style):
        if i > 0:
            out[f"style.{i}.weight"] = np_(layer.weight)
            out[f"style.{i}.bias"] = np_(layer.bias)
    out["style.z"] = np_(z)
    out["style.full"] = np_(g.style_forward(z))
    out["style.depth3"] = np_(g.style_forward(z, depth=3))
    out["style.skip3"] = np_(g.style_forward(g.style_forward(z, depth=3), skip=3))
    np.savez_compressed(os.path.join(OUT, "mapping.npz"), **out)


def geometry_golden():
    ru = _load_by_path("ref_renderer_utils", os.path.join(REF, "GAN2Shape/renderer/utils.py"))
    out = {}
    out["face_idx_4x4"] = np_(ru.get_face_idx(2, 4, 4))
    out["face_idx_3x5"] = np_(ru.get_face_idx(1, 3, 5))
    out["grid_norm"] = np_(ru.get_grid(2, 3, 4, normalize=True))
    out["grid_px"] = np_(ru.get_grid(1, 3, 4, normalize=False))
    torch.manual_seed(0)
    v6 = torch.randn(3, 6) * 0.3
    out["view6"] = np_(v6)
    for n in (3, 5, 6):
        r, t = ru.get_transform_matrices(v6[:, :n])
        out[f"rot{n}"] = np_(r)
        out[f"trans{n}"] = np_(t)

    # Renderer geometry: import renderer.py with an import-only placeholder for neural_renderer
    # (gives the pure-torch geometry methods; render_depth stays unavailable).
    sys.modules.setdefault("neural_renderer", types.ModuleType("neural_renderer"))
    pkg = types.ModuleType("ref_renderer_pkg")
    pkg.__path__ = [os.path.join(REF, "GAN2Shape/renderer")]
    sys.modules["ref_renderer_pkg"] = pkg
    sys.modules["ref_renderer_pkg.utils"] = _load_by_path(
        "ref_renderer_pkg.utils", os.path.join(REF, "GAN2Shape/renderer/utils.py"))
    rr = _load_by_path("ref_renderer_pkg.renderer", os.path.join(REF, "GAN2Shape/renderer/renderer.py"))
    S = 16
    R = object.__new__(rr.Renderer)  # __init__ calls .cuda() (renderer.py:33-42)
    R.image_size, R.min_depth, R.max_depth = S, 0.9, 1.1
    R.rot_center_depth, R.fov = 1.0, 10
    fx = (S - 1) / 2 / (math.tan(R.fov / 2 * math.pi / 180))
    c = (S - 1) / 2
    K = torch.FloatTensor([[fx, 0., c], [0., fx, c], [0., 0., 1.]])
    R.K = K.unsqueeze(0)
    R.inv_K = torch.inverse(K).unsqueeze(0)
    torch.manual_seed(0)
    depth = 0.9 + 0.2 * torch.rand(2, S, S)
    view = torch.tensor([[0.1, -0.2, 0.05, 0.01, 0.02, -0.03],
                         [-0.3, 0.4, -0.1, -0.02, 0.01, 0.05]])
    R.set_transform_matrices(view)
    out["r.depth"] = np_(depth)
    out["r.view"] = np_(view)
    out["r.K"] = np_(R.K)
    out["r.inv_K"] = np_(R.inv_K)
    out["r.grid3d"] = np_(R.depth_to_3d_grid(depth))
    out["r.warped3d"] = np_(R.get_warped_3d_grid(depth))
    out["r.invwarped3d"] = np_(R.get_inv_warped_3d_grid(depth))
    out["r.invwarped2d"] = np_(R.get_inv_warped_2d_grid(depth))
    out["r.normal"] = np_(R.get_normal_from_depth(depth))
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), **out)


def misc_golden():
    sys.path.insert(0, REF)
    gu = _load_by_path("ref_g2s_utils", os.path.join(REF, "GAN2Shape/utils.py"))
    out = {}
    torch.manual_seed(0)
    x = torch.randn(2, 3, 8, 8)
    out["resize.x"] = np_(x)
    out["resize.up"] = np_(gu.resize(x, [16, 16]))
    out["resize.down"] = np_(gu.resize(x, [4, 4]))
    out["resize.same"] = np_(gu.resize(x, [8, 8]))
    out["resize.x3"] = np_(x[:, 0])
    out["resize.down3"] = np_(gu.resize(x[:, 0], [4, 4]))
    np.savez_compressed(os.path.join(OUT, "misc.npz"), **out)


def fill_deterministic(module, seed):
    """Overwrite every parameter and buffer with seeded values, visiting the state dict in sorted
    key order — the test does the same on the build's state-dict-compatible module, so the weights
    need not be stored."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for k in sorted(sd.keys()):
            t = sd[k]
            if not t.is_floating_point():
                continue
            if k.endswith("kernel"):
                continue  # FIR taps stay as constructed
            v = torch.randn(t.shape, generator=g)
            if "modulation.bias" in k:
                v = 1 + 0.1 * v
            elif k.endswith("bias") or "noise" in k:
                v = 0.1 * v
            t.copy_(v)


def gan_golden():
    sys.path.insert(0, SG2)
    import model as sg2
    out = {}
    g = sg2.Generator(16, 32, 3, channel_multiplier=1)
    fill_deterministic(g, 123)
    g.eval()
    torch.manual_seed(0)
    w = torch.randn(2, 32, requires_grad=True)
    img, _ = g([w], input_is_w=True, randomize_noise=False)
    gy = torch.randn_like(img)
    (gw,) = torch.autograd.grad(img, w, gy)
    out["g.w"], out["g.img"], out["g.gy"], out["g.gw"] = np_(w), np_(img), np_(gy), np_(gw)
    z = torch.randn(2, 32)
    img2, _ = g([z], input_is_w=False, randomize_noise=False)
    out["g.z"], out["g.img_from_z"] = np_(z), np_(img2)
    out["g.n_keys"] = np.array(len(g.state_dict()))

    d = sg2.Discriminator(16, channel_multiplier=1)
    fill_deterministic(d, 321)
    d.eval()
    x = torch.randn(2, 3, 16, 16, requires_grad=True)
    _, feats = d(x, ftr_num=2)
    gf = [torch.randn_like(f) for f in feats]
    (gx,) = torch.autograd.grad(feats, x, gf)
    out["d.x"], out["d.gx"] = np_(x), np_(gx)
    for i, (f, g_) in enumerate(zip(feats, gf)):
        out[f"d.f{i}"], out[f"d.gf{i}"] = np_(f), np_(g_)
    score, feats_all = d(x)
    out["d.score"] = np_(score)
    np.savez_compressed(os.path.join(OUT, "gan.npz"), **out)


if __name__ == "__main__":
    ops_golden()
    geometry_golden()
    misc_golden()
    gan_golden()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
