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

    # texture-cube helpers feeding render_rgb (renderer/utils.py:83-109)
    torch.manual_seed(2)
    im = torch.randn(2, 3, 4, 5)
    out["tex.im"] = np_(im)
    out["tex.size1"] = np_(ru.get_textures_from_im(im, tx_size=1))
    out["tex.size2"] = np_(ru.get_textures_from_im(im, tx_size=2))
    vc = torch.randn(1, 2, 3, 3)
    out["tex.vcolors"], out["tex.cube"] = np_(vc), np_(ru.vcolor_to_texture_cube(vc))

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
    # gradients of <output, cotangent> w.r.t. depth and view through the reference's own autograd
    # graph (pins the analytic backward kernels of csrc/geometry.hip directly)
    for name, fn in (("warped3d", R.get_warped_3d_grid), ("invwarped2d", R.get_inv_warped_2d_grid),
                     ("normal", R.get_normal_from_depth)):
        d_ = depth.clone().requires_grad_(True)
        v_ = view.clone().requires_grad_(True)
        R.set_transform_matrices(v_)
        y = fn(d_)
        torch.manual_seed(1)
        cot = torch.randn_like(y)
        grads = torch.autograd.grad((y * cot).sum(), (d_, v_), allow_unused=True)
        out[f"r.{name}.cot"] = np_(cot)
        out[f"r.{name}.gdepth"] = np_(grads[0])
        if grads[1] is not None:
            out[f"r.{name}.gview"] = np_(grads[1])
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


def _import_reference_model():
    """GAN2Shape/model.py, losses.py, priors.py with import-only placeholders for what is absent
    offline: `neural_renderer` (external CUDA package) and the `GAN2Shape.stylegan2`
    package __init__ (it pulls lpips -> IPython / skimage / torchvision); the placeholder exposes
    the vendored Generator / Discriminator and no PerceptualLoss behaviour.  The pure-torch methods
    are then called on bare objects; `.cuda()` (hard-coded in model.py:342,351,452-456) is the
    identity for the duration of the call so that the reference's own lines run on CPU tensors."""
    sys.path.insert(0, REF)
    sys.path.insert(0, SG2)
    sys.modules.setdefault("neural_renderer", types.ModuleType("neural_renderer"))
    import model as sg2
    stub = types.ModuleType("GAN2Shape.stylegan2")
    stub.Generator, stub.Discriminator, stub.PerceptualLoss = sg2.Generator, sg2.Discriminator, None
    sys.modules["GAN2Shape.stylegan2"] = stub
    import GAN2Shape.losses as ref_losses
    import GAN2Shape.model as ref_model
    import GAN2Shape.priors as ref_priors
    return ref_model, ref_losses, ref_priors


class _cuda_is_identity:
    def __enter__(self):
        self.orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda t, *a, **k: t

    def __exit__(self, *exc):
        torch.Tensor.cuda = self.orig


def model_golden():
    """Model-level math of the training step (GAN2Shape/model.py:85-93,330-360,448-470), the three
    loss classes (losses.py:6-79) and the priors (priors.py:26-107) — outputs of the reference's own
    code on seeded inputs; the stand-in discriminator / depth net / parsing mask come from
    tests/model_cases.py and are rebuilt by the tests."""
    sys.path.insert(0, os.path.dirname(OUT))
    import tempfile
    import model_cases as mc
    ref_model, ref_losses, ref_priors = _import_reference_model()
    out = {}
    M = object.__new__(ref_model.GAN2Shape)
    torch.nn.Module.__init__(M)
    M.max_depth, M.min_depth = 1.1, 0.9
    M.border_depth = 0.7 * M.max_depth + 0.3 * M.min_depth
    M.xyz_rotation_range, M.xy_translation_range, M.z_translation_range = 60, 0.1, 0.1
    M.depth_net = mc.fake_depth_net
    S = 16
    with _cuda_is_identity():
        torch.manual_seed(0)
        raw = torch.randn(2, S, S)
        out["m.depth_raw"] = np_(raw)
        out["m.rescale"] = np_(M.rescale_depth(torch.tanh(raw)))
        out["m.clamped"] = np_(M.get_clamped_depth(raw, S, S))
        out["m.clamped_noborder"] = np_(M.get_clamped_depth(raw, S, S, clamp_border=False))
        view = torch.randn(3, 6)
        out["m.view"], out["m.view_trans"] = np_(view), np_(M.get_view_transformation(view))
        light = torch.randn(3, 4) * 0.7
        la, lb, ld = M.get_lighting_directions(light)
        out["m.light"], out["m.light_a"], out["m.light_b"], out["m.light_d"] = np_(light), np_(la), np_(lb), np_(ld)
        normal = torch.randn(3, S, S, 3)
        normal = normal / normal.norm(dim=3, keepdim=True)
        albedo = torch.tanh(torch.randn(3, 3, S, S))
        diffuse, texture = M.get_shading(normal, la, lb, ld, albedo)
        out["m.normal"], out["m.albedo"] = np_(normal), np_(albedo)
        out["m.diffuse"], out["m.texture"] = np_(diffuse), np_(texture)
        # gradients of sum(texture * gt) + sum(diffuse * gd) w.r.t. normal / light / albedo
        n_, l_, a_ = (t.clone().requires_grad_(True) for t in (normal, light, albedo))
        d2, t2 = M.get_shading(n_, *M.get_lighting_directions(l_), a_)
        gt, gd = torch.randn_like(t2), torch.randn_like(d2)
        gn, gl, ga = torch.autograd.grad((t2 * gt).sum() + (d2 * gd).sum(), (n_, l_, a_))
        for k_, v in dict(gt=gt, gd=gd, gn=gn, gl=gl, ga=ga).items():
            out[f"m.shade.{k_}"] = np_(v)
        # prior pre-training forward (model.py:88-93) incl. its 4-D broadcast
        x = torch.randn(2, 3, S, S)
        prior = 0.91 + 0.11 * torch.rand(1, S, S)
        loss, depth = M.depth_net_forward(x, prior)
        out["m.dnf.x"], out["m.dnf.prior"] = np_(x), np_(prior)
        out["m.dnf.loss"], out["m.dnf.depth"] = np_(loss), np_(depth)

        # ---- ViewLightSampler (model.py:448-470): draw order + view_scale, CPU generator
        with tempfile.TemporaryDirectory() as tmp:
            vm, lm = torch.randn(6) * 0.1, torch.randn(4) * 0.1
            a = torch.randn(6, 6) * 0.2
            vc = a @ a.T + 0.01 * torch.eye(6)
            b = torch.randn(4, 4) * 0.2
            lc = b @ b.T + 0.01 * torch.eye(4)
            torch.save({"mean": vm, "cov": vc}, os.path.join(tmp, "v.pth"))
            torch.save({"mean": lm, "cov": lc}, os.path.join(tmp, "l.pth"))
            sampler = ref_model.ViewLightSampler(os.path.join(tmp, "v.pth"), os.path.join(tmp, "l.pth"), 0.5)
        torch.manual_seed(0)
        out["vls.views"] = np_(sampler.sample(5, "view"))
        out["vls.lights"] = np_(sampler.sample(3, "light"))
        out["vls.views2"] = np_(sampler.sample(2))
        for k_, v in dict(vm=vm, vc=vc, lm=lm, lc=lc).items():
            out[f"vls.{k_}"] = np_(v)

    # ---- losses (losses.py:6-79)
    torch.manual_seed(0)
    a, b = torch.randn(2, 3, 32, 32), torch.randn(2, 3, 32, 32)
    mask = (torch.rand(2, 1, 32, 32) > 0.3).float()
    sigma = torch.rand(2, 1, 32, 32) + 0.1
    out["l.a"], out["l.b"], out["l.mask"], out["l.sigma"] = np_(a), np_(b), np_(mask), np_(sigma)
    P, Sm, DL = ref_losses.PhotometricLoss(), ref_losses.SmoothLoss(), ref_losses.DiscriminatorLoss()
    out["l.photo"] = np_(P(a, b))
    out["l.photo_mask"] = np_(P(a, b, mask=mask))
    out["l.photo_sigma"] = np_(P(a, b, mask=mask, conf_sigma=sigma))
    out["l.smooth3"] = np_(Sm(a[:, 0]))
    out["l.smooth4"] = np_(Sm(a))
    out["l.smooth_pyr"] = np_(Sm([a, b[:, :, ::2, ::2]]))
    ar = a.clone().requires_grad_(True)
    (g,) = torch.autograd.grad(Sm(ar[:, 0]), ar)
    out["l.smooth3_grad"] = np_(g)
    D = mc.FakeD()
    fake = a.clone().requires_grad_(True)
    for name, m_ in (("", None), ("_mask", mask)):
        val = DL(D, fake, b, mask=m_)
        (g,) = torch.autograd.grad(val, fake)
        out[f"l.dloss{name}"], out[f"l.dloss{name}_grad"] = np_(val), np_(g)
    out["l.dloss_ftr2"] = np_(ref_losses.DiscriminatorLoss(ftr_num=2)(D, a, b, mask=mask))

    # ---- priors (priors.py:26-107) from the synthetic parsing mask
    for size in (64, 128):
        fm = mc.FakeMaskingModel(size)
        img = torch.zeros(1, 3, size, size)
        for name in mc.PRIOR_NAMES:
            pg = object.__new__(ref_priors.PriorGenerator)   # __init__ loads parsing checkpoints
            pg.image_size, pg.category, pg.prior = size, "face", name
            pg.noise_threshold, pg.near, pg.far = 0.7, 0.91, 1.02
            pg.base_prior = torch.Tensor(1, size, size).fill_(1.02)
            pg.masking_model = fm
            if size == 128 and name not in ("ellipsoid", "smoothed_box"):
                continue
            out[f"p.{name}.{size}"] = np_(pg(img, device="cpu")).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "model.npz"), **out)


def _oracle_neural_renderer(precision=np.float32):
    """A `neural_renderer` module for the reference's Renderer to bind (renderer.py:6,47-54,120): its
    render_depth is THIS repo's C oracle (oracle/raster_body.inc — SURVEY.md Appendix A restated;
    PARITY UNPINNED, the real package is un-vendored), forward and backward.  Everything around it
    — the three step functions, shading, sampling, losses, the trained nets, G and D — is the
    reference's own code, so the step-level fixtures pin that code conditional on the rasterizer."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import capi
    maps = []
    # `precision[0]`: np.float32 = the rasterizer's own arithmetic; np.float64 = the same algorithm in double
    # (oracle/raster_body.inc is compiled for both) — used ONLY to measure how much of a fixture's gradient
    # hangs on fp32 rounding inside the rasterizer (steps_golden, the smooth step-3 case)
    precision = [precision]

    class _RenderDepth(torch.autograd.Function):
        @staticmethod
        def forward(ctx, verts, faces, S, K):
            dt = precision[0]
            v = verts.detach().numpy().astype(dt)
            f = faces[0].numpy().astype(np.int32)
            ref = capi.render_depth(v, f, S, K.astype(dt), dtype=dt)
            ctx.save_for_backward(verts)
            ctx.aux = (f, S, K, ref["face_idx"], ref["bary"], dt)
            maps.append((ref["face_idx"], ref["bary"]))
            return torch.from_numpy(ref["depth"].astype(np.float32))

        @staticmethod
        def backward(ctx, g):
            (verts,) = ctx.saved_tensors
            f, S, K, fidx, bary, dt = ctx.aux
            gv = capi.render_depth_bwd(verts.detach().numpy().astype(dt), f,
                                       g.contiguous().numpy().astype(dt), fidx, bary, S, K.astype(dt), dtype=dt)
            return torch.from_numpy(gv.astype(np.float32)), None, None, None

    class Renderer:
        def __init__(self, K=None, image_size=256, **kwargs):
            self.K, self.image_size = K, image_size

        def render_depth(self, vertices, faces):
            return _RenderDepth.apply(vertices, faces, self.image_size, np_(self.K)[0].astype(np.float32))

    mod = types.ModuleType("neural_renderer")
    mod.Renderer = Renderer
    mod.maps = maps          # (face_idx, bary) of every render_depth call, in call order
    mod.precision = precision
    return mod


def fold_report(face_idx, bary, S, edge_tol=1e-5):
    """How a rasterization of the regular S x S grid mesh (renderer/utils.py:77-81: quad q = row * (S-1) + col
    carries faces q and q + (S-1)^2; fill_back copies follow at + F) is conditioned.  Two supersamples that are
    8-neighbours see a DEPTH DISCONTINUITY when one is background and the other is not, or when their winning
    quads are more than 2 rows / columns apart (a fold: the surface occludes itself); a supersample is an EDGE
    sample when its smallest clamped barycentric weight is below edge_tol — which face wins there is decided by
    fp32 rounding.  Returns (discontinuous pairs, edge samples, edge samples AT a discontinuity, largest quad jump
    between covered neighbours)."""
    nq = (S - 1) * (S - 1)
    F = 2 * nq
    covered = face_idx >= 0
    quad = np.where(covered, (face_idx % F) % nq, 0)
    row, col = quad // (S - 1), quad % (S - 1)
    edge = covered & (bary.min(-1) < edge_tol)
    disc = np.zeros_like(covered)
    n_pairs, max_jump = 0, 0
    H, W = face_idx.shape[-2:]
    for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
        a = (slice(None), slice(0, H - dy), slice(max(0, -dx), W - max(0, dx)))
        b = (slice(None), slice(dy, H), slice(max(0, dx), W - max(0, -dx)))
        both = covered[a] & covered[b]
        jump = np.maximum(np.abs(row[a] - row[b]), np.abs(col[a] - col[b]))
        max_jump = max(max_jump, int(jump[both].max()) if both.any() else 0)
        d = (covered[a] != covered[b]) | (both & (jump > 2))
        n_pairs += int(d.sum())
        disc[a] |= d
        disc[b] |= d
    return n_pairs, int(edge.sum()), int((edge & disc).sum()), max_jump


class _torch12_grid_sample:
    """The reference targets torch 1.2, whose grid_sample is what torch >= 1.3 calls
    align_corners=True; its call sites (model.py:151,270, renderer.py:261,263)
    pass no flag."""
    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.orig = F, F.grid_sample
        orig = self.orig
        F.grid_sample = lambda *a, **k: orig(*a, **{"align_corners": True, **k})

    def __exit__(self, *exc):
        self.F.grid_sample = self.orig


def steps_golden():
    """The three training steps of the REFERENCE's GAN2Shape (model.py:95-328) run on the CPU: face
    config at 128x128, n_proj = 2, seeded weights (regenerated by the test: tests/model_cases.py),
    a stand-in perceptual loss, the oracle rasterizer behind the neural_renderer boundary.
    Stored: inputs, the random draws of step 2, every `collected` hand-off, the losses and
    per-net gradient norms."""
    import tempfile
    sys.path.insert(0, os.path.dirname(OUT))
    import model_cases as mc
    sys.modules["neural_renderer"] = _oracle_neural_renderer()
    for name in [m for m in sys.modules if m.startswith("GAN2Shape.renderer")]:
        del sys.modules[name]       # re-import against the behavioural stand-in
    ref_model, ref_losses, _ = _import_reference_model()
    import importlib
    importlib.reload(sys.modules["GAN2Shape.renderer.renderer"]) if "GAN2Shape.renderer.renderer" in sys.modules else None
    importlib.reload(ref_model)
    sys.path.insert(0, SG2)
    import model as sg2
    from GAN2Shape import networks as ref_nets
    cfg = mc.STEP_CFG
    S, n = cfg["image_size"], cfg["n_proj"]
    out = {}
    with _cuda_is_identity(), _torch12_grid_sample():
        M = object.__new__(ref_model.GAN2Shape)
        torch.nn.Module.__init__(M)
        M.z_dim, M.debug, M.image_size, M.collected = cfg["z_dim"], False, S, None
        M.generator = sg2.Generator(cfg["gan_size"], cfg["z_dim"], 8, channel_multiplier=1).eval()
        M.discriminator = sg2.Discriminator(cfg["gan_size"], channel_multiplier=1).eval()
        mc.prepare_generator(M.generator, cfg["seeds"]["G"], fill_deterministic)
        fill_deterministic(M.discriminator, cfg["seeds"]["D"])
        for name, cls in (("lighting", "LightingNet"), ("viewpoint", "ViewpointNet"), ("depth", "DepthNet"),
                          ("albedo", "AlbedoNet"), ("offset_encoder", "OffsetEncoder")):
            net = getattr(ref_nets, cls)(S)
            mc.fill_scaled(net, cfg["seeds"][name])
            setattr(M, f"{name}_net", net)
        M.max_depth, M.min_depth = 1.1, 0.9
        M.border_depth = 0.7 * M.max_depth + 0.3 * M.min_depth
        M.lam_perc, M.lam_smooth, M.lam_regular = 1, 0.01, 0.01
        M.xyz_rotation_range, M.xy_translation_range, M.z_translation_range = 60, 0.1, 0.1
        M.use_mask, M.relative_encoding = True, False
        M.rand_light = [-1, 1, -0.2, 0.8, -0.1, 0.6, -0.6]
        M.truncation, M.mean_latent = 1, None
        from GAN2Shape.renderer import Renderer as RefRenderer
        M.renderer = RefRenderer({"rot_center_depth": 1.0, "fov": 10, "tex_cube_size": 2}, S, M.min_depth, M.max_depth)
        g = torch.Generator().manual_seed(7)
        vm, lm = 0.05 * torch.randn(6, generator=g), 0.05 * torch.randn(4, generator=g)
        vc = torch.diag(torch.tensor([.05, .15, .03, .05, .05, .05]) ** 2)
        lc = 0.01 * torch.eye(4)
        with tempfile.TemporaryDirectory() as tmp:
            torch.save({"mean": vm, "cov": vc}, os.path.join(tmp, "v.pth"))
            torch.save({"mean": lm, "cov": lc}, os.path.join(tmp, "l.pth"))
            M.view_light_sampler = ref_model.ViewLightSampler(os.path.join(tmp, "v.pth"), os.path.join(tmp, "l.pth"), 1)
        for k_, v in dict(vm=vm, vc=vc, lm=lm, lc=lc).items():
            out[f"vls.{k_}"] = np_(v)
        M.smooth_loss, M.photometric_loss = ref_losses.SmoothLoss(), ref_losses.PhotometricLoss()
        M.discriminator_loss = ref_losses.DiscriminatorLoss()
        M.perceptual_loss = mc.fake_perceptual

        image = torch.tanh(torch.nn.functional.interpolate(torch.randn(1, 3, 32, 32, generator=g), scale_factor=4,
                                                           mode="bilinear"))
        with torch.no_grad():
            latent = M.generator.style_forward(torch.randn(1, cfg["z_dim"], generator=g))
        out["image"], out["latent"] = np_(image), np_(latent)
        nets = ("lighting", "viewpoint", "depth", "albedo", "offset_encoder")

        def grad_norms(tag):
            for name in nets:
                ps = [p.grad for p in getattr(M, f"{name}_net").parameters() if p.grad is not None]
                out[f"{tag}.gnorm.{name}"] = np.array(float(sum((q.double() ** 2).sum() for q in ps)) ** 0.5 if ps else 0.0)
                if ps:   # directional evidence: per-tensor norms + <grad, r_k> on fixed probe directions (model_cases)
                    out[f"{tag}.gtnorm.{name}"], out[f"{tag}.gproj.{name}"] = mc.grad_evidence(getattr(M, f"{name}_net"))
            for p in M.parameters():
                p.grad = None

        # ---- step 1 (model.py:95-173)
        loss1, c1 = M.forward_step1(image, latent, None)
        loss1.backward()
        grad_norms("s1")
        out["s1.loss"] = np_(loss1)
        for k_, v in zip(("normal", "light_a", "light_b", "albedo", "depth"), c1[:5]):
            out[f"s1.{k_}"] = np_(v)
        recon_im, recon_depth = M.forward_step1(image, None, None, eval=True)
        out["s1.recon_im"], out["s1.recon_depth"] = np_(recon_im), np_(recon_depth)

        # ---- step 2 (model.py:175-223, 291-328); the draws are replayed after the same seed
        c1d = tuple(t.detach() if torch.is_tensor(t) else t for t in c1)
        torch.manual_seed(11)
        dxy = torch.FloatTensor(n, 2)
        dxy[:, 0].uniform_(M.rand_light[0], M.rand_light[1])
        dxy[:, 1].uniform_(M.rand_light[2], M.rand_light[3])
        rnd = torch.FloatTensor(n, 1, 1, 1).uniform_(M.rand_light[4], M.rand_light[5])
        views = M.view_light_sampler.sample(n, "view")
        out["s2.draw.dxy"], out["s2.draw.rand"], out["s2.draw.views"] = np_(dxy), np_(rnd), np_(views)
        torch.manual_seed(11)
        with torch.no_grad():
            pseudo_im, pmask = M.sample_pseudo_imgs(n, *c1d[:5], c1d[5])
        out["s2.pseudo_im"], out["s2.pseudo_mask"] = np_(pseudo_im), np_(pmask)
        torch.manual_seed(11)
        loss2, c2 = M.forward_step2(image, latent, c1d, n_proj_samples=n)
        loss2.backward()
        grad_norms("s2")
        out["s2.loss"] = np_(loss2)
        out["s2.loss_l1"], out["s2.loss_rec"], out["s2.loss_latent_norm"] = np_(M.loss_l1), np_(M.loss_rec), np_(M.loss_latent_norm)
        out["s2.projected"], out["s2.mask"] = np_(c2[0]), np_(c2[1])

        # ---- step 3 (model.py:225-280)
        # the tensors where the loss meets the geometry chain, with d loss / d tensor: the view vectors
        # ([0]: the image's own, inner step-1 pass; [1]: the n projected samples'), the canonical depth,
        # the warped depth maps of both rasterizer calls (B = 1 and B = n)
        with mc.capture_step_tensors(M) as cap:
            loss3, _ = M.forward_step3(image, latent, c2)
            loss3.backward()
        grad_norms("s3")
        out["s3.loss"] = np_(loss3)
        assert len(cap.view) == 2 and len(cap.depth) == 1 and len(cap.recon_depth) == 2 and len(cap.verts) == 2
        for i in range(2):
            out[f"s3.view{i}"], out[f"s3.gview{i}"] = np_(cap.view[i]), np_(cap.view[i].grad)
            out[f"s3.recon_depth{i}"], out[f"s3.grecon{i}"] = np_(cap.recon_depth[i]), np_(cap.recon_depth[i].grad)
            out[f"s3.gverts{i}"] = np_(cap.verts[i].grad)     # d loss / d mesh vertices: the rasterizer's backward
        out["s3.depth"], out["s3.gdepth"] = np_(cap.depth[0]), np_(cap.depth[0].grad)
        nr = sys.modules["neural_renderer"]
        rough = [fold_report(f, b_, S) for f, b_ in nr.maps[-2:]]
        out["s3.fold_report"] = np.array(rough)      # per rasterizer call: discontinuous pairs, edge samples, both, max quad jump
        print("step 3, rough surface  (discontinuous pairs, edge samples, edge samples at a discontinuity, max quad jump):", rough)

        # ---- step 3 again on a SMOOTH surface.  The fixture above is ill-conditioned by
        # construction: the seeded depth net's output swings over the whole tanh range from pixel to pixel, the
        # warped mesh folds over itself, and a supersample within fp32 rounding of an edge at such a fold decides
        # 10 % of the gradient norm.  Here the depth net's last convolution is scaled down (model_cases.
        # smooth_depth_net: relief of a few 1e-3 around the mean depth, slopes far below what the +-few-degree
        # views need to fold) — everything else, the hand-off included, is unchanged.  ASSERTED below: no depth
        # discontinuity inside the mesh, and no edge sample at the silhouette either, so no winner of the
        # z-test hangs on rounding; MEASURED: the same step with the rasterizer in float64.
        mc.smooth_depth_net(M.depth_net)
        # second inherent discontinuity of the step, and how the fixture is conditioned against it: the gradient of
        # F.grid_sample with respect to the sampling POSITION jumps at every texel row / column (bilinear cells); a
        # position within fp32 rounding of one takes either slope (measured on the GPU build: ONE pixel of 49 152
        # off by 4 % of the largest gradient moved d loss / d view by 1e-2).  Positions are spread evenly, so ~4 tol of
        # all samples lie within tol of a border whatever the scene.  A first pass finds those of THIS run
        # (model_cases.grid_kinks, tol 2e-3 px); the recorded pass — and the test, with the masks stored here — zero
        # the gradient reaching the grid at exactly those pixels.  Nothing is selected by looking at an error.
        with mc.capture_step_tensors(M) as probe, torch.no_grad():
            M.forward_step3(image, latent, c2)
        kinks = [mc.grid_kinks(gr) for gr in probe.grid]
        assert len(kinks) == 2
        for i, km in enumerate(kinks):
            out[f"s3s.kink{i}"] = np.argwhere(km).astype(np.int16)
        print("step 3, smooth surface: grid_sample kink pixels masked per call:", [int(km.sum()) for km in kinks],
              "of", [km.size for km in kinks], "; (1, 102, 112) of call 1 among them:", bool(kinks[1][1, 102, 112]))
        n0 = len(nr.maps)
        with mc.capture_step_tensors(M, kink_masks=kinks) as cap:
            loss3s, _ = M.forward_step3(image, latent, c2)
            loss3s.backward()
        ev32 = {name: mc.grad_evidence(getattr(M, f"{name}_net")) for name in ("viewpoint", "depth", "lighting", "albedo")}
        grad_norms("s3s")
        out["s3s.loss"] = np_(loss3s)
        smooth = [fold_report(f, b_, S) for f, b_ in nr.maps[n0:]]
        print("step 3, smooth surface (discontinuous pairs, edge samples, edge samples at a discontinuity, max quad jump):", smooth)
        out["s3s.fold_report"] = np.array(smooth)
        for i, (f, b_) in enumerate(nr.maps[n0:]):
            covered = f >= 0
            # inside the mesh: neighbouring supersamples never see quads more than 2 apart -> the only depth
            # discontinuity is the silhouette against the background, and no supersample sits on it to 1e-5
            assert smooth[i][3] <= 2, smooth[i]
            assert smooth[i][2] == 0, smooth[i]
        for i in range(2):
            out[f"s3s.view{i}"], out[f"s3s.gview{i}"] = np_(cap.view[i]), np_(cap.view[i].grad)
            out[f"s3s.recon_depth{i}"], out[f"s3s.grecon{i}"] = np_(cap.recon_depth[i]), np_(cap.recon_depth[i].grad)
            out[f"s3s.gverts{i}"] = np_(cap.verts[i].grad)
        out["s3s.depth"], out["s3s.gdepth"] = np_(cap.depth[0]), np_(cap.depth[0].grad)
        g32 = dict(view0=cap.view[0].grad.clone(), view1=cap.view[1].grad.clone(), depth=cap.depth[0].grad.clone(),
                   verts0=cap.verts[0].grad.clone(), verts1=cap.verts[1].grad.clone())
        # the same step with the rasterizer (forward AND backward) in float64: how much of these gradients
        # hangs on fp32 rounding inside the rasterizer — the floor of any tolerance against them
        nr.precision[0] = np.float64
        with mc.capture_step_tensors(M, kink_masks=kinks) as cap64:
            loss64, _ = M.forward_step3(image, latent, c2)
            loss64.backward()
        nr.precision[0] = np.float32
        g64 = dict(view0=cap64.view[0].grad, view1=cap64.view[1].grad, depth=cap64.depth[0].grad,
                   verts0=cap64.verts[0].grad, verts1=cap64.verts[1].grad)
        sens = {k_: float((g32[k_].double() - g64[k_].double()).norm() / g64[k_].double().norm()) for k_ in g32}
        for name in ("viewpoint", "depth"):
            t64, p64 = mc.grad_evidence(getattr(M, f"{name}_net"))
            sens[f"gproj.{name}"] = float(np.abs(ev32[name][1] - p64).max() / np.abs(p64).max())
            sens[f"gtnorm.{name}"] = float(np.abs(ev32[name][0] - t64).max() / t64.max())
        sens["loss"] = abs(float(loss3s) - float(loss64)) / abs(float(loss64))
        print("step 3, smooth surface: fp32 vs float64 rasterizer, relative differences:", sens)
        out["s3s.raster64_sensitivity"] = np.array([sens[k_] for k_ in sorted(sens)])
        # measured: view / depth gradients and the nets' projections 4e-6 .. 5e-5, the mesh-vertex gradient itself 2e-4 /
        # 7e-4 (an edge sample that changes to the ADJACENT face moves its share between neighbouring vertices; the sum
        # downstream does not notice).  The tests hold the GPU to 1e-3 on the former and 5e-3 on the latter.
        assert max(v for k_, v in sens.items() if not k_.startswith("verts")) <= 2e-4, sens
        assert max(sens["verts0"], sens["verts1"]) <= 2e-3, sens
        for p in M.parameters():
            p.grad = None
    np.savez_compressed(os.path.join(OUT, "steps.npz"), **{k: (v.astype(np.float32) if v.dtype == np.float64 and v.ndim and ".gproj." not in k
                                                                and ".gtnorm." not in k and "sensitivity" not in k else v)
                                                            for k, v in out.items()})


def trainer_golden():
    """The reference's Trainer.fit (GAN2Shape/trainer.py:13-171) driving a toy model
    (tests/model_cases.py:ToyStepModel) on the CPU: the order of calls, every loss, the final
    parameters.  `plotting` (matplotlib / plotly) is an import-only placeholder and the
    PriorGenerator (needs parsing-net checkpoints) returns a fixed map."""
    sys.path.insert(0, os.path.dirname(OUT))
    import model_cases as mc
    _import_reference_model()
    plotting = types.ModuleType("plotting")
    plotting.plot_predicted_depth_map = plotting.plot_reconstructions = lambda *a, **k: None
    sys.modules["plotting"] = plotting
    import GAN2Shape.trainer as ref_trainer

    class FixedPrior:
        def __init__(self, *a, **k):
            pass

        def __call__(self, image, *a, **k):
            return torch.full((1, 8, 8), 0.97)
    ref_trainer.PriorGenerator = FixedPrior
    ref_trainer.tqdm = lambda it, *a, **k: _Quiet(it)
    with _cuda_is_identity():
        t = ref_trainer.Trainer(mc.ToyStepModel, dict(mc.TOY_CFG))
        t.fit(mc.toy_dataset(), stages=mc.TOY_STAGES)
    out = {"log": np.array(t.model.log, np.float64),
           "params": np_(torch.cat([p.reshape(-1) for p in t.model.parameters()]))}

    # ---- the joint loop of BASELINE config 4: the reference's GeneralizingTrainer2.fit
    # (trainer.py:338-479 on GeneralizingTrainer.pretrain_on_prior :296-335) over 5 images in batches
    # of 2 (last batch ragged), 2 epochs: every call (kind, loss, batch size, digest of the
    # hand-off it received) and the final parameters
    class ImagePrior(FixedPrior):
        def __call__(self, image, *a, **k):
            return mc.toy_prior(image)
    ref_trainer.PriorGenerator = ImagePrior
    with _cuda_is_identity():
        t = ref_trainer.GeneralizingTrainer2(mc.ToyJointModel, dict(mc.TOY_JOINT_CFG))
        t.fit(mc.toy_dataset(5), stages=mc.TOY_JOINT_STAGES, batch_size=2)
    out["joint.log"] = np.array(t.model.log, np.float64)
    out["joint.params"] = np_(torch.cat([p.reshape(-1) for p in t.model.parameters()]))
    np.savez_compressed(os.path.join(OUT, "trainer.npz"), **out)


def lpips_golden():
    """The reference's PerceptualLoss(model='net-lin', net='vgg') (lpips/__init__.py:12-39 ->
    dist_model.py:28-99 -> networks_basic.py:27-110 PNetLin / ScalingLayer / NetLinLayer /
    spatial_average, pretrained_networks.py:97-135 vgg16 slices) run on the CPU with its own
    `weights/v0.1/vgg.pth` linear layers.  Absent offline: skimage and IPython (import-only
    placeholders — nothing of them is called on this path) and torchvision; the placeholder
    `torchvision.models.vgg16().features` is tests/model_cases.vgg16_features (torchvision's layout,
    SEEDED weights), so the reference's own slice indices define the five taps.  What this pins:
    scaling layer, slice taps, channel normalisation, squared difference, 1x1 `lin` layers, spatial
    mean, argument order and the gradient w.r.t. the prediction.  NOT pinned: torchvision's
    pretrained VGG16 weights (not in the reference, not downloadable)."""
    sys.path.insert(0, os.path.dirname(OUT))
    import model_cases as mc
    for name in ("skimage", "skimage.color", "skimage.transform", "IPython", "torchvision", "torchvision.models"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage"].color, sys.modules["skimage"].transform = sys.modules["skimage.color"], sys.modules["skimage.transform"]
    sys.modules["IPython"].embed = lambda *a, **k: None
    tvm = sys.modules["torchvision.models"]
    sys.modules["torchvision"].models = tvm
    seed = mc.LPIPS_CFG["vgg_seed"]
    tvm.vgg16 = lambda pretrained=True, **k: types.SimpleNamespace(features=mc.vgg16_features(seed))
    sys.path.insert(0, SG2)
    import lpips as ref_lpips
    # the reference reads its vgg.pth with a bare torch.load (lpips/dist_model.py:75): force the loader that
    # executes nothing from the file, whatever this torch's default is
    plain_load = torch.load
    torch.load = lambda *a, **k: plain_load(*a, **{**k, "weights_only": True})
    try:
        P = ref_lpips.PerceptualLoss(model='net-lin', net='vgg', use_gpu=False)
    finally:
        torch.load = plain_load
    net = P.model.net
    assert not net.training and net.version == '0.1' and net.lpips and not net.spatial
    out = {}
    for k in range(5):  # the five learned 1x1 layers of weights/v0.1/vgg.pth (1472 numbers): data
        out[f"lin{k}"] = np_(getattr(net, f"lin{k}").model[1].weight)
    for name, (B, S, s_in) in mc.LPIPS_CFG["cases"].items():
        pred, target = mc.lpips_inputs(B, S, s_in)
        pred = pred.requires_grad_(True)
        val = P(pred, target)
        cot = torch.linspace(0.5, 1.5, B).view(B, 1, 1, 1)
        (gp,) = torch.autograd.grad((val * cot).sum(), pred)
        _, per_layer = net.forward(target, pred.detach(), retPerLayer=True)
        out[f"{name}.val"], out[f"{name}.gpred"] = np_(val), np_(gp)
        out[f"{name}.per_layer"] = np.stack([np_(r).reshape(B) for r in per_layer])
        # the reference in float64: measures the reference's own fp32 error, so the GPU bound is
        # derived rather than guessed
        P64 = ref_lpips.PerceptualLoss(model='net-lin', net='vgg', use_gpu=False)
        P64.model.net.double()
        p64 = pred.detach().double().requires_grad_(True)
        v64 = P64(p64, target.double())
        (g64,) = torch.autograd.grad((v64 * cot.double()).sum(), p64)
        out[f"{name}.val64"], out[f"{name}.gpred64"] = np_(v64), np_(g64).astype(np.float32)
        out[f"{name}.ref_fp32_err"] = np.array([float((val.double() - v64).abs().max() / v64.abs().max()),
                                                float((gp.double() - g64).norm() / g64.norm())])
    np.savez_compressed(os.path.join(OUT, "lpips.npz"), **out)


class _Quiet:
    """tqdm stand-in: iterable with the two methods trainer.py calls on it."""
    def __init__(self, it):
        self.it = it

    def __iter__(self):
        return iter(self.it)

    def __len__(self):
        return len(self.it)

    def set_description(self, *a, **k):
        pass


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

    # ---- the BASELINE-size generator (configs/face.yml: size 128, z 512, 8 mapping layers,
    # channel_multiplier 1) at batch 8: image and latent gradient through the reference's own
    # Generator.forward (stylegan2-pytorch/model.py:545-627).  Weights are NOT stored: the test
    # refills its state-dict-compatible module with fill_deterministic(seed 77).
    out = {}
    g = sg2.Generator(128, 512, 8, channel_multiplier=1)
    fill_deterministic(g, 77)
    g.eval()
    torch.manual_seed(0)
    w = (0.5 * torch.randn(8, 512)).requires_grad_(True)
    img, _ = g([w], input_is_w=True, randomize_noise=False)
    gy = torch.randn(img.shape, generator=torch.Generator().manual_seed(5))  # regenerated by the test
    (gw,) = torch.autograd.grad(img, w, gy)
    out["g128.w"], out["g128.img"], out["g128.gw"] = np_(w), np_(img), np_(gw)
    # the same gradient through the reference module in float64: the fp32 reference itself is
    # 7.7e-4 * max|gw| (L2: 5.4e-4) away from it — sums over 131072 pixels x 17 layers — so the
    # test holds the GPU result to the float64 value at that scale instead of to fp32 noise
    g64 = sg2.Generator(128, 512, 8, channel_multiplier=1)
    fill_deterministic(g64, 77)
    g64 = g64.eval().double()
    w64 = w.detach().double().requires_grad_(True)
    img64, _ = g64([w64], input_is_w=True, randomize_noise=False)
    (gw64,) = torch.autograd.grad(img64, w64, gy.double())
    out["g128.gw64"] = np_(gw64)
    out["g128.ref_fp32_err"] = np.array([float((gw.double() - gw64).abs().max() / gw64.abs().max()),
                                         float((gw.double() - gw64).norm() / gw64.norm())])
    d = sg2.Discriminator(128, channel_multiplier=1)
    fill_deterministic(d, 78)
    d.eval()
    # pre-activations of every leaky ReLU (FusedLeakyReLU: input + bias; op/fused_act.py:86-92) within
    # fp32 rounding of zero: the units whose slope (0.2 or 1) is decided by summation order
    near_zero = np.zeros(3, np.int64)   # |pre| < 1e-6 std, < 1e-5 std, total

    def count_near_zero(mod, inp):
        pre = (inp[0] + mod.bias.view(1, -1, 1, 1)).detach()
        s_ = float(pre.std())
        near_zero[:] += [int((pre.abs() < 1e-6 * s_).sum()), int((pre.abs() < 1e-5 * s_).sum()), pre.numel()]
    hooks = [m_.register_forward_pre_hook(count_near_zero) for m_ in d.modules() if type(m_).__name__ == "FusedLeakyReLU"]
    x = torch.tanh(img.detach()[:2] / img.detach().abs().max() * 3).requires_grad_(True)
    _, feats = d(x, ftr_num=4)
    for h_ in hooks:
        h_.remove()
    out["d128.near_zero_preacts"] = near_zero
    gf = [torch.randn(f.shape, generator=torch.Generator().manual_seed(10 + i)) for i, f in enumerate(feats)]
    (gx,) = torch.autograd.grad(feats, x, gf)
    out["d128.gx"] = np_(gx)
    for i, f in enumerate(feats):   # per-level summaries (the maps themselves are 8 MB): mean, abs-mean, 4x4 pooled
        out[f"d128.f{i}.pool"] = np_(torch.nn.functional.adaptive_avg_pool2d(f, 4))
        out[f"d128.f{i}.absmean"] = np_(f.abs().mean((1, 2, 3)))
    # the same gradient through the reference module in float64, and how far the reference's OWN fp32
    # run is from it (L2, share of elements beyond 5e-3 max, max / max): units that sit within rounding
    # of a leaky-ReLU kink take the other slope — the GPU result is held to this measured scatter
    d64 = sg2.Discriminator(128, channel_multiplier=1)
    fill_deterministic(d64, 78)
    d64 = d64.eval().double()
    x64 = x.detach().double().requires_grad_(True)
    _, feats64 = d64(x64, ftr_num=4)
    (gx64,) = torch.autograd.grad(feats64, x64, [t.double() for t in gf])
    out["d128.gx64"] = np_(gx64).astype(np.float32)
    e = (gx.double() - gx64).abs()
    out["d128.ref_fp32_err"] = np.array([float((gx.double() - gx64).norm() / gx64.norm()),
                                         float((e > 5e-3 * gx64.abs().max()).double().mean()),
                                         float(e.max() / gx64.abs().max())])
    np.savez_compressed(os.path.join(OUT, "gan128.npz"), **out)


if __name__ == "__main__":
    # python tests/golden/make_golden.py [ops geometry misc gan model steps trainer lpips]   (default: all)
    for name in sys.argv[1:] or ["ops", "geometry", "misc", "gan", "model", "steps", "trainer", "lpips"]:
        globals()[f"{name}_golden"]()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
