"""-m gpu: the StyleGAN2 host mirror and the GAN2Shape steps on the HIP kernels.

Generator / Discriminator / mapping network are compared with outputs of the reference's own
modules (tests/golden/gan.npz, mapping.npz: weights are regenerated from a seed in sorted
state-dict order, exactly as tests/golden/make_golden.py filled the reference modules).
warp_canon_depth is checked at the BASELINE size (S = 128) against the brute-force oracle."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import capi  # noqa: E402
from oracle import geometry as og  # noqa: E402


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def fill_deterministic(module, seed):
    """Same recipe as tests/golden/make_golden.py:fill_deterministic."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for k in sorted(sd.keys()):
            t = sd[k]
            if not t.is_floating_point() or k.endswith("kernel"):
                continue
            v = torch.randn(t.shape, generator=g)
            if "modulation.bias" in k:
                v = 1 + 0.1 * v
            elif k.endswith("bias") or "noise" in k:
                v = 0.1 * v
            t.copy_(v)


@pytest.fixture(scope="module")
def sg2():
    import gan2shape_amd  # noqa: F401
    from gan2shape_amd import lib, stylegan2
    lib.load()
    return stylegan2


def test_mapping_network_golden(sg2, golden):
    g = golden("mapping")
    G = sg2.Generator(8, 32, 4, channel_multiplier=1)
    with torch.no_grad():
        for i in range(1, 5):
            G.style[i].weight.copy_(torch.tensor(g[f"style.{i}.weight"]))
            G.style[i].bias.copy_(torch.tensor(g[f"style.{i}.bias"]))
    G = G.cuda().requires_grad_(False)
    z = dev(g["style.z"])
    np.testing.assert_allclose(G.style_forward(z).cpu().numpy(), g["style.full"], rtol=1e-5, atol=1e-6)
    d3 = G.style_forward(z, depth=3)
    np.testing.assert_allclose(d3.cpu().numpy(), g["style.depth3"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(G.style_forward(d3, skip=3).cpu().numpy(), g["style.skip3"], rtol=1e-5, atol=1e-6)


def test_generator_golden(sg2, golden):
    g = golden("gan")
    G = sg2.Generator(16, 32, 3, channel_multiplier=1)
    assert len(G.state_dict()) == int(g["g.n_keys"])
    fill_deterministic(G, 123)
    G = G.cuda().eval()
    w = dev(g["g.w"]).requires_grad_(True)
    img, _ = G([w], input_is_w=True, randomize_noise=False)
    (gw,) = torch.autograd.grad(img, w, dev(g["g.gy"]))
    scale = np.abs(g["g.img"]).max()
    np.testing.assert_allclose(img.detach().cpu().numpy(), g["g.img"], atol=1e-4 * scale, rtol=1e-4)
    np.testing.assert_allclose(gw.cpu().numpy(), g["g.gw"], atol=2e-4 * np.abs(g["g.gw"]).max(), rtol=1e-3)
    with torch.no_grad():
        img2, _ = G([dev(g["g.z"])], input_is_w=False, randomize_noise=False)
    np.testing.assert_allclose(img2.cpu().numpy(), g["g.img_from_z"], atol=1e-4 * scale, rtol=1e-4)
    # the frozen-generator path (requires_grad False: fused noise+bias+act) gives the same image
    G.requires_grad_(False)
    img3, _ = G([w], input_is_w=True, randomize_noise=False)
    torch.testing.assert_close(img3, img, rtol=1e-5, atol=1e-5 * scale)
    (gw3,) = torch.autograd.grad(img3, w, dev(g["g.gy"]))
    torch.testing.assert_close(gw3, gw, rtol=1e-4, atol=1e-4 * float(gw.abs().max()))


def test_discriminator_golden(sg2, golden):
    g = golden("gan")
    D = sg2.Discriminator(16, channel_multiplier=1)
    fill_deterministic(D, 321)
    D = D.cuda().eval()
    x = dev(g["d.x"]).requires_grad_(True)
    _, feats = D(x, ftr_num=2)
    assert len(feats) == 2
    (gx,) = torch.autograd.grad(feats, x, [dev(g["d.gf0"]), dev(g["d.gf1"])])
    for i, f in enumerate(feats):
        ref = g[f"d.f{i}"]
        np.testing.assert_allclose(f.detach().cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max(), rtol=1e-3)
    np.testing.assert_allclose(gx.cpu().numpy(), g["d.gx"], atol=3e-4 * np.abs(g["d.gx"]).max(), rtol=1e-3)
    score, _ = D(x)
    np.testing.assert_allclose(score.detach().cpu().numpy(), g["d.score"],
                               atol=3e-4 * np.abs(g["d.score"]).max(), rtol=1e-3)


@pytest.fixture(scope="module")
def trainer():
    import bench
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import Trainer
    torch.manual_seed(0)
    t = Trainer(GAN2Shape, bench.face_config(n_proj=3), device="cuda")
    t.sample = bench.synthetic_sample(t.model, 1234, torch.device("cuda"))
    return t


def _grad_norms(model):
    out = {}
    for name in model.NETS:
        ps = [p.grad for p in getattr(model, f"{name}_net").parameters() if p.grad is not None]
        out[name] = float(sum((g.double() ** 2).sum() for g in ps)) if ps else 0.0
    return out


def _zero(model):
    for p in model.parameters():
        p.grad = None


def test_warp_canon_depth_full_size_vs_oracle(trainer):
    """renderer.py:116-125 at S = 128 against the brute-force oracle (bit-exact depth)."""
    m = trainer.model
    S = 128
    torch.manual_seed(1)
    base = torch.tensor(og.Geometry(S).get_normal_from_depth(np.ones((1, S, S), np.float32))[..., 2])
    depth = (1.0 + 0.06 * torch.sin(torch.arange(S).float()[None, :, None] / 7)
             * torch.cos(torch.arange(S).float()[None, None, :] / 11) * base).cuda()
    view = torch.tensor([[0.35, -0.6, 0.1, 0.03, -0.02, 0.05]], device="cuda")
    m.renderer.set_transform_matrices(view)
    with torch.no_grad():
        out = m.renderer.warp_canon_depth(depth)
        verts = m.renderer.get_warped_3d_grid(depth).reshape(1, -1, 3)
    faces = og.get_face_idx(1, S, S)[0]
    ref = capi.render_depth(verts.cpu().numpy(), faces, S, m.renderer.K[0].cpu().numpy())
    np.testing.assert_array_equal(out.cpu().numpy(), np.clip(ref["depth"], 0.8, 1.2).astype(np.float32))
    assert (out < 1.2).float().mean() > 0.5


def test_three_steps_train_the_right_networks(trainer):
    m = trainer.model
    image, latent = trainer.sample
    _zero(m)
    loss1, col1 = m.forward_step1(image, latent, None)
    assert torch.isfinite(loss1) and len(col1) == 6 and col1[4].shape == (1, 128, 128)
    loss1.backward()
    n = _grad_norms(m)
    assert n["albedo"] > 0 and n["depth"] == n["viewpoint"] == n["lighting"] == n["offset_encoder"] == 0
    assert all(p.grad is None for p in m.generator.parameters())

    _zero(m)
    loss2, col2 = m.forward_step2(image, latent, col1, n_proj_samples=3)
    assert torch.isfinite(loss2)
    proj, mask = col2
    assert proj.shape == (3, 3, 128, 128) and mask.shape == (3, 1, 128, 128) and proj.is_cuda
    loss2.backward()
    n = _grad_norms(m)
    assert n["offset_encoder"] > 0 and n["albedo"] == n["depth"] == n["viewpoint"] == n["lighting"] == 0

    _zero(m)
    loss3, col3 = m.forward_step3(image, latent, col2)
    assert torch.isfinite(loss3) and col3 is None
    loss3.backward()
    n = _grad_norms(m)
    assert n["depth"] > 0 and n["viewpoint"] > 0 and n["lighting"] > 0 and n["albedo"] > 0
    assert n["offset_encoder"] == 0
    rim, rdepth = m.evaluate_results(image)
    assert rim.shape == (1, 3, 128, 128) and rdepth.shape == (1, 128, 128)


def test_step2_is_reproducible_under_a_seed(trainer):
    m = trainer.model
    image, latent = trainer.sample
    with torch.no_grad():
        _, col1 = m.forward_step1(image, latent, None)
    vals = []
    for _ in range(2):
        torch.manual_seed(7)
        loss, _ = m.forward_step2(image, latent, col1, n_proj_samples=3)
        vals.append(loss.item())
    # same random draws; split-K float atomics in the small modconv layers reorder sums
    assert abs(vals[0] - vals[1]) < 1e-5 * abs(vals[0])


def test_trainer_fit_and_checkpoint_roundtrip(trainer, tmp_path):
    from gan2shape_amd.model import GAN2Shape
    image, latent = trainer.sample
    data = [(image[0].cpu(), latent[0].cpu(), 0), (image[0].cpu().flip(2), latent[0].cpu(), 1)]
    trainer.n_epochs_prior = 2
    trainer.history.clear()
    n = trainer.fit(data, stages=[{'step1': 1, 'step2': 1, 'step3': 1}])
    assert n == 6 and len(trainer.history) == 6
    assert all(math.isfinite(h[3]) for h in trainer.history)
    # rank 1 of 2 sees only image 1
    trainer.history.clear()
    trainer.fit(data, stages=[{'step1': 1, 'step2': 0, 'step3': 0}], rank=1, world_size=2)
    assert {h[0] for h in trainer.history} == {1}
    m = trainer.model
    m.ckpt_paths = {'VLADE_nets': str(tmp_path)}
    m.save_checkpoint(0, 0, 6, 'face')
    paths, ids = m.build_checkpoint_path(str(tmp_path), 'face')
    assert ids == [0]
    before = {k: v.clone() for k, v in m.depth_net.state_dict().items()}
    with torch.no_grad():
        for p in m.depth_net.parameters():
            p.add_(1.0)
    m.load_from_checkpoint(paths[0])
    for k, v in m.depth_net.state_dict().items():
        torch.testing.assert_close(v, before[k])
    assert set(GAN2Shape.NETS) == {'lighting', 'viewpoint', 'depth', 'albedo', 'offset_encoder'}


def test_lpips_fused_tail_matches_reference_form():
    """The one-node LPIPS (lpips._VggLpips: fused tails, hand-written trunk backward) vs the op-by-op form
    of lpips/networks_basic.py:64-92 (same module, the unfused branch is taken when the target requires a
    gradient).  The two halves of the trunk forward are launched apart here, so that both forms see
    bit-identical activations; what the batch-2N forward adds (ReLU / max-pool tie decisions) is measured
    in tests/test_gpu_golden.py::test_lpips_gpu_vs_reference_run."""
    import gan2shape_amd  # noqa: F401
    from gan2shape_amd import lpips as lp
    from gan2shape_amd.lpips import PerceptualLoss
    torch.manual_seed(0)
    p = PerceptualLoss().cuda()
    with torch.no_grad():
        for k in range(5):
            getattr(p.net, f"lin{k}").model[-1].weight.uniform_(0.0, 1.0)
    lp._VggLpips.SPLIT_FORWARD = True
    try:
        for B in (1, 3):
            pred = (torch.rand(B, 3, 64, 64, device="cuda") * 2 - 1).requires_grad_(True)
            target = torch.rand(B, 3, 64, 64, device="cuda") * 2 - 1
            fused = p(pred, target)
            (g_fused,) = torch.autograd.grad(fused.sum(), pred)
            ref = p(pred, target.clone().requires_grad_(True))
            (g_ref,) = torch.autograd.grad(ref.sum(), pred)
            assert fused.shape == ref.shape == (B, 1, 1, 1)
            torch.testing.assert_close(fused, ref, rtol=2e-5, atol=1e-7)
            torch.testing.assert_close(g_fused, g_ref, rtol=2e-4, atol=2e-5 * float(g_ref.abs().max()) + 1e-9)
    finally:
        lp._VggLpips.SPLIT_FORWARD = False
    # and the batch-2N forward gives the same VALUE (decisions may move gradient patches, not the sum)
    pred = (torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1)
    target = torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1
    torch.testing.assert_close(p(pred, target), p(pred, target.clone().requires_grad_(True)), rtol=2e-5, atol=1e-7)


def test_fused_geometry_matches_torch_path():
    """csrc/geometry.hip kernels vs the torch restatement of renderer.py / losses.py they replace
    (values and gradients w.r.t. depth and view)."""
    import gan2shape_amd  # noqa: F401
    from gan2shape_amd.losses import SmoothLoss
    from gan2shape_amd.renderer import Renderer
    torch.manual_seed(0)
    S, B = 64, 3
    R = Renderer({"rot_center_depth": 1.0, "fov": 10}, S, 0.9, 1.1, device="cuda")
    depth0 = (0.9 + 0.2 * torch.rand(B, S, S, device="cuda"))
    view0 = torch.randn(B, 6, device="cuda") * 0.3
    scales = (math.pi / 180 * 60, 0.1, 0.1)
    outs = {}
    for fused in (True, False):
        R.fused = fused
        depth = depth0.clone().requires_grad_(True)
        view = view0.clone().requires_grad_(True)
        R.set_view(view, *scales)
        rot, trans = R.rot_mat, R.trans_xyz
        verts = R.get_warped_3d_grid(depth)
        grid = R.get_inv_warped_2d_grid(depth)
        sm = SmoothLoss()(depth) if fused else _smooth_ref(depth)
        gen = torch.Generator(device="cuda").manual_seed(1)
        wv = torch.randn(verts.shape, device="cuda", generator=gen)
        wg = torch.randn(grid.shape, device="cuda", generator=gen)
        loss = (verts * wv).sum() + (grid * wg).sum() * 0.01 + 3.0 * sm
        gd, gvw = torch.autograd.grad(loss, (depth, view))
        outs[fused] = dict(rot=rot, trans=trans, verts=verts, grid=grid, sm=sm, gd=gd, gv=gvw)
    a, b = outs[True], outs[False]
    for k in ("rot", "trans", "verts"):
        torch.testing.assert_close(a[k], b[k], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a["grid"], b["grid"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(a["sm"], b["sm"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(a["gd"], b["gd"], rtol=1e-3, atol=1e-4 * float(b["gd"].abs().max()))
    torch.testing.assert_close(a["gv"], b["gv"], rtol=1e-3, atol=1e-4 * float(b["gv"].abs().max()))
    # 4-D map (diffuse shading) through the fused smoothness loss
    x = torch.rand(2, 1, 20, 33, device="cuda", requires_grad=True)
    l1, l2 = SmoothLoss()(x), _smooth_ref(x)
    torch.testing.assert_close(l1, l2, rtol=1e-5, atol=1e-7)
    (g1,), (g2,) = torch.autograd.grad(l1, x), torch.autograd.grad(l2, x)
    torch.testing.assert_close(g1, g2, rtol=1e-5, atol=1e-8)


def test_fused_normal_and_shading_match_torch_path():
    import gan2shape_amd  # noqa: F401
    import bench
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.renderer import Renderer
    torch.manual_seed(0)
    S = 48
    R = Renderer({"rot_center_depth": 1.0, "fov": 10}, S, 0.9, 1.1, device="cuda")
    d0 = 0.9 + 0.2 * torch.rand(2, S, S, device="cuda")
    w = torch.randn(2, S, S, 3, device="cuda")
    res = {}
    for fused in (True, False):
        R.fused = fused
        d = d0.clone().requires_grad_(True)
        n = R.get_normal_from_depth(d)
        (g,) = torch.autograd.grad((n * w).sum(), d)
        res[fused] = (n, g)
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(res[True][1], res[False][1], rtol=1e-3, atol=1e-3 * float(res[False][1].abs().max()))

    m = GAN2Shape.__new__(GAN2Shape)          # only the shading helpers are exercised
    torch.nn.Module.__init__(m)
    m.renderer = R
    for B, Bn, Ba in [(3, 3, 3), (4, 1, 1), (1, 1, 1)]:
        normal0 = torch.nn.functional.normalize(torch.randn(Bn, S, S, 3, device="cuda"), dim=3)
        light0 = torch.randn(B, 4, device="cuda") * 0.5
        albedo0 = torch.tanh(torch.randn(Ba, 3, S, S, device="cuda"))
        wt, wd = torch.randn(B, 3, S, S, device="cuda"), torch.randn(B, 1, S, S, device="cuda")
        out = {}
        for fused in (True, False):
            R.fused = fused
            n_, l_, a_ = (t.clone().requires_grad_(True) for t in (normal0, light0, albedo0))
            la, lb, dif, tex = m._shade(n_, l_, a_)
            loss = (tex * wt).sum() + (dif * wd).sum() + la.sum() * 0.3 + lb.sum() * 0.7
            out[fused] = (la, lb, dif, tex) + torch.autograd.grad(loss, (n_, l_, a_))
        for x, y in zip(out[True], out[False]):
            torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-5 * (1 + float(y.abs().max())))


def _smooth_ref(pred):
    """losses.py:54-79 in plain torch ops."""
    p = pred.reshape(-1, pred.size(-2), pred.size(-1))
    dx, dy = p[:, :, 1:] - p[:, :, :-1], p[:, 1:] - p[:, :-1]
    dx2, dxdy = dx[:, :, 1:] - dx[:, :, :-1], dx[:, 1:] - dx[:, :-1]
    dydx, dy2 = dy[:, :, 1:] - dy[:, :, :-1], dy[:, 1:] - dy[:, :-1]
    return dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()


def test_vgg_trunk_on_mfma_kernel_matches_miopen():
    """conv3x3 + bias + ReLU through g2s_conv_bias_act vs torch.nn (MIOpen), features and the
    gradient w.r.t. the input image."""
    import gan2shape_amd  # noqa: F401
    from gan2shape_amd.lpips import VGG16Features
    torch.manual_seed(0)
    v = VGG16Features().cuda()
    for B in (1, 3):
        x0 = torch.rand(B, 3, 64, 64, device="cuda") * 2 - 1
        res = {}
        for fused in (True, False):
            next(v.parameters()).requires_grad_(not fused)   # a trainable weight forces the torch path
            x = x0.clone().requires_grad_(True)
            feats = v(x)
            loss = sum((f * f).mean() for f in feats)
            (g,) = torch.autograd.grad(loss, x)
            res[fused] = (feats, g)
        next(v.parameters()).requires_grad_(False)
        for a, b in zip(res[True][0], res[False][0]):
            torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-4 * float(b.abs().max()))
        # a feature within rounding of zero may flip its ReLU gate between the two summation
        # orders: compare the gradients in the L2 sense, and bound the worst element loosely
        ga, gb = res[True][1], res[False][1]
        assert float((ga - gb).norm() / gb.norm()) < 2e-3
        assert float((ga - gb).abs().max()) < 2e-2 * float(gb.abs().max())


def test_step3_batched_nets_equal_reference_order(trainer):
    """forward_step3 runs V / L / LPIPS once on (image + projected samples); the reference calls
    them twice (model.py:113-131,243-250,159,275).  Same loss and gradients."""
    import torch.nn.functional as F
    m = trainer.model
    image, latent = trainer.sample
    with torch.no_grad():
        _, col1 = m.forward_step1(image, latent, None)
        torch.manual_seed(3)
        _, col2 = m.forward_step2(image, latent, col1, n_proj_samples=3)
    _zero(m)
    loss_b, _ = m.forward_step3(image, latent, col2)
    loss_b.backward()
    gb = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    # reference order, literally (two calls of each net / of LPIPS)
    _zero(m)
    projected, masks = col2
    step1_loss, collected = m.forward_step1(image, None, None, step1=False)
    normal, _, _, albedo, depth, _ = collected
    b = len(projected)
    view = m.viewpoint_net(projected) + m.view_light_sampler.view_mean.unsqueeze(0)
    m._set_view(view)
    light = m.lighting_net(projected) + m.view_light_sampler.light_mean.unsqueeze(0)
    _, _, _, texture = m._shade(normal, light, albedo)
    recon_depth = m.renderer.warp_canon_depth(depth.expand(b, 128, 128))
    grid = m.renderer.get_inv_warped_2d_grid(recon_depth)
    mask = (recon_depth < 1.2).float().unsqueeze(1).detach() * masks
    recon = F.grid_sample(texture, grid, mode='bilinear', align_corners=True).clamp(min=-1, max=1)
    loss_r = step1_loss + m.photometric_loss(recon, projected, mask=mask) + \
        torch.mean(m.perceptual_loss(recon * mask, projected * mask))
    loss_r.backward()
    assert abs(loss_b.item() - loss_r.item()) < 2e-5 * abs(loss_r.item())
    # split-K atomics reorder fp32 sums between the two runs; an activation input within rounding of
    # zero then takes the other slope, which moves single small tensors by up to a few per cent:
    # every tensor within 5e-2, all gradients together within 5e-3
    err2 = ref2 = 0.0
    for n, p in m.named_parameters():
        if p.grad is not None:
            ref = p.grad
            e = float((gb[n] - ref).norm())
            assert e <= 5e-2 * float(ref.norm()) + 1e-9, n
            err2 += e * e
            ref2 += float(ref.norm()) ** 2
    assert err2 ** 0.5 <= 5e-3 * ref2 ** 0.5


def test_trainer_fit_with_hip_graphs():
    """Trainer.fit(graphs=True): same loop, iterations replayed as HIP graphs; the result of a few
    iterations stays close to the eager run from the same initial state (random draws differ)."""
    import bench
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import Trainer
    dev = torch.device("cuda")
    torch.cuda.set_stream(torch.cuda.Stream(dev))     # captures need a non-default stream
    cfg = bench.face_config(n_proj=2)
    stages = [{'step1': 6, 'step2': 5, 'step3': 5}]
    losses = {}
    for graphs in (False, True):
        torch.manual_seed(0)
        t = Trainer(GAN2Shape, cfg, device=dev, capturable=True)
        image, latent = bench.synthetic_sample(t.model, 1234, dev)
        n = t.fit([(image[0].cpu(), latent[0].cpu(), 0)], stages=stages, graphs=graphs)
        assert n == 16
        losses[graphs] = [h[3] for h in t.history]
        assert all(math.isfinite(v) for v in losses[graphs])
    # step 1 is deterministic given the initial weights: the 6th-iteration losses agree
    assert abs(losses[True][0] - losses[False][0]) < 5e-3 * abs(losses[False][0])
    torch.cuda.set_stream(torch.cuda.default_stream(dev))


def test_joint_trainer_on_gpu():
    """GeneralizingTrainer2 (trainer.py:338-479) with the real model: prior pre-training over all
    images, step 1 on a batch of 2 images, steps 2 / 3 image by image; finite losses, every trained
    net moves."""
    import bench
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import GeneralizingTrainer2
    dev = torch.device("cuda")
    cfg = bench.face_config(n_proj=2)
    cfg.update(n_epochs_prior=2, n_epochs_generalized=1)
    torch.manual_seed(0)
    t = GeneralizingTrainer2(GAN2Shape, cfg, device=dev)
    data = []
    for i in range(2):
        image, latent = bench.synthetic_sample(t.model, 100 + i, dev)
        data.append((image[0].cpu(), latent[0].cpu(), i))
    before = {n: p.detach().clone() for n, p in t.model.named_parameters() if p.requires_grad}
    n = t.fit(data, stages=[{'step1': 2, 'step2': 1, 'step3': 1}], batch_size=2)
    assert n == 2 + 2 * (1 + 1)
    assert all(h[3] is None or math.isfinite(h[3]) for h in t.history) and len(t.history) == 1 + 2 * 2
    moved = {n.split('.')[0] for n, p in t.model.named_parameters()
             if n in before and not torch.equal(before[n], p.detach())}
    assert {'depth_net', 'albedo_net', 'viewpoint_net', 'lighting_net', 'offset_encoder_net'} <= moved


def _joint_dp_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    from gan2shape_amd import sharding
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import GeneralizingTrainer2
    sharding.init_distributed("gloo")      # both ranks share the one GPU of the test box
    dev = torch.device("cuda", 0)
    cfg = bench.face_config(n_proj=2)
    cfg.update(n_epochs_prior=1, n_epochs_generalized=1)
    torch.manual_seed(0)
    t = GeneralizingTrainer2(GAN2Shape, cfg, device=dev)
    data = []
    for i in range(2):
        image, latent = bench.synthetic_sample(t.model, 100 + i, dev)
        data.append((image[0].cpu(), latent[0].cpu(), i))
    n = t.fit(data, stages=[{'step1': 1, 'step2': 1, 'step3': 1}], batch_size=2, rank=rank, world_size=world)
    flat = torch.cat([p.detach().reshape(-1).cpu() for nme, p in t.model.named_parameters()
                      if nme.split('.')[0] in ('depth_net', 'albedo_net', 'viewpoint_net', 'lighting_net',
                                               'offset_encoder_net')])
    q.put((rank, n, float(flat.double().sum()), float(flat.double().abs().sum()), bool(torch.isfinite(flat).all())))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_joint_trainer_data_parallel_two_ranks_on_gpu():
    """GeneralizingTrainer2 with world_size 2 (gloo between two processes on the one GPU): each rank
    trains its image of the batch, gradients are averaged before every optimiser step, the depth
    centre is the all-rank mean — both ranks must end with the same parameters."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_joint_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=280) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, s0, a0, ok0), (_, n1, s1, a1, ok1) = res
    assert ok0 and ok1 and n0 == n1 == 1 + 1 * (1 + 1)
    assert abs(s0 - s1) <= 1e-6 * a0 and abs(a0 - a1) <= 1e-6 * a0


@pytest.mark.parametrize("image_size,gan_size,cm,prior,n_proj,joint", [
    (64, 64, 1, "ellipsoid", 2, False),         # BASELINE config 1 (minimal_config restated at 64x64), on the GPU
    (128, 256, 1, "smoothed_box", 16, False),   # config 3 (cat: configs/cat.yml, n_proj_samples 16)
    (128, 512, 2, "ellipsoid", 8, True)])       # config 4 (car: configs/car.yml, joint training)
def test_other_baseline_configs_run(image_size, gan_size, cm, prior, n_proj, joint):
    """BASELINE configs 1 / 3 / 4.  Config 1: the minimal single-image 64x64 run (ellipsoid prior, one
    stage; SURVEY §8d — the reference's OffsetEncoder(64) cannot be constructed, ours emits 512
    channels) as GPU plumbing: 2 prior pre-training steps + one iteration of each kind.  Config 3
    (cat: gan_size 256, smoothed-box prior, 16 projected samples): instance trainer.  Config 4 (car:
    gan_size 512, channel multiplier 2, 8 projected samples): the joint trainer
    (GeneralizingTrainer2, trainer.py:338-479) on two images — its one-rank form; the gradient
    all-reduce of the 8-GPU form is covered by the gloo tests.  G output is area-resized to 128.
    Finite losses, every trained net moves."""
    import bench
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import GeneralizingTrainer2, Trainer
    dev = torch.device("cuda")
    cfg = bench.face_config(n_proj=n_proj)
    cfg.update(image_size=image_size, gan_size=gan_size, channel_multiplier=cm, prior_name=prior,
               n_epochs_prior=2 if image_size == 64 else 1, n_epochs_generalized=1,
               category={64: "face", 256: "cat", 512: "car"}[gan_size])
    torch.manual_seed(0)
    t = (GeneralizingTrainer2 if joint else Trainer)(GAN2Shape, cfg, device=dev)
    g = torch.Generator().manual_seed(7)
    data = []
    for i in range(2 if joint else 1):
        image = torch.tanh(torch.nn.functional.interpolate(
            torch.randn(1, 3, image_size // 4, image_size // 4, generator=g), scale_factor=4, mode="bilinear")).to(dev)
        with torch.no_grad():
            latent = t.model.generator.style_forward(torch.randn(1, 512, generator=g).to(dev))
        data.append((image[0].cpu(), latent[0].cpu(), i))
    before = {n: p.detach().clone() for n, p in t.model.named_parameters() if p.requires_grad}
    if joint:
        n = t.fit(data, stages=[{'step1': 1, 'step2': 1, 'step3': 1}], batch_size=2)
        assert n == 1 + 2 * (1 + 1) and len(t.history) == 1 + 2 * 2
    else:
        n = t.fit(data, stages=[{'step1': 1, 'step2': 1, 'step3': 1}])
        assert n == 3 and len(t.history) == 3
    assert all(h[3] is None or math.isfinite(h[3]) for h in t.history)
    moved = {k.split('.')[0] for k, p in t.model.named_parameters() if k in before and not torch.equal(before[k], p.detach())}
    assert {'depth_net', 'albedo_net', 'viewpoint_net', 'lighting_net', 'offset_encoder_net'} <= moved


def test_graph_capture_on_default_stream_raises():
    """An eager backward on the legacy default stream followed by a capture ends in a host segfault
    inside hipStreamEndCapture (gpurun_out/dbg.log of round 1): GraphedSteps.capture refuses to
    start from the default stream instead."""
    import bench
    from gan2shape_amd.graphs import GraphedSteps
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import Trainer
    dev = torch.device("cuda")
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
    torch.manual_seed(0)
    t = Trainer(GAN2Shape, bench.face_config(n_proj=2), device=dev, capturable=True)
    image, latent = bench.synthetic_sample(t.model, 1234, dev)
    g = GraphedSteps(t, image, latent)
    with pytest.raises(RuntimeError, match="non-default current stream"):
        g.capture(1)
    with pytest.raises(RuntimeError, match="non-default current stream"):
        t.fit([(image[0].cpu(), latent[0].cpu(), 0)], stages=[{'step1': 1, 'step2': 1, 'step3': 1}], graphs=True)
    with pytest.raises(RuntimeError, match="capturable"):
        Trainer(GAN2Shape, bench.face_config(n_proj=2), device=dev).fit([], graphs=True)


def _fresh_graph_fit_worker(q):
    import bench
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import Trainer
    dev = torch.device("cuda")
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    cfg = bench.face_config(n_proj=3)
    # fewer iterations than the default warm-up in some blocks; two stages, two images
    stages = [{'step1': 2, 'step2': 1, 'step3': 2}, {'step1': 4, 'step2': 3, 'step3': 1}]
    out = {}
    for graphs in (True, False):      # graphs FIRST: nothing has grown any cache before the captures
        torch.manual_seed(0)
        t = Trainer(GAN2Shape, cfg, device=dev, capturable=True)
        data = []
        for i in range(2):
            image, latent = bench.synthetic_sample(t.model, 1234 + i, dev)
            data.append((image[0].cpu(), latent[0].cpu(), i))
        n = t.fit(data, stages=stages, graphs=graphs)
        torch.cuda.synchronize()
        out[graphs] = (n, [(h[0], h[1], h[2], h[3]) for h in t.history])
    q.put(out)


def test_fresh_process_graph_fit_two_stages():
    """fit(graphs=True) in a process that has run nothing before: step 1 is captured at raster
    batch 1, step 2 / 3 then run at batch n_proj (the rasterizer scratch is allocated per call, so
    no captured graph keeps a pointer into a buffer that a later, larger call released).  Iteration
    counts equal the request even where a block asks for fewer iterations than the warm-up, and the
    deterministic step-1 blocks agree with the eager loop."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_fresh_graph_fit_worker, args=(q,))
    p.start()
    out = q.get(timeout=500)
    p.join(timeout=60)
    assert p.exitcode == 0
    (n_g, hist_g), (n_e, hist_e) = out[True], out[False]
    assert n_g == n_e == 2 * (2 + 1 + 2 + 4 + 3 + 1)
    assert [h[:3] for h in hist_g] == [h[:3] for h in hist_e] and len(hist_g) == 12
    assert all(math.isfinite(h[3]) for h in hist_g + hist_e)
    # image 0, stage 0, step 1: no random draw is involved before it -> same loss in both modes
    assert abs(hist_g[0][3] - hist_e[0][3]) < 5e-3 * abs(hist_e[0][3])


def test_bench_self_launches_ranks():
    """`python bench.py --gpus 2` without torchrun starts two ranks itself (gloo rehearsal on the one
    GPU of the test box) and reports n_gpus = 2 with the whole-job rate."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["G2S_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4",
                        "--warmup", "2", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["value"] > 0
    assert line["scaling"] == "weak" and line["cpu_baseline"] is None


def test_config5_fp16_operands_and_256():
    """BASELINE config 5 (256x256 face, confidence-map prior, fp16-operand MFMA): (a) the generator
    with fp16 operands stays within 1e-2 of its fp32 output, latent gradient within
    2e-2 (L2); (b) the 256x256 model — extended V / L / E nets, confidence prior from a synthetic soft
    mask — runs prior pre-training and one iteration of each kind with finite losses."""
    import bench
    from gan2shape_amd import modconv as mc
    from gan2shape_amd import stylegan2 as sg2
    from gan2shape_amd.model import GAN2Shape
    from gan2shape_amd.trainer import Trainer
    G = sg2.Generator(128, 512, 8, channel_multiplier=1)
    fill_deterministic(G, 77)
    G = G.cuda().eval().requires_grad_(False)
    torch.manual_seed(0)
    w = (0.5 * torch.randn(4, 512, device="cuda")).requires_grad_(True)
    out = {}
    saved = mc.OPERANDS
    try:
        for mode in ("f32", "f16"):
            mc.OPERANDS = mode
            img, _ = G([w], input_is_w=True, randomize_noise=False)
            (gw,) = torch.autograd.grad(img, w, torch.ones_like(img))
            out[mode] = (img.detach(), gw)
        scale = float(out["f32"][0].abs().max())
        assert float((out["f16"][0] - out["f32"][0]).abs().max()) <= 1e-2 * scale
        assert float((out["f16"][1] - out["f32"][1]).norm()) <= 2e-2 * float(out["f32"][1].norm())
        assert float((out["f16"][0] - out["f32"][0]).abs().max()) > 0          # the fp16 path did run
        # (b) the 256x256 configuration
        cfg = bench.face_config(n_proj=2, workload="face256_fp16")
        cfg.update(n_epochs_prior=1)
        torch.manual_seed(0)
        t = Trainer(GAN2Shape, cfg, device=torch.device("cuda"))
        assert mc.OPERANDS == "f16" and t.model.image_size == 256
        image, latent = bench.synthetic_sample(t.model, 3, torch.device("cuda"))
        assert tuple(image.shape) == (1, 3, 256, 256)
        n = t.fit([(image[0].cpu(), latent[0].cpu(), 0)], stages=[{'step1': 1, 'step2': 1, 'step3': 1}])
        assert n == 3 and all(math.isfinite(h[3]) for h in t.history)
    finally:
        mc.OPERANDS = saved
