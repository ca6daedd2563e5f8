"""-m "not gpu": host logic that needs no kernel — the geometry mirror against the golden vectors,
losses, resize, priors, state-dict compatibility with the reference modules (live, in the build
container only), the C-ABI surface, sharding helpers."""
import math
import os
import re
import sys

import numpy as np
import pytest
import torch

import gan2shape_amd  # noqa: F401
from gan2shape_amd import lib, losses, networks, priors, utils
from gan2shape_amd.renderer import Renderer
from gan2shape_amd.renderer import utils as ru

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SG2 = os.path.join(REF, "GAN2Shape/stylegan2/stylegan2-pytorch")
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")


def T(a):
    return torch.as_tensor(a)


# ----------------------------------------------------------------------------- C ABI
def _gfx950_code_objects(so_path):
    """The device code objects inside libg2s.so: section .hip_fatbin is a sequence of clang offload
    bundles (magic, entry count, then {offset, size, id length, id} per entry)."""
    import shutil
    import struct
    import subprocess
    import tempfile
    objcopy = shutil.which("objcopy")
    if objcopy is None:
        pytest.skip("objcopy not available")
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", so_path, fat], check=True)
        blob = open(fat, "rb").read()
    magic, pos, out = b"__CLANG_OFFLOAD_BUNDLE__", 0, []
    while (i := blob.find(magic, pos)) >= 0:
        (count,) = struct.unpack_from("<Q", blob, i + 24)
        p = i + 32
        for _ in range(count):
            off, size, idlen = struct.unpack_from("<QQQ", blob, p)
            ident = blob[p + 24:p + 24 + idlen].decode()
            p += 24 + idlen
            if "gfx950" in ident and size:
                out.append(blob[i + off:i + off + size])
        pos = i + len(magic)
    return out


def test_mfma_kernels_keep_buffer_descriptors_in_sgprs():
    """A buffer descriptor the compiler cannot prove wave-uniform turns EVERY load that uses it into
    a waterfall loop (v_readfirstlane x4, compare, s_and_saveexec, load, s_xor exec, branch) — the K
    loop of the convolution kernels then runs at half speed without any test failing.  Disassemble
    the shipped code objects and require that no buffer load sits inside such a loop."""
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    objs = _gfx950_code_objects(lib.LIB_PATH)
    assert len(objs) >= 10
    checked = 0
    for blob in objs:
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(blob)
            f.flush()
            asm = subprocess.run([objdump, "-d", f.name], check=True, capture_output=True, text=True).stdout
        lines = asm.splitlines()
        kernel = None
        for n, line in enumerate(lines):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                kernel = m.group(1)
                checked += any(k in kernel for k in ("modconv_kernel", "wino_kernel", "conv_wgrad_kernel"))
            if "s_xor_b64 exec, exec" in line:
                # a waterfall loop = v_readfirstlane of the descriptor dwords, s_and_saveexec, the load,
                # s_xor exec, s_cbranch_execnz; an ordinary divergent if / else around a load (s_or_saveexec
                # ... s_xor exec) has no readfirstlane in front of it
                window = " ".join(lines[max(0, n - 3):n])
                head = " ".join(lines[max(0, n - 12):n])
                looped = "v_readfirstlane" in head and "s_and_saveexec" in head
                assert not (looped and ("buffer_load" in window or "global_load" in window)), \
                    f"waterfall loop around a load in {kernel}: {head}"
    assert checked >= 8   # the MFMA kernel instances were among the disassembled symbols


def test_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "g2s.h")).read()
    declared = set(re.findall(r"\b(g2s_[a-z0-9_]+)\s*\(", header))
    assert {"g2s_raster_depth_fwd", "g2s_raster_depth_bwd", "g2s_fused_bias_act", "g2s_upfirdn2d",
            "g2s_modconv", "g2s_noise_bias_act", "g2s_raster_workspace_bytes", "g2s_last_error",
            "g2s_abi_version"} <= declared
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    L = lib.load()  # loads without a GPU; getattr fails for a missing symbol
    assert L.g2s_abi_version() == 1
    assert L.g2s_raster_workspace_bytes(8, 128 * 128, 2 * 127 * 127, 128) >= 8 * (16384 + 256) * 16
    assert L.g2s_raster_workspace_bytes(0, 1, 1, 1) == 0


def test_argument_validation_without_gpu():
    """Validation happens before any launch, so it can be exercised on the CPU box."""
    L = lib.load()
    rc = L.g2s_fused_bias_act(None, None, None, None, 16, 1, 1, 3, 0, 0.2, 1.0, 0, None)
    assert rc == -1 and b"NULL" in L.g2s_last_error()
    rc = L.g2s_upfirdn2d(None, None, None, 1, 4, 4, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, 0, None)
    assert rc == -1
    rc = L.g2s_modconv(None, None, None, None, None, 1, 1, 1, 4, 4, 3, 0, 0, None)
    assert rc == -1
    K = (lib.C.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
    rc = L.g2s_raster_depth_fwd(None, None, 1, 16, 18, 4, K, 4.0, 2, 1, 0.1, 100.0, None, None, None,
                                None, 0, None)
    assert rc == -1
    with pytest.raises(lib.G2SError):
        lib.check(rc)


def test_gpu_path_has_no_fallback_cpu_tensors_take_the_reference_split(golden, monkeypatch):
    """The reference's `op` package answers CPU tensors with plain PyTorch and CUDA tensors with the
    native module (op/fused_act.py:87, op/upfirdn2d.py:145).  Same split here: CPU tensors reproduce
    the reference goldens; the native plugins refuse CPU tensors like the reference's CHECK_CUDA; and
    nothing stands in for a missing libg2s.so."""
    from gan2shape_amd.op import fused_leaky_relu, upfirdn2d
    from gan2shape_amd.plugins import fused, upfirdn2d_op
    g = golden("ops")
    for name in ("4d", "2d"):
        x = T(g[f"fused.{name}.x"]).requires_grad_(True)
        b = T(g[f"fused.{name}.b"]).requires_grad_(True)
        y = fused_leaky_relu(x, b)
        gx, gb = torch.autograd.grad(y, (x, b), T(g[f"fused.{name}.gy"]))
        np.testing.assert_allclose(y.detach().numpy(), g[f"fused.{name}.y"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(gx.numpy(), g[f"fused.{name}.gx"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(gb.numpy(), g[f"fused.{name}.gb"], rtol=1e-5, atol=1e-6)
    for name in ("blur_up", "rgb_up", "d_blur3", "d_blur1", "down2", "crop"):
        up, down, p0, p1 = (int(v) for v in g[f"upfirdn2d.{name}.args"])
        x = T(g[f"upfirdn2d.{name}.x"]).requires_grad_(True)
        y = upfirdn2d(x, T(g[f"upfirdn2d.{name}.k"]), up=up, down=down, pad=(p0, p1))
        (gx,) = torch.autograd.grad(y, x, T(g[f"upfirdn2d.{name}.gy"]))
        np.testing.assert_allclose(y.detach().numpy(), g[f"upfirdn2d.{name}.y"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(gx.numpy(), g[f"upfirdn2d.{name}.gx"], rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):      # the native modules themselves: CUDA tensors only
        fused.fused_bias_act(torch.randn(2, 3), torch.zeros(3), torch.empty(0), 3, 0, 0.2, 1.0)
    with pytest.raises(RuntimeError):
        upfirdn2d_op.upfirdn2d(torch.randn(1, 4, 4, 1), torch.ones(4, 4), 1, 1, 1, 1, 0, 0, 0, 0)
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", os.path.join(ROOT, "no_such_dir", "libg2s.so"))
    with pytest.raises(lib.G2SError):      # a missing library raises: nothing stands in for it
        lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gan-2d-to-3d_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                for pat in ("import oracle", "from oracle", "libg2s_oracle", "oracle.capi", "#include \"../../oracle"):
                    assert pat not in src, (f, pat)


# ----------------------------------------------------------------------------- geometry mirror
def test_renderer_utils_golden(golden):
    g = golden("geometry")
    np.testing.assert_array_equal(ru.get_face_idx(2, 4, 4).numpy(), g["face_idx_4x4"])
    np.testing.assert_array_equal(ru.get_face_idx(1, 3, 5).numpy(), g["face_idx_3x5"])
    assert ru.get_face_idx(1, 4, 4).dtype == torch.int32
    np.testing.assert_allclose(ru.get_grid(2, 3, 4, True).numpy(), g["grid_norm"], atol=1e-7)
    np.testing.assert_array_equal(ru.get_grid(1, 3, 4, False).numpy(), g["grid_px"])
    for n in (3, 5, 6):
        r, t = ru.get_transform_matrices(T(g["view6"][:, :n]))
        np.testing.assert_allclose(r.numpy(), g[f"rot{n}"], atol=1e-6)
        np.testing.assert_array_equal(t.numpy(), g[f"trans{n}"])
    with pytest.raises(Exception):
        ru.get_transform_matrices(torch.zeros(1, 4))
    # texture-cube helpers (renderer/utils.py:83-109)
    im = T(g["tex.im"])
    np.testing.assert_array_equal(ru.get_textures_from_im(im, 1).numpy(), g["tex.size1"])
    np.testing.assert_allclose(ru.get_textures_from_im(im, 2).numpy(), g["tex.size2"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ru.vcolor_to_texture_cube(T(g["tex.vcolors"])).numpy(), g["tex.cube"], rtol=1e-6, atol=1e-7)
    with pytest.raises(NotImplementedError):
        ru.get_textures_from_im(im, 3)


def test_renderer_geometry_golden(golden):
    g = golden("geometry")
    S = g["r.depth"].shape[1]
    R = Renderer({"rot_center_depth": 1.0, "fov": 10}, S, 0.9, 1.1, device="cpu")
    np.testing.assert_allclose(R.K.numpy(), g["r.K"], rtol=1e-6)
    np.testing.assert_allclose(R.inv_K.numpy(), g["r.inv_K"], rtol=1e-5, atol=1e-7)
    R.set_transform_matrices(T(g["r.view"]))
    d = T(g["r.depth"])
    np.testing.assert_allclose(R.depth_to_3d_grid(d).numpy(), g["r.grid3d"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(R.get_warped_3d_grid(d).numpy(), g["r.warped3d"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(R.get_inv_warped_3d_grid(d).numpy(), g["r.invwarped3d"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(R.get_inv_warped_2d_grid(d).numpy(), g["r.invwarped2d"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(R.get_normal_from_depth(d).numpy(), g["r.normal"], rtol=1e-4, atol=1e-5)


def test_resize_golden(golden):
    g = golden("misc")
    x = T(g["resize.x"])
    np.testing.assert_allclose(utils.resize(x, [16, 16]).numpy(), g["resize.up"], atol=1e-6)
    np.testing.assert_allclose(utils.resize(x, [4, 4]).numpy(), g["resize.down"], atol=1e-6)
    assert utils.resize(x, [8, 8]) is x
    np.testing.assert_allclose(utils.resize(T(g["resize.x3"]), [4, 4]).numpy(), g["resize.down3"], atol=1e-6)


# ----------------------------------------------------------------------------- losses / priors
def test_losses_closed_form():
    a = torch.tensor([[[[1., 2.], [3., 4.]]]])
    b = torch.zeros_like(a)
    m = torch.tensor([[[[1., 0.], [0., 1.]]]])
    assert losses.PhotometricLoss()(a, b).item() == 2.5
    assert losses.PhotometricLoss()(a, b, mask=m).item() == 2.5
    assert losses.PhotometricLoss()(a, b, mask=torch.tensor([[[[1., 0.], [0., 0.]]]])).item() == 1.0
    # second differences of a quadratic ramp: d2/dx2 (x^2) = 2, everything else 0
    x = torch.arange(6.).view(1, 1, 1, 6).expand(1, 1, 6, 6) ** 2
    assert abs(losses.SmoothLoss()(x).item() - 2.0) < 1e-6
    assert losses.SmoothLoss()(x[0]).item() == losses.SmoothLoss()([x]).item()

    class FakeD:
        def __call__(self, img, ftr_num):
            return 0, [img * 2, img[:, :, ::2, ::2]][:ftr_num]
    fake, real = torch.ones(1, 1, 4, 4), torch.zeros(1, 1, 4, 4)
    mask = torch.ones(1, 1, 4, 4)
    mask[..., 2:] = 0
    dl = losses.DiscriminatorLoss(ftr_num=2)
    assert abs(dl(FakeD(), fake, real).item() - 3.0) < 1e-6
    assert abs(dl(FakeD(), fake, real, mask=mask).item() - 3.0) < 1e-6


def test_priors_from_synthetic_mask():
    img = torch.zeros(1, 3, 64, 64)
    for name in ["ellipsoid", "smoothed_box", "masked_box", "box", "confidence", "smoothed_confidence"]:
        p = priors.PriorGenerator(64, "face", name)(img, device="cpu")
        assert p.shape == (1, 64, 64) and torch.isfinite(p).all()
    e = priors.PriorGenerator(64, "face", "ellipsoid")(img, device="cpu")
    assert abs(e.max().item() - 1.02) < 1e-6 and 0.90 < e.min().item() < 0.93   # near = 0.91
    assert e[0, 32, 32] < e[0, 32, 12]                                            # dome
    s = priors.PriorGenerator(64, "face", "smoothed_box")(img, device="cpu")
    assert abs(s.max().item() - 1.02) < 1e-5 and abs(s.min().item() - 0.91) < 1e-5


# ----------------------------------------------------------------------------- model-level math (reference fixtures)
def test_model_math_golden(golden):
    """rescale / clamp / view / lighting / shading / prior forward of GAN2Shape/model.py:85-93,
    330-360 — this package's torch path against outputs of the reference's own methods
    (tests/golden/model.npz)."""
    from model_cases import bare_model
    g = golden("model")
    m = bare_model("cpu")
    raw = T(g["m.depth_raw"])
    S = raw.shape[-1]
    np.testing.assert_allclose(m.rescale_depth(torch.tanh(raw)).numpy(), g["m.rescale"], rtol=1e-6)
    np.testing.assert_allclose(m.get_clamped_depth(raw, S, S).numpy(), g["m.clamped"], rtol=1e-6)
    np.testing.assert_allclose(m.get_clamped_depth(raw, S, S, clamp_border=False).numpy(),
                               g["m.clamped_noborder"], rtol=1e-6)
    np.testing.assert_allclose(m.get_view_transformation(T(g["m.view"])).numpy(), g["m.view_trans"], rtol=1e-6)
    la, lb, ld = m.get_lighting_directions(T(g["m.light"]))
    for got, key in ((la, "m.light_a"), (lb, "m.light_b"), (ld, "m.light_d")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-6, atol=1e-7)
    diffuse, texture = m.get_shading(T(g["m.normal"]), la, lb, ld, T(g["m.albedo"]))
    np.testing.assert_allclose(diffuse.numpy(), g["m.diffuse"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(texture.numpy(), g["m.texture"], rtol=1e-6, atol=1e-6)
    a, b, d2, t2 = m._shade(T(g["m.normal"]), T(g["m.light"]), T(g["m.albedo"]))   # the step's entry point
    np.testing.assert_allclose(t2.numpy(), g["m.texture"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(a.numpy(), g["m.light_a"], rtol=1e-6)
    loss, depth = m.depth_net_forward(T(g["m.dnf.x"]), T(g["m.dnf.prior"]))
    assert depth.shape == g["m.dnf.depth"].shape                    # the 4-D (1,B,H,W) broadcast
    np.testing.assert_allclose(depth.numpy(), g["m.dnf.depth"], rtol=1e-6)
    np.testing.assert_allclose(loss.item(), g["m.dnf.loss"], rtol=1e-6)


def test_view_light_sampler_golden(golden):
    """ViewLightSampler (model.py:448-470): n sequential MultivariateNormal draws, view_scale on
    column 1 — a seeded CPU run reproduces the reference's samples exactly."""
    from gan2shape_amd.model import ViewLightSampler
    g = golden("model")
    vls = ViewLightSampler(None, None, 0.5, "cpu",
                           {"mean": g["vls.vm"].tolist(), "cov": g["vls.vc"].tolist()},
                           {"mean": g["vls.lm"].tolist(), "cov": g["vls.lc"].tolist()})
    torch.manual_seed(0)
    np.testing.assert_allclose(vls.sample(5, "view").numpy(), g["vls.views"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(vls.sample(3, "light").numpy(), g["vls.lights"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(vls.sample(2).numpy(), g["vls.views2"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(vls.view_mean.numpy(), g["vls.vm"])


def test_losses_golden(golden):
    """The three loss classes against outputs of the reference's losses.py:6-79 (fixed fake D from
    tests/model_cases.py)."""
    from model_cases import FakeD
    g = golden("model")
    a, b, mask, sigma = (T(g[k]) for k in ("l.a", "l.b", "l.mask", "l.sigma"))
    P, Sm = losses.PhotometricLoss(), losses.SmoothLoss()
    np.testing.assert_allclose(P(a, b).item(), g["l.photo"], rtol=1e-6)
    np.testing.assert_allclose(P(a, b, mask=mask).item(), g["l.photo_mask"], rtol=1e-6)
    np.testing.assert_allclose(P(a, b, mask=mask, conf_sigma=sigma).item(), g["l.photo_sigma"], rtol=1e-6)
    np.testing.assert_allclose(Sm(a[:, 0]).item(), g["l.smooth3"], rtol=1e-6)
    np.testing.assert_allclose(Sm(a).item(), g["l.smooth4"], rtol=1e-6)
    np.testing.assert_allclose(Sm([a, b[:, :, ::2, ::2]]).item(), g["l.smooth_pyr"], rtol=1e-6)
    ar = a.clone().requires_grad_(True)
    (gr,) = torch.autograd.grad(Sm(ar[:, 0]), ar)
    np.testing.assert_allclose(gr.numpy(), g["l.smooth3_grad"], rtol=1e-5, atol=1e-8)
    D, DL = FakeD(), losses.DiscriminatorLoss()
    for name, m_ in (("", None), ("_mask", mask)):
        fake = a.clone().requires_grad_(True)
        val = DL(D, fake, b, mask=m_)
        (gr,) = torch.autograd.grad(val, fake)
        np.testing.assert_allclose(val.item(), g[f"l.dloss{name}"], rtol=1e-6)
        np.testing.assert_allclose(gr.numpy(), g[f"l.dloss{name}_grad"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(losses.DiscriminatorLoss(ftr_num=2)(D, a, b, mask=mask).item(),
                               g["l.dloss_ftr2"], rtol=1e-6)


def test_priors_golden(golden):
    """The six priors against the reference's PriorGenerator (priors.py:26-107) fed the same
    synthetic parsing mask."""
    from model_cases import PRIOR_NAMES, FakeMaskingModel
    g = golden("model")
    for size in (64, 128):
        fm = FakeMaskingModel(size)
        img = torch.zeros(1, 3, size, size)
        for name in PRIOR_NAMES:
            key = f"p.{name}.{size}"
            if key not in g:
                continue
            source = fm.confidence_mask if "confidence" in name else fm
            p = priors.PriorGenerator(size, "face", name, masking_model=source)(img, device="cpu")
            np.testing.assert_allclose(p.numpy(), g[key], rtol=2e-6, atol=1e-6, err_msg=key)


# ----------------------------------------------------------------------------- trainer loop (reference fixture)
def test_trainer_fit_follows_reference_loop(golden):
    """Trainer.fit against the reference's own Trainer.fit (GAN2Shape/trainer.py:57-128,130-171)
    driving the same toy model on the CPU (tests/golden/trainer.npz): identical order of calls
    (prior pre-training, stages x steps x iterations, the `collected` hand-off), identical losses
    and final parameters — three persistent Adams, a fresh one per image for the prior."""
    from gan2shape_amd.trainer import Trainer
    from model_cases import TOY_CFG, TOY_STAGES, ToyStepModel, toy_dataset
    g = golden("trainer")

    class FixedPrior:
        def __call__(self, image, device="cpu"):
            return torch.full((1, 8, 8), 0.97)
    t = Trainer(ToyStepModel, dict(TOY_CFG), device="cpu")
    t.prior_generator = FixedPrior()
    n = t.fit(toy_dataset(), stages=TOY_STAGES)
    log = np.array(t.model.log, np.float64)
    assert n == int((g["log"][:, 0] > 0).sum())
    np.testing.assert_array_equal(log[:, 0], g["log"][:, 0])          # call order
    np.testing.assert_allclose(log[:, 1], g["log"][:, 1], rtol=1e-6)   # every loss
    params = torch.cat([p.reshape(-1) for p in t.model.parameters()]).detach().numpy()
    np.testing.assert_allclose(params, g["params"], rtol=1e-6, atol=1e-7)


def test_joint_trainer_fit_follows_reference_run(golden):
    """GeneralizingTrainer2.fit (BASELINE config 4's loop) against the reference's own
    GeneralizingTrainer2.fit run (GAN2Shape/trainer.py:338-479, pretrain_on_prior :296-335) on the toy
    model, 5 images in batches of 2 (ragged last batch), 2 epochs: order and batch size of every
    call, the hand-off each step received, every loss, the final parameters."""
    from gan2shape_amd.trainer import GeneralizingTrainer2
    from model_cases import TOY_JOINT_CFG, TOY_JOINT_STAGES, ToyJointModel, toy_dataset, toy_prior
    g = golden("trainer")

    class ImagePrior:
        def __call__(self, image, device="cpu"):
            return toy_prior(image)
    t = GeneralizingTrainer2(ToyJointModel, dict(TOY_JOINT_CFG), device="cpu")
    t.prior_generator = ImagePrior()
    n = t.fit(toy_dataset(5), stages=TOY_JOINT_STAGES, batch_size=2)
    log, ref = np.array(t.model.log, np.float64), g["joint.log"]
    assert log.shape == ref.shape
    assert n == int((ref[:, 0] > 0).sum())
    np.testing.assert_array_equal(log[:, 0], ref[:, 0])               # call order
    np.testing.assert_array_equal(log[:, 2], ref[:, 2])               # batch size of every call
    np.testing.assert_allclose(log[:, 3], ref[:, 3], rtol=1e-5)       # the `collected` each step received
    np.testing.assert_allclose(log[:, 1], ref[:, 1], rtol=1e-5)       # every loss
    params = torch.cat([p.reshape(-1) for p in t.model.parameters()]).detach().numpy()
    np.testing.assert_allclose(params, g["joint.params"], rtol=1e-5, atol=1e-6)


# ----------------------------------------------------------------------------- reference (live)
def _ref_sg2():
    if SG2 not in sys.path:
        sys.path.insert(0, SG2)
    import model as ref_sg2
    return ref_sg2


@needs_ref
def test_small_nets_equal_reference():
    sys.path.insert(0, REF)
    from GAN2Shape import networks as ref
    torch.manual_seed(0)
    x = torch.randn(2, 3, 128, 128)
    for name in ["DepthNet", "AlbedoNet", "ViewpointNet", "LightingNet", "OffsetEncoder"]:
        r = getattr(ref, name)(128)
        m = getattr(networks, name)(128)
        m.load_state_dict(r.state_dict(), strict=True)
        with torch.no_grad():
            torch.testing.assert_close(m(x), r(x), rtol=0, atol=0)
    m64 = networks.OffsetEncoder(64)
    assert m64(torch.randn(1, 3, 64, 64)).shape == (1, 512)


@needs_ref
@pytest.mark.parametrize("size,cm", [(128, 1), (64, 2)])
def test_stylegan2_state_dict_compatible(size, cm):
    ref = _ref_sg2()
    from gan2shape_amd import stylegan2 as sg2
    for cls, args in (("Generator", (size, 512, 8)), ("Discriminator", (size,))):
        r = getattr(ref, cls)(*args, channel_multiplier=cm)
        m = getattr(sg2, cls)(*args, channel_multiplier=cm)
        rs, ms = r.state_dict(), m.state_dict()
        assert list(rs.keys()) == list(ms.keys())
        assert all(rs[k].shape == ms[k].shape for k in rs)
        m.load_state_dict(rs, strict=True)


@needs_ref
def test_lpips_state_dict_keys_match_reference_weight_file():
    from gan2shape_amd.lpips import PerceptualLoss
    path = os.path.join(SG2, "lpips/weights/v0.1/vgg.pth")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    p = PerceptualLoss()
    own = p.net.state_dict()
    assert set(sd) <= set(own) and all(sd[k].shape == own[k].shape for k in sd)
    p.load_lin_weights(path)
    torch.testing.assert_close(p.net.lin3.model[1].weight, sd["lin3.model.1.weight"])
    # the metric itself (plain torch, runs on CPU): identical inputs -> 0, symmetric
    torch.manual_seed(0)
    a, b = torch.rand(1, 3, 32, 32) * 2 - 1, torch.rand(1, 3, 32, 32) * 2 - 1
    assert p(a, a).abs().max() == 0
    torch.testing.assert_close(p(a, b), p(b, a))
    assert p(a, b).shape == (1, 1, 1, 1) and p(a, b).item() > 0


def seeded_lpips(golden_lpips, device="cpu"):
    """This package's PerceptualLoss with the fixture's trunk (tests/model_cases.vgg16_features) and
    the reference's five `lin` layers as stored in tests/golden/lpips.npz."""
    import model_cases as mc
    from gan2shape_amd.lpips import PerceptualLoss
    p = PerceptualLoss()
    p.load_vgg_features({f"features.{k}": v for k, v in mc.vgg16_features(mc.LPIPS_CFG["vgg_seed"]).state_dict().items()})
    p.load_lin_weights({f"lin{k}.model.1.weight": torch.from_numpy(golden_lpips[f"lin{k}"]) for k in range(5)})
    return p.to(device).eval()


@pytest.mark.parametrize("case", ["b1_128", "b9_64"])
def test_lpips_vs_reference_run(golden, case):
    """PerceptualLoss (host form, CPU) against the reference's own PerceptualLoss('net-lin', 'vgg')
    run (lpips/__init__.py:12-39, networks_basic.py:27-110, pretrained_networks.py:97-135) on the
    seeded trunk: value, per-layer terms and the gradient w.r.t. the prediction.  torchvision's
    pretrained VGG16 weights themselves stay unpinned (absent offline)."""
    import model_cases as mc
    g = golden("lpips")
    p = seeded_lpips(g)
    B, S, seed = mc.LPIPS_CFG["cases"][case]
    pred, target = mc.lpips_inputs(B, S, seed)
    pred.requires_grad_(True)
    val = p(pred, target)
    assert val.shape == (B, 1, 1, 1)
    np.testing.assert_allclose(val.detach().numpy(), g[f"{case}.val"], rtol=2e-5)
    cot = torch.linspace(0.5, 1.5, B).view(B, 1, 1, 1)
    (gp,) = torch.autograd.grad((val * cot).sum(), pred)
    ref = g[f"{case}.gpred64"].astype(np.float64)
    err = np.linalg.norm(gp.numpy() - ref) / np.linalg.norm(ref)
    # the reference's own fp32 run is ref_fp32_err[1] away from its float64 run (ReLU / max-pool ties)
    assert err <= max(3 * g[f"{case}.ref_fp32_err"][1], 2e-5), err


# ----------------------------------------------------------------------------- sharding
def test_shard_indices_partition():
    from gan2shape_amd.trainer import shard_indices
    for n, w in [(8, 1), (8, 2), (32, 4), (5, 8), (7, 3)]:
        shards = [shard_indices(n, r, w) for r in range(w)]
        assert sorted(i for s in shards for i in s) == list(range(n))
        assert shards[0][:2] == [0, w][:len(shards[0][:2])]


# ----------------------------------------------------------------------------- dataset format
def _make_dataset(root):
    from PIL import Image
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(root, "latents"))
    names = []
    for i, (w, h) in enumerate([(40, 40), (48, 36), (30, 50)]):
        name = f"img_{i}.png"
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, name))
        names.append(name)
        lat = torch.randn(1, 512) if i != 1 else torch.randn(10, 512)
        obj = lat if i == 0 else ({"latent": lat} if i == 1 else {name: {"latent": lat}})
        torch.save(obj, os.path.join(root, "latents", f"img_{i}.pt"))
    with open(os.path.join(root, "list.txt"), "w") as f:
        f.write("\n".join(names) + "\n")


def test_dataset_formats(tmp_path):
    """list.txt + latents/*.pt layout of GAN2Shape/dataset.py:8-79 (tensor, {'latent': w} and
    {name: {'latent': w}} latent files; subsets; [-1, 1] images)."""
    from gan2shape_amd import dataset
    root = str(tmp_path)
    _make_dataset(root)
    ds = dataset.ImageLatentDataset(root, transform=dataset.default_transform(32))
    assert len(ds) == 3
    shapes = [(3, 32, 32), (3, 32, 42), (3, 53, 32)]
    for i in range(3):
        image, latent, index = ds[i]
        assert index == i and tuple(image.shape) == shapes[i]
        assert image.dtype == torch.float32 and -1 <= float(image.min()) and float(image.max()) <= 1
        assert tuple(latent.shape) == ((512,) if i != 1 else (10, 512))
    sub = dataset.ImageLatentDataset(root, transform=dataset.default_transform(32), subset=[2, 0])
    assert len(sub) == 2 and torch.equal(sub[0][1], ds[2][1]) and torch.equal(sub[1][0], ds[0][0])
    with pytest.raises(IndexError):
        dataset.ImageDataset(root, subset=[5])


@needs_ref
def test_dataset_equals_reference(tmp_path):
    sys.path.insert(0, REF)
    from GAN2Shape import dataset as ref
    from gan2shape_amd import dataset
    root = str(tmp_path)
    _make_dataset(root)
    tf = dataset.default_transform(32)
    a = dataset.ImageLatentDataset(root, transform=tf, subset=[1, 2, 0])
    b = ref.ImageLatentDataset(root, transform=tf, subset=[1, 2, 0])
    assert len(a) == len(b)
    for i in range(len(a)):
        (ia, la, xa), (ib, lb, xb) = a[i], b[i]
        assert xa == xb and torch.equal(ia, ib) and torch.equal(la, lb)


def test_pass_arena_slices_and_fallbacks():
    """op/conv.py:PassArena — zero-filled slices per pass; oversize / exhausted requests get None
    (the caller then allocates and lets the library clear); the backward arena appears on first use."""
    from gan2shape_amd.op.conv import PassArena
    a = PassArena(torch.device("cpu"), 64 + 128)
    v1 = a.take_fwd((2, 5, 2, 3))           # 60 elements -> padded to 64
    v2 = a.take_fwd((128,))
    assert v1.shape == (2, 5, 2, 3) and v2.shape == (128,) and float(v1.abs().sum() + v2.abs().sum()) == 0
    assert v1.data_ptr() + 64 * 4 == v2.data_ptr()          # distinct, 256-byte aligned slices
    assert a.take_fwd((1,)) is None                          # exhausted
    assert a.take_fwd((PassArena.LIMIT + 1,)) is None        # too large for the arena
    assert a.take_bwd((4,)) is None                          # nothing reserved
    a.reserve_bwd(10)
    a.reserve_bwd(PassArena.LIMIT + 1)                       # ignored: keeps its own allocation
    g = a.take_bwd((10,))
    assert g is not None and a.bwd.numel() == 64 and float(g.abs().sum()) == 0
    assert a.take_bwd((10,)) is None                         # second backward through the graph
    empty = PassArena(torch.device("cpu"), 0)
    assert empty.take_fwd((4,)) is None
