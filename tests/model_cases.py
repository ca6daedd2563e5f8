"""Deterministic stand-ins shared by tests/golden/make_golden.py (which feeds them to the
REFERENCE's classes) and the tests (which feed them to this package's classes): a fixed fake
discriminator, a fixed fake depth net, a synthetic parsing mask.  They are pure functions of their
input, so both sides see identical operands; only the code under test differs."""
import torch
import torch.nn.functional as F


class FakeD:
    """D(img, ftr_num) -> (score, [features]) with the call shape of
    stylegan2-pytorch/model.py:741-750: 4 feature levels at 1/2, 1/4, 1/8, 1/16 resolution."""

    def __call__(self, img, ftr_num=None):
        feats = []
        x = img
        for level in range(4):
            x = F.avg_pool2d(x, 2)
            x = torch.cat([x * (1.5 + level), torch.tanh(x).flip(1) - 0.25 * level], 1)
            feats.append(x)
            if ftr_num is not None and len(feats) >= ftr_num:
                return 0, feats
        return x.mean((1, 2, 3)), feats


def fake_depth_net(x):
    """(B,3,H,W) -> (B,1,H,W): stands in for DepthNet in depth_net_forward fixtures."""
    return (x[:, :1] * 1.7 - x[:, 1:2] * 0.6 + 0.3 * x[:, 2:3] ** 2)


def parsing_mask(size, cx=0.55, cy=0.45, rx=0.3, ry=0.38):
    """Soft off-centre ellipse in [0, 1], (1, 1, S, S): what MaskingModel.image_mask /
    confidence_mask (model.py:473-551) would hand to the priors."""
    yy, xx = torch.meshgrid(torch.linspace(0, 1, size), torch.linspace(0, 1, size), indexing="ij")
    r = torch.sqrt(((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2)
    return (1.25 - r).clamp(0, 1)[None, None]


class FakeMaskingModel:
    """Object with the two methods the reference's PriorGenerator calls (priors.py:37,76,100)."""

    def __init__(self, size):
        self.mask = parsing_mask(size)

    def image_mask(self, image):
        return self.mask.clone()

    def confidence_mask(self, image):
        return self.mask.clone() ** 2

    def __call__(self, image):   # this package's PriorGenerator takes a callable
        return self.mask.clone()


PRIOR_NAMES = ["box", "masked_box", "smoothed_box", "ellipsoid", "confidence", "smoothed_confidence"]


def bare_model(device="cpu", fused=True):
    """This package's GAN2Shape without its networks (the model-level math needs none of them):
    the attributes the pure-math methods read, as GAN2Shape.__init__ sets them (model.py:48-66)."""
    import types
    from gan2shape_amd.model import GAN2Shape
    m = object.__new__(GAN2Shape)
    torch.nn.Module.__init__(m)
    m.device = torch.device(device)
    m.max_depth, m.min_depth = 1.1, 0.9
    m.border_depth = 0.7 * m.max_depth + 0.3 * m.min_depth
    m.xyz_rotation_range, m.xy_translation_range, m.z_translation_range = 60, 0.1, 0.1
    m.batch_mean = None
    m._depth_border = None
    m.depth_net = fake_depth_net
    m.renderer = types.SimpleNamespace(fused=fused)
    return m


def fill_scaled(module, seed, gain=1.0):
    """Seeded, variance-preserving fill of a trained net (sorted state-dict order, like
    make_golden.fill_deterministic): conv / linear weights ~ N(0, gain^2 / fan_in), norm scales
    1 + 0.1 N, biases 0.1 N — keeps the activations of the 7-17-layer nets in range."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for k in sorted(sd.keys()):
            t = sd[k]
            if not t.is_floating_point():
                continue
            v = torch.randn(t.shape, generator=g)
            if t.dim() > 1:
                fan_in = t[0].numel()
                v = v * (gain / fan_in ** 0.5)
            elif k.endswith("weight"):
                v = 1 + 0.1 * v
            else:
                v = 0.1 * v
            t.copy_(v)


def fake_perceptual(pred, target):
    """Stand-in for LPIPS (the reference's lpips package needs torchvision's pretrained VGG16):
    same call shape PerceptualLoss(pred, target) -> (B, 1, 1, 1)."""
    d = F.avg_pool2d(pred, 4) - F.avg_pool2d(target, 4)
    return (d ** 2).mean((1, 2, 3)).view(-1, 1, 1, 1) + (d.abs() ** 1.5).mean((1, 2, 3)).view(-1, 1, 1, 1)


STEP_CFG = dict(image_size=128, gan_size=128, z_dim=512, channel_multiplier=1, n_proj=2,
                seeds=dict(G=41, D=42, lighting=43, viewpoint=44, depth=45, albedo=46, offset_encoder=47))


def prepare_generator(G, seed, fill_deterministic):
    """fill_deterministic leaves the mapping network's weights at N(0, 1) although EqualLinear keeps
    them at N(0, 1) / lr_mul (lr_mul = 0.01): restore that scale so that latent offsets matter."""
    fill_deterministic(G, seed)
    with torch.no_grad():
        for layer in G.style:
            if hasattr(layer, "weight"):
                layer.weight.mul_(100.0)
