"""LPIPS-VGG perceptual loss, restated from the vendored lpips package of the reference
(GAN2Shape/stylegan2/stylegan2-pytorch/lpips/__init__.py:12-39 PerceptualLoss,
networks_basic.py:27-110 PNetLin / ScalingLayer / NetLinLayer, pretrained_networks.py:97-135 vgg16,
__init__.py:40-42 normalize_tensor).  Plain torch on MIOpen: inside every
step-1/3 iteration, but not a custom-kernel target.

State-dict keys equal the reference's PNetLin (`net.slice{1..5}.<torchvision index>.weight`,
`lin{0..4}.model.1.weight`), so `lpips/weights/v0.1/vgg.pth` (5 tensors) and a torchvision VGG16
`features` state dict load unchanged through `load_lin_weights` / `load_vgg_features`.  torchvision
and its pretrained download are unavailable offline: without those files the network is
random-initialised (benchmarks / tests), and says so via `.pretrained`.
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import lib as _lib
from . import zeropool

# torchvision.models.vgg16().features layout: conv indices and the pools in front of each slice
_VGG_SLICES = [
    [(0, 3, 64), (2, 64, 64)],
    [(5, 64, 128), (7, 128, 128)],
    [(10, 128, 256), (12, 256, 256), (14, 256, 256)],
    [(17, 256, 512), (19, 512, 512), (21, 512, 512)],
    [(24, 512, 512), (26, 512, 512), (28, 512, 512)],
]
_POOL_BEFORE = [None, 4, 9, 16, 23]


def normalize_tensor(in_feat, eps=1e-10):
    norm_factor = torch.sqrt(torch.sum(in_feat ** 2, dim=1, keepdim=True))
    return in_feat / (norm_factor + eps)


class _LpipsLayer(Function):
    """out[n] += mean_hw sum_c w_c (normalize(f0) - normalize(f1))^2 in one kernel (g2s_lpips_layer_*).
    Gradient w.r.t. f0 only."""

    @staticmethod
    def forward(ctx, f0, f1, w, acc):
        f0, f1 = f0.contiguous(), f1.contiguous()
        N, C, H, W = f0.shape
        wv = w.reshape(-1).contiguous()
        L = _lib.load()
        _lib.check(L.g2s_lpips_layer_fwd(_lib.ptr(f0), _lib.ptr(f1), _lib.ptr(wv), _lib.ptr(acc),
                                         N, C, H * W, _lib.stream()))
        ctx.save_for_backward(f0, f1, wv)
        ctx.mark_dirty(acc)
        return acc

    @staticmethod
    def backward(ctx, gout):
        f0, f1, wv = ctx.saved_tensors
        N, C, H, W = f0.shape
        g0 = torch.empty_like(f0)
        L = _lib.load()
        _lib.check(L.g2s_lpips_layer_bwd(_lib.ptr(f0), _lib.ptr(f1), _lib.ptr(wv),
                                         _lib.ptr(gout.contiguous()), _lib.ptr(g0), N, C, H * W,
                                         _lib.stream()))
        return g0, None, None, gout


class _MaxPool2x2(Function):
    """nn.MaxPool2d(2, 2) on g2s_maxpool2x2_*: the backward routes each output gradient to the first
    maximum of its window (torch's rule) straight from the saved input — no cleared gradient buffer,
    no index tensor."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C_, H, W = x.shape
        y = torch.empty((B, C_, H // 2, W // 2), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().g2s_maxpool2x2_fwd(_lib.ptr(x), _lib.ptr(y), B * C_, H, W, _lib.stream()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        B, C_, H, W = x.shape
        gx = torch.empty_like(x)
        _lib.check(_lib.load().g2s_maxpool2x2_bwd(_lib.ptr(x), _lib.ptr(gy.contiguous()), _lib.ptr(gx), B * C_, H, W,
                                                  _lib.stream()))
        return gx


def _pool_ok(mod):
    return (mod.kernel_size in (2, (2, 2)) and mod.stride in (2, (2, 2)) and mod.padding in (0, (0, 0))
            and not mod.ceil_mode and mod.dilation in (1, (1, 1)))


class _VggLpips(Function):
    """The whole metric as ONE autograd node (frozen VGG16 + lin layers, gradient-free target):

    forward   prediction and target go through the trunk TOGETHER as one batch of 2N — 13 convolution
              launches instead of 26, and twice the tiles per launch for the few-tile layers the Winograd
              kernel has to stream-K (B = 9: 72-144 workgroups on 256 CUs) — then the five fused tails
              (g2s_lpips_layer_fwd) on the two halves;
    backward  hand-written over the PREDICTION half only (the first N samples of every saved activation
              are a contiguous view): per slice, g2s_lpips_layer_bwd_ex adds the tail's gradient to the
              one arriving through the next slice's max pool and applies the ReLU gate of the slice's
              last convolution in the same launch; then data-gradient GEMMs with the ReLU gates of the
              inner convolutions between them, the max-pool backward from the saved input.
    Same arithmetic as PNetLin's op-by-op form (lpips/networks_basic.py:64-92); tested against the
    reference's own run (tests/golden/lpips.npz)."""

    SPLIT_FORWARD = False

    @staticmethod
    def forward(ctx, pred, target, net):
        from .modconv import conv_bias_act_raw
        N = pred.shape[0]
        sl = net.scaling_layer
        x = (torch.cat([pred, target], 0) - sl.shift) / sl.scale
        L = _lib.load()
        acc = zeropool.zeros(N, pred.device)
        saved = []
        for si in range(5):
            entry = {"pool_in": None, "convs": []}
            for mod in getattr(net.net, f"slice{si + 1}"):
                if isinstance(mod, nn.MaxPool2d):
                    entry["pool_in"] = x
                    B, C_, H, W = x.shape
                    y = torch.empty((B, C_, H // 2, W // 2), dtype=torch.float32, device=x.device)
                    _lib.check(L.g2s_maxpool2x2_fwd(_lib.ptr(x), _lib.ptr(y), B * C_, H, W, _lib.stream()))
                    x = y
                elif isinstance(mod, nn.Conv2d):
                    if _VggLpips.SPLIT_FORWARD:   # debugging aid: the two halves as separate launches
                        x = torch.cat([conv_bias_act_raw(x[:N], mod.weight, mod.bias),
                                       conv_bias_act_raw(x[N:], mod.weight, mod.bias)], 0)
                    else:
                        x = conv_bias_act_raw(x, mod.weight, mod.bias)
                    entry["convs"].append((mod.weight, x))
            wv = getattr(net, f"lin{si}").model[-1].weight.reshape(-1).contiguous()
            C_, HW = x.shape[1], x.shape[2] * x.shape[3]
            _lib.check(L.g2s_lpips_layer_fwd(_lib.ptr(x[:N]), _lib.ptr(x[N:]), _lib.ptr(wv), _lib.ptr(acc),
                                             N, C_, HW, _lib.stream()))
            entry["lin"] = wv
            saved.append(entry)
        ctx.saved, ctx.N, ctx.scale = saved, N, sl.scale
        return acc.view(-1, 1, 1, 1)

    @staticmethod
    @once_differentiable        # the backward runs raw kernels on saved activations: no double backward
    def backward(ctx, gout):
        from .modconv import PLAIN, modconv_raw, relu_gate
        L = _lib.load()
        N = ctx.N
        gout = gout.contiguous().view(-1)
        g = None
        for si in range(4, -1, -1):
            e = ctx.saved[si]
            y = e["convs"][-1][1]
            C_, HW = y.shape[1], y.shape[2] * y.shape[3]
            gn = torch.empty((N,) + tuple(y.shape[1:]), dtype=torch.float32, device=y.device)
            # (gradient from the next slice + this slice's tail) * (y > 0): the last convolution's ReLU
            _lib.check(L.g2s_lpips_layer_bwd_ex(_lib.ptr(y[:N]), _lib.ptr(y[N:]), _lib.ptr(e["lin"]), _lib.ptr(gout),
                                                _lib.ptr(g), 1, _lib.ptr(gn), N, C_, HW, _lib.stream()))
            g = gn
            convs = e["convs"]
            for ci in range(len(convs) - 1, -1, -1):
                w, yc = convs[ci]
                if ci != len(convs) - 1:
                    g = relu_gate(g, yc[:N])
                g = modconv_raw(g, w, None, None, PLAIN, 1)
            if e["pool_in"] is not None:
                xin = e["pool_in"]
                B2, C2, H, W = xin.shape
                gx = torch.empty((N, C2, H, W), dtype=torch.float32, device=xin.device)
                _lib.check(L.g2s_maxpool2x2_bwd(_lib.ptr(xin[:N]), _lib.ptr(g.contiguous()), _lib.ptr(gx), N * C2, H, W,
                                                _lib.stream()))
                g = gx
        return g / ctx.scale, None, None


def max_pool_2x2(mod, x):
    """`mod(x)` for nn.MaxPool2d(2, 2): the libg2s kernels on CUDA float32 maps with even H and W % 8 == 0."""
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] % 2 == 0 and x.shape[3] % 8 == 0
            and mod.kernel_size in (2, (2, 2)) and mod.stride in (2, (2, 2)) and mod.padding in (0, (0, 0))
            and not mod.ceil_mode and mod.dilation in (1, (1, 1))):
        return _MaxPool2x2.apply(x)
    return mod(x)


class VGG16Features(nn.Module):
    """relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 (pretrained_networks.py:97-135)."""

    def __init__(self):
        super().__init__()
        for si, convs in enumerate(_VGG_SLICES):
            seq = nn.Sequential()
            if _POOL_BEFORE[si] is not None:
                seq.add_module(str(_POOL_BEFORE[si]), nn.MaxPool2d(kernel_size=2, stride=2))
            for idx, cin, cout in convs:
                seq.add_module(str(idx), nn.Conv2d(cin, cout, kernel_size=3, padding=1))
                seq.add_module(str(idx + 1), nn.ReLU(inplace=True))
            setattr(self, f"slice{si + 1}", seq)
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        fused = x.is_cuda and x.dtype == torch.float32 and not any(p.requires_grad for p in self.parameters())
        outs = []
        for si in range(5):
            if fused:  # conv + bias + ReLU in one launch of the MFMA kernel (frozen weights)
                from .modconv import conv_bias_relu
                for mod in getattr(self, f"slice{si + 1}"):
                    if isinstance(mod, nn.Conv2d):
                        x = conv_bias_relu(x, mod.weight, mod.bias)
                    elif isinstance(mod, nn.MaxPool2d):
                        x = max_pool_2x2(mod, x)
            else:
                x = getattr(self, f"slice{si + 1}")(x)
            outs.append(x)
        return outs


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer('shift', torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer('scale', torch.Tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, inp):
        return (inp - self.shift) / self.scale


class NetLinLayer(nn.Module):
    def __init__(self, chn_in, chn_out=1, use_dropout=True):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)


class PNetLin(nn.Module):
    """networks_basic.py:27-92 for pnet_type='vgg', lpips=True, spatial=False, version 0.1."""

    ONE_NODE = True   # the whole metric as one autograd node with a hand-written backward (_VggLpips)

    def __init__(self):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        self.chns = [64, 128, 256, 512, 512]
        self.net = VGG16Features()
        for i, c in enumerate(self.chns):
            setattr(self, f"lin{i}", NetLinLayer(c))
        for p in self.parameters():
            p.requires_grad = False

    def features(self, x):
        return [normalize_tensor(f) for f in self.net(self.scaling_layer(x))]

    def forward(self, in0, in1):
        """in0 = target branch, in1 = prediction branch (PerceptualLoss calls forward(target, pred),
        lpips/__init__.py:39).  On the GPU with a gradient-free target the five per-layer tails run
        as fused kernels; otherwise the reference's op-by-op form is used (CPU tests, or a target
        that needs a gradient)."""
        fused = in0.is_cuda and not in0.requires_grad and not any(
            getattr(self, f"lin{k}").model[-1].weight.requires_grad for k in range(5))
        if (fused and self.ONE_NODE and in0.dtype == torch.float32 and in0.shape == in1.shape
                and in0.shape[2] % 16 == 0 and in0.shape[3] % 64 == 0
                and not any(p.requires_grad for p in self.net.parameters())
                and all(_pool_ok(m) for s_ in range(1, 6) for m in getattr(self.net, f"slice{s_}")
                        if isinstance(m, nn.MaxPool2d))):
            return _VggLpips.apply(in1.contiguous(), in0.contiguous(), self)
        if not fused:
            f0, f1 = self.features(in0), self.features(in1)
            val = 0
            for k in range(5):
                diff = (f0[k] - f1[k]) ** 2
                val = val + getattr(self, f"lin{k}").model(diff).mean([2, 3], keepdim=True)
            return val
        with torch.no_grad():
            ft = self.net(self.scaling_layer(in0))
        fp = self.net(self.scaling_layer(in1))
        # updated in place by _LpipsLayer: its own tensor (an in-place update of a pool slice would bump the
        # version of every other view of the pool)
        acc = torch.zeros(in1.shape[0], dtype=torch.float32, device=in1.device)
        for k in range(5):
            acc = _LpipsLayer.apply(fp[k], ft[k], getattr(self, f"lin{k}").model[-1].weight, acc)
        return acc.view(-1, 1, 1, 1)


class PerceptualLoss(nn.Module):
    """PerceptualLoss(model='net-lin', net='vgg') as constructed at GAN2Shape/model.py:79-81.
    forward(pred, target) -> (N, 1, 1, 1); the reference calls model.forward(target, pred)
    (lpips/__init__.py:39) — the metric is symmetric."""

    def __init__(self, model='net-lin', net='vgg', use_gpu=True, gpu_ids=None,
                 lin_weights_path=None, vgg_weights_path=None):
        super().__init__()
        if model != 'net-lin' or net != 'vgg':
            raise NotImplementedError("only model='net-lin', net='vgg' (GAN2Shape/model.py:79)")
        self.net = PNetLin()
        self.pretrained = False
        if lin_weights_path is not None:
            self.load_lin_weights(lin_weights_path)
        if vgg_weights_path is not None:
            self.load_vgg_features(vgg_weights_path)
            self.pretrained = lin_weights_path is not None
        self.net.eval()

    def load_lin_weights(self, path):
        """`lpips/weights/v0.1/vgg.pth` (or the same five tensors as a dict)."""
        sd = path if isinstance(path, dict) else torch.load(path, map_location="cpu", weights_only=True)
        missing = self.net.load_state_dict(sd, strict=False)
        assert not [k for k in missing.unexpected_keys if k.startswith("lin")], missing

    def load_vgg_features(self, path):
        """A torchvision vgg16 state dict ('features.<idx>.weight') or its `.features` sub-dict."""
        sd = path if isinstance(path, dict) else torch.load(path, map_location="cpu", weights_only=True)
        sd = {k.replace("features.", ""): v for k, v in sd.items() if "classifier" not in k}
        own = {}
        for si, convs in enumerate(_VGG_SLICES):
            for idx, _, _ in convs:
                for suffix in ("weight", "bias"):
                    own[f"slice{si + 1}.{idx}.{suffix}"] = sd[f"{idx}.{suffix}"]
        self.net.net.load_state_dict(own)

    def forward(self, pred, target, normalize=False):
        if normalize:
            target = 2 * target - 1
            pred = 2 * pred - 1
        return self.net(target, pred)
