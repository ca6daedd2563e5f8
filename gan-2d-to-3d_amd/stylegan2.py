"""StyleGAN2 generator / discriminator on the libg2s kernels.

Host-side mirror of GAN2Shape/stylegan2/stylegan2-pytorch/model.py (the rosinality port with
GAN2Shape's additions `style_forward`, `invert`, `invert_sub`, `StridedStyle`, and the
discriminator's `ftr_num` feature tap).  Module and parameter names equal the reference's, so a
reference `state_dict` (checkpoint keys 'g_ema' / 'd', GAN2Shape/model.py:31-35) loads unchanged.

What is different underneath (results equal up to fp32 rounding):
  * ModulatedConv2d (model.py:195-291): instead of building B*Cout*Cin*k*k per-sample weights and
    running a grouped convolution, the shared weights multiply style-scaled activations in one
    fp32-MFMA implicit GEMM (g2s_modconv) with the demodulation factor applied in its epilogue;
  * StyledConv (model.py:321-355): noise injection + bias + leaky-ReLU are one pass
    (g2s_noise_bias_act) when the fixed noise buffers are used (randomize_noise=False — the only way
    GAN2Shape calls the generator, GAN2Shape/model.py:193-195, stylegan2 model.py:519-520);
  * Blur / Upsample / Downsample go through g2s_upfirdn2d, FusedLeakyReLU through
    g2s_fused_bias_act (no PyTorch fallback).
"""
import math
import random

import torch
from torch import nn
from torch.nn import functional as F

from . import synthesis
from .modconv import DOWN2, PLAIN, UP2, conv2d, conv2d_supported, conv_bias_act, modconv, modconv_demod
from .op import FusedLeakyReLU, add_bias_scale, clamp, fused_leaky_relu, fused_noise_bias_act, upfirdn2d


class PixelNorm(nn.Module):
    """model.py:14-19."""

    def forward(self, input):
        return input * torch.rsqrt(torch.mean(input ** 2, dim=1, keepdim=True) + 1e-8)


def make_kernel(k):
    """model.py:22-30 — separable FIR taps, normalised to sum 1."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.dim() == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


class Upsample(nn.Module):
    """model.py:33-51."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    """model.py:54-72."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):
    """model.py:75-91."""

    def __init__(self, kernel, pad, upsample_factor=1, down=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer('kernel', kernel)
        self.pad = pad
        self.down = down  # > 1: only every down-th output is computed (see ConvLayer)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, down=self.down, pad=self.pad)


class _ScaledWeight:
    """weight * scale, recomputed only when the parameter changes (frozen G / D: once)."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, weight, scale):
        if weight.requires_grad and torch.is_grad_enabled():
            return weight * scale
        key = (weight.data_ptr(), weight._version, weight.device)
        if self._key != key:
            self._val = (weight.detach() * scale)
            self._key = key
        return self._val


class EqualConv2d(nn.Module):
    """model.py:94-129.  The discriminator's convolutions run on the same fp32-MFMA kernel as the
    generator's (no modulation): as fast as MIOpen's fp32 Winograd on the stride-1 layers, 20-45 %
    faster on the stride-2 ones, and no NCHW<->NHWC transposes around them."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride = stride
        self.padding = padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None
        self._w = _ScaledWeight()

    def forward(self, input):
        w = self._w.get(self.weight, self.scale)
        mode = conv2d_supported(input, w, self.stride, self.padding)
        if mode is not None and (input.shape[0] * input.shape[2] * input.shape[3] >= 1024):
            out = conv2d(input, w, mode)       # fp32-MFMA implicit GEMM (g2s_modconv, no scales)
            return out if self.bias is None else out + self.bias.view(1, -1, 1, 1)
        return F.conv2d(input, w, bias=self.bias, stride=self.stride, padding=self.padding)

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]},'
                f' {self.weight.shape[2]}, stride={self.stride}, padding={self.padding})')


class EqualLinear(nn.Module):
    """model.py:138-180."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul
        self._w = _ScaledWeight()
        self._b = _ScaledWeight()

    def forward(self, input):
        w = self._w.get(self.weight, self.scale)
        b = None if self.bias is None else self._b.get(self.bias, self.lr_mul)
        if self.activation:
            return fused_leaky_relu(F.linear(input, w), b)
        return F.linear(input, w, bias=b)

    def __repr__(self):
        return f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})'


class ScaledLeakyReLU(nn.Module):
    """model.py:183-192."""

    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, negative_slope=self.negative_slope) * math.sqrt(2)


class PreStyle:
    """A modulation already applied: s = modulation(style) [B, Cin], computed for several layers at
    once by Generator._batched_styles and handed to ModulatedConv2d.forward in place of the style."""

    def __init__(self, s):
        self.s = s


class ModulatedConv2d(nn.Module):
    """model.py:195-291, input-scaling formulation on g2s_modconv."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True,
                 upsample=False, downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1),
                             upsample_factor=factor)
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self._w = _ScaledWeight()
        self._wsq_key = None
        self._wsq = None

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.in_channel}, {self.out_channel}, '
                f'{self.kernel_size}, upsample={self.upsample}, downsample={self.downsample})')

    def _weights(self):
        """(scale*W [Cout,Cin,k,k], sum_taps (scale*W)^2 [Cout,Cin])."""
        w = self._w.get(self.weight, self.scale)[0]
        if not self.demodulate:
            return w, None
        if w.requires_grad:
            return w, w.pow(2).sum((2, 3))
        key = (self.weight.data_ptr(), self.weight._version)
        if self._wsq_key != key:
            self._wsq = w.pow(2).sum((2, 3))
            self._wsq_key = key
        return w, self._wsq

    def forward(self, input, style):
        s = style.s if isinstance(style, PreStyle) else self.modulation(style)  # [B, Cin] (model.py:253)
        w, wsq = self._weights()
        if self.demodulate:                             # model.py:256-258, one node with the convolution
            conv = lambda x, mode: modconv_demod(x, w, s, wsq, self.eps, mode)   # noqa: E731
        else:
            conv = lambda x, mode: modconv(x, w, s, None, mode)                  # noqa: E731
        if self.upsample:                               # model.py:264-275
            return self.blur(conv(input, UP2))
        if self.downsample:                             # model.py:277-283
            return conv(self.blur(input), DOWN2)
        return conv(input, PLAIN)                       # model.py:285-289


class NoiseInjection(nn.Module):
    """model.py:294-305."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):
    """model.py:308-318."""

    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):
    """model.py:321-355."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False,
                 blur_kernel=[1, 3, 3, 1], demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim,
                                    upsample=upsample, blur_kernel=blur_kernel,
                                    demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None):
        out = self.conv(input, style)
        frozen = not (self.noise.weight.requires_grad or self.activate.bias.requires_grad)
        if noise is not None and noise.shape[0] == 1 and (frozen or not torch.is_grad_enabled()):
            return fused_noise_bias_act(out, noise, self.noise.weight, self.activate.bias,
                                        self.activate.negative_slope, self.activate.scale)
        return self.activate(self.noise(out, noise=noise))


class ToRGB(nn.Module):
    """model.py:358-377."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None):
        # conv + bias (+ upsample(skip)) as one pass
        return add_bias_scale(self.conv(input, style), None if skip is None else self.upsample(skip), self.bias)


class _StyleRow:
    """lat[i] of Generator.forward: the latent row, or — in forward order of the layers — the
    pre-computed modulation of the next layer that asks (each layer indexes exactly once)."""

    def __init__(self, lat, pre):
        self.lat, self.pre, self.n = lat, pre, 0

    def __getitem__(self, i):
        if self.pre is None:
            return self.lat[i]
        p = self.pre[self.n]
        self.n += 1
        return p


class NamedTensor(nn.Module):
    """model.py:380-385."""

    def forward(self, x):
        return x


class StridedStyle(nn.ModuleList):
    """model.py:388-396."""

    def __init__(self, n_latents):
        super().__init__([NamedTensor() for _ in range(n_latents)])
        self.n_latents = n_latents

    def forward(self, x):
        return torch.stack([self[i](x[:, i, :]) for i in range(self.n_latents)], dim=1)


class Generator(nn.Module):
    """model.py:398-627."""

    ONE_NODE = True   # frozen generator on the GPU: the synthesis loop as ONE autograd node (synthesis._Synthesis)

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1],
                 lr_mlp=0.01):
        super().__init__()
        self.size = size
        self.style_dim = style_dim
        layers = [PixelNorm()]
        for _ in range(n_mlp):
            layers.append(EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation='fused_lrelu'))
        self.style = nn.ModuleList(layers)
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
                         128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
                         512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim,
                                blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        in_channel = self.channels[4]
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f'noise_{layer_idx}', torch.randn(1, 1, 2 ** res, 2 ** res))
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True,
                                         blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim,
                                         blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2
        self.strided_style = StridedStyle(self.n_latent)

    def _style_layers(self):
        """(ModulatedConv2d, latent index) in forward order (model.py:493-503)."""
        layers = [(self.conv1.conv, 0), (self.to_rgb1.conv, 1)]
        i = 1
        for conv1, conv2, to_rgb in zip(self.convs[::2], self.convs[1::2], self.to_rgbs):
            layers += [(conv1.conv, i), (conv2.conv, i + 1), (to_rgb.conv, i + 2)]
            i += 2
        return layers

    def _batched_styles(self, lat, shared=None):
        """The 17 modulation linears (EqualLinear(style_dim, Cin), model.py:253) as one baddbmm per
        distinct Cin instead of 17 addmm launches forward and 34 backward.  Frozen G on the GPU
        only; returns per-latent-index entries: PreStyle lists are looked up by the caller through
        _StyleRow.  `shared` [B, style_dim]: the one w every layer receives (GAN2Shape's only case,
        model.py:189-190, 207-210) — the batch of inputs is then an expanded view of it (no stack; the
        backward is one sum per group instead of a stack, per-row selects and their accumulation)."""
        layers = self._style_layers()
        mods = [m.modulation for m, _ in layers]
        if shared is None and (not lat[0].is_cuda or not self._styles_frozen()):
            return _StyleRow(lat, None)
        key = tuple((m.weight.data_ptr(), m.weight._version, m.bias._version) for m in mods)
        if getattr(self, '_style_key', None) != key:
            groups = {}
            for n, (m, _) in enumerate(layers):
                groups.setdefault(m.in_channel, []).append(n)
            stacks = []
            with torch.no_grad():
                for cin, ns in groups.items():
                    w = torch.stack([mods[n].weight * mods[n].scale for n in ns]).transpose(1, 2).contiguous()
                    b = torch.stack([mods[n].bias * mods[n].lr_mul for n in ns]).unsqueeze(1).contiguous()
                    stacks.append((ns, w, b))   # w [L, style_dim, Cin], b [L, 1, Cin]
            self._style_stacks, self._style_key = stacks, key
        pre = [None] * len(layers)
        for ns, w, b in self._style_stacks:
            if shared is not None:
                x = shared.unsqueeze(0).expand(len(ns), -1, -1)     # [L, B, style_dim], a view
            else:
                x = torch.stack([lat[layers[n][1]] for n in ns])    # [L, B, style_dim]
            s = torch.baddbmm(b, x, w).unbind(0)                     # L x [B, Cin]; one stack backward
            for j, n in enumerate(ns):
                pre[n] = PreStyle(s[j])
        return _StyleRow(lat, pre)

    def _styles_frozen(self):
        return not any(m.modulation.weight.requires_grad or m.modulation.bias.requires_grad
                       for m, _ in self._style_layers())

    def make_noise(self):
        """One [1,1,r,r] map per styled conv: 4, then two per octave up to `size` (model.py:468-477)."""
        sides = [4] + [2 ** (3 + j // 2) for j in range(self.num_layers - 1)]
        device = self.input.input.device
        return [torch.randn(1, 1, r, r, device=device) for r in sides]

    def mean_latent(self, n_latent):
        z = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style_forward(z).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style_forward(input)

    def style_forward(self, input, skip=0, depth=100):
        """model.py:508-515 — layer 0 is PixelNorm: depth=6 -> PixelNorm + 5 FC, skip=6 -> last 3 FC."""
        out = input
        for i, layer in enumerate(self.style):
            if i >= depth:
                break
            if i >= skip:
                out = layer(out)
        return out

    def invert_sub(self, latent_projection, truncation, mean_latent):
        """model.py:517-521."""
        offset, latent = latent_projection
        gan_im, _ = self([latent], input_is_w=True, truncation_latent=mean_latent,
                         truncation=truncation, randomize_noise=False)
        return clamp(gan_im, -1, 1), offset      # gan_im.clamp(min=-1, max=1): one launch each way

    def invert(self, image, latent_projection, truncation, mean_latent, batchify=0):
        """model.py:523-534 (the reference's batchify branch passes image slices to invert_sub and
        cannot run; only batchify=0 is reachable from GAN2Shape/model.py:207-210)."""
        if batchify > 0:
            raise NotImplementedError("batchify > 0 is broken in the reference (model.py:524-531)")
        return self.invert_sub(latent_projection, truncation, mean_latent)

    def _per_layer_latents(self, styles, inject_index):
        """[B, n_latent, style_dim] from what the caller passed (model.py:568-600): one w (broadcast
        to every layer, or already per layer), two ws (style mixing at `inject_index`), or one w per
        layer.  GAN2Shape only ever passes one (model.py:189-190, 207-210)."""
        n = self.n_latent
        if len(styles) == 1:
            w = styles[0]
            return w if w.dim() == 3 else w[:, None].expand(-1, n, -1).contiguous()
        if len(styles) == 2:
            cut = random.randint(1, n - 1) if inject_index is None else inject_index
            rows = [styles[0]] * cut + [styles[1]] * (n - cut)
        elif len(styles) == n:
            rows = list(styles)
        else:
            raise AssertionError(f'Expected {n} latents, got {len(styles)}')
        return self.strided_style(torch.stack(rows, dim=1))

    def forward(self, styles, return_latents=False, inject_index=None, truncation=1,
                truncation_latent=None, input_is_w=False, noise=None, randomize_noise=False,
                return_features=False):
        if not input_is_w:
            styles = [self.style_forward(s) for s in styles]
        if noise is None:
            if randomize_noise:
                noise = [None] * self.num_layers
            else:
                noise = [getattr(self.noises, f'noise_{i}') for i in range(self.num_layers)]
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        shared = styles[0] if len(styles) == 1 and styles[0].dim() == 2 else None
        if shared is not None and shared.is_cuda and self._styles_frozen():
            latent = None                                  # built only if the caller asks for it
            out = self.input(shared)
            lat = self._batched_styles(None, shared)
        else:
            latent = self._per_layer_latents(styles, inject_index)
            out = self.input(latent)
            # one unbind instead of 2 * n_latent selects: the backward is a single stack, not a
            # zero-fill + copy + add per use (model.py:493-503 indexes latent[:, i] per layer)
            lat = self._batched_styles(latent.unbind(1))
        if (self.ONE_NODE and not return_features and getattr(lat, 'pre', None) is not None
                and synthesis.eligible(self, out, [p.s for p in lat.pre], noise)):
            # the whole layer loop below as one autograd node with fused epilogues (synthesis.py)
            image = synthesis.synthesize(self, noise, out, [p.s for p in lat.pre])
            if return_latents:
                return image, latent if latent is not None else self._per_layer_latents(styles, inject_index)
            return image, None
        out = self.conv1(out, lat[0], noise=noise[0])
        skip = self.to_rgb1(out, lat[1])
        i = 1
        features = []
        for conv1, conv2, noise1, noise2, to_rgb in zip(self.convs[::2], self.convs[1::2],
                                                        noise[1::2], noise[2::2], self.to_rgbs):
            out = conv1(out, lat[i], noise=noise1)
            out = conv2(out, lat[i + 1], noise=noise2)
            skip = to_rgb(out, lat[i + 2], skip)
            features.append(out)
            i += 2
        image = skip
        if return_features:
            return image, features
        if return_latents:
            return image, latent if latent is not None else self._per_layer_latents(styles, inject_index)
        return image, None


class ConvLayer(nn.Sequential):
    """model.py:630-676."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False,
                 blur_kernel=[1, 3, 3, 1], bias=True, activate=True):
        layers = []
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            # A 1x1 stride-2 convolution reads every other blurred pixel only: the blur computes
            # just those (upfirdn2d down=2, a quarter of the work) and the convolution runs at
            # stride 1 on the contiguous result — same values as blur + strided conv (model.py:650-668).
            point = kernel_size == 1
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2), down=2 if point else 1))
            stride = 1 if point else 2
            self.padding = 0
        else:
            stride = 1
            self.padding = kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding,
                                  stride=stride, bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*layers)

    def forward(self, input):
        conv, act = self[-2], self[-1]
        # frozen D on the GPU: convolution + bias + leaky-ReLU * sqrt(2) in one launch of the MFMA
        # kernel instead of conv, then an elementwise pass over the activation
        if (isinstance(act, FusedLeakyReLU) and isinstance(conv, EqualConv2d) and conv.bias is None
                and input.is_cuda and not (torch.is_grad_enabled() and
                                           (conv.weight.requires_grad or act.bias.requires_grad))):
            x = self[0](input) if len(self) == 3 else input
            w = conv._w.get(conv.weight, conv.scale)
            mode = conv2d_supported(x, w, conv.stride, conv.padding)
            if mode is not None and x.shape[0] * x.shape[2] * x.shape[3] >= 1024:
                return conv_bias_act(x, w, act.bias, mode, act.negative_slope, act.scale)
        return super().forward(input)


class ResBlock(nn.Module):
    """model.py:679-697."""

    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def forward(self, input):
        out = self.conv2(self.conv1(input))
        return add_bias_scale(out, self.skip(input), None, 1 / math.sqrt(2))


class Discriminator(nn.Module):
    """model.py:700-769 with the `ftr_num` early-exit feature tap (:741-750)."""

    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
                    128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
                    512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        convs = [ConvLayer(3, channels[size], 1)]
        log_size = int(math.log(size, 2))
        in_channel = channels[size]
        for i in range(log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            convs.append(ResBlock(in_channel, out_channel, blur_kernel))
            in_channel = out_channel
        self.convs = nn.ModuleList(convs)
        self.stddev_group = 4
        self.stddev_feat = 1
        self.final_conv = ConvLayer(in_channel + 1, channels[4], 3)
        self.final_linear = nn.Sequential(
            EqualLinear(channels[4] * 4 * 4, channels[4], activation='fused_lrelu'),
            EqualLinear(channels[4], 1))

    def _group_stddev(self, out):
        """Minibatch-stddev feature (model.py:756-765): samples b, b+G', b+2G', ... form a group; the
        per-element standard deviation over a group, averaged over (C / feat, H, W), is appended as
        `feat` constant maps to every member."""
        B, C, H, W = out.shape
        G, F_ = min(B, self.stddev_group), self.stddev_feat
        members = out.reshape(G, B // G, F_, C // F_, H * W)
        spread = (members.var(0, unbiased=False) + 1e-8).sqrt().mean((2, 3))       # [B/G, feat]
        return spread.repeat(G, 1)[:, :, None, None].expand(B, F_, H, W)

    def forward(self, input, ftr_num=100):
        out = input
        features = []
        for i, block in enumerate(self.convs):
            out = block(out)
            if i > 0:
                features.append(out)
            if len(features) >= ftr_num:
                return 0, features
        out = torch.cat([out, self._group_stddev(out)], 1)
        out = self.final_conv(out)
        features.append(out)
        out = self.final_linear(out.flatten(1))
        return out, features
