"""`FusedLeakyReLU`, `fused_leaky_relu` — the Python op API of
GAN2Shape/stylegan2/stylegan2-pytorch/op/fused_act.py:74-92 — and `fused_noise_bias_act`, on the
`fused` plugin (libg2s.so).  CUDA tensors always take the plugin (no fallback: a missing library
raises); CPU tensors get the reference's plain-PyTorch answer (op/cpu_tensors.py), as in the
reference's own `op` package.

Autograd structure (own): the activation's derivative  g -> g * scale * (out > 0 ? 1 : slope)  is
linear in g, so ONE function (`_SlopeGate`) serves as the backward of every fused activation and
as its own backward (double backward comes for free)."""
import torch
from torch import nn
from torch.autograd import Function

from gan2shape_amd import lib as _lib
from gan2shape_amd.plugins import fused

_LRELU = 3  # `act` code of fused_bias_act for leaky-ReLU (fused_bias_act_kernel.cu:37-45)


class _SlopeGate(Function):
    """g * scale * (ref > 0 ? 1 : slope) — fused_bias_act(act=3, grad=1) with `ref` = the forward
    output."""

    @staticmethod
    def forward(ctx, g, ref, slope, scale):
        ctx.save_for_backward(ref)
        ctx.gate = (slope, scale)
        return fused.fused_bias_act(g, g.new_empty(0), ref, _LRELU, 1, slope, scale)

    @staticmethod
    def backward(ctx, gg):
        ref, = ctx.saved_tensors
        return _SlopeGate.apply(gg, ref, *ctx.gate), None, None, None


class _BiasLeakyReLU(Function):
    """scale * leaky_relu(x + bias[c], slope)."""

    @staticmethod
    def forward(ctx, x, bias, slope, scale):
        out = fused.fused_bias_act(x, bias, x.new_empty(0), _LRELU, 0, slope, scale)
        ctx.save_for_backward(out)
        ctx.gate = (slope, scale)
        return out

    @staticmethod
    def backward(ctx, gy):
        out, = ctx.saved_tensors
        gx = _SlopeGate.apply(gy, out, *ctx.gate)
        gb = None
        if ctx.needs_input_grad[1]:  # frozen G / D biases (GAN2Shape never optimises them) skip this
            gb = gx.sum([d for d in range(gx.dim()) if d != 1])
        return gx, gb, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    """sqrt(2) * leaky_relu(input + bias[c], 0.2) by default (fused_act.py:86-92)."""
    if not input.is_cuda:      # the reference's device split (fused_act.py:87)
        from . import cpu_tensors
        return cpu_tensors.fused_leaky_relu(input, bias, negative_slope, scale)
    _lib.require_cuda(input, bias)
    return _BiasLeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    """fused_act.py:74-83: a learnable per-channel bias followed by the scaled leaky-ReLU."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope, self.scale = negative_slope, scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


class _NoiseBiasLeakyReLU(Function):
    """StyledConv tail (stylegan2-pytorch/model.py:349-355) in one pass:
    scale * leaky_relu(x + noise_w * noise + bias[c]).  Gradient only w.r.t. x (noise weight, noise
    map and bias belong to the frozen generator)."""

    @staticmethod
    def forward(ctx, x, noise, noise_w, bias, slope, scale):
        x = x.contiguous()
        B, C, H, W = x.shape
        flat_noise = None if noise is None else noise.contiguous().view(-1)
        if flat_noise is not None and flat_noise.numel() != H * W:
            raise RuntimeError("noise must be one (1,1,H,W) map")
        y = torch.empty_like(x)
        L = _lib.load()
        _lib.check(L.g2s_noise_bias_act(
            _lib.ptr(x), _lib.ptr(flat_noise), _lib.ptr(None if flat_noise is None else noise_w.contiguous()),
            _lib.ptr(bias.contiguous()), _lib.ptr(y), B, C, H * W, float(slope), float(scale), _lib.stream()))
        ctx.save_for_backward(y)
        ctx.gate = (slope, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        return _SlopeGate.apply(gy, y, *ctx.gate), None, None, None, None, None


def fused_noise_bias_act(x, noise, noise_w, bias, negative_slope=0.2, scale=2 ** 0.5):
    _lib.require_cuda(x, bias)
    if x.dtype != torch.float32:
        raise RuntimeError("fused_noise_bias_act: float32 only")
    return _NoiseBiasLeakyReLU.apply(x, noise, noise_w, bias, negative_slope, scale)


class _AddBiasScale(Function):
    """(a + b + bias[c]) * scale in one launch (g2s_add_bias_scale); b / bias may be None."""

    @staticmethod
    def forward(ctx, a, b, bias, scale):
        _lib.require_cuda(a, b, bias)
        a = a.contiguous()
        bc = None if b is None else b.contiguous()
        bv = None if bias is None else bias.reshape(-1).contiguous()
        y = torch.empty_like(a)
        C_, hw = a.shape[1], a.shape[2] * a.shape[3]
        _lib.check(_lib.load().g2s_add_bias_scale(_lib.ptr(a), _lib.ptr(bc), _lib.ptr(bv), _lib.ptr(y), a.numel(), hw,
                                                  C_, float(scale), _lib.stream()))
        ctx.scale = float(scale)
        ctx.bias_shape = None if bias is None else bias.shape
        return y

    @staticmethod
    def backward(ctx, g):
        gs = g if ctx.scale == 1.0 else g * ctx.scale
        gbias = None
        if ctx.bias_shape is not None and ctx.needs_input_grad[2]:
            gbias = gs.sum((0, 2, 3)).reshape(ctx.bias_shape)
        return gs, (gs if ctx.needs_input_grad[1] else None), gbias, None


class _Clamp(Function):
    """torch.clamp(x, lo, hi) with a one-launch backward (g2s_clamp): autograd's ClampBackward runs
    ge + le + logical_and + where (+ a fill) per clamp, and a step has four of them."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.load().g2s_clamp(_lib.ptr(x), None, _lib.ptr(y), x.numel(), float(lo), float(hi), 0, _lib.stream()))
        ctx.save_for_backward(x)
        ctx.bounds = (float(lo), float(hi))
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        gx = torch.empty_like(x)
        _lib.check(_lib.load().g2s_clamp(_lib.ptr(x), _lib.ptr(g.contiguous()), _lib.ptr(gx), x.numel(), *ctx.bounds, 1,
                                         _lib.stream()))
        return gx, None, None


def clamp(x, lo, hi):
    """x.clamp(min=lo, max=hi); CUDA float32 tensors that need a gradient take the libg2s pair."""
    if x.is_cuda and x.dtype == torch.float32 and x.requires_grad and x.numel() > 0:
        return _Clamp.apply(x, lo, hi)
    return x.clamp(min=lo, max=hi)


def add_bias_scale(a, b=None, bias=None, scale=1.0):
    """(a + b + bias) * scale for [B, C, H, W] tensors (bias broadcast over channels, any shape with C
    elements): ToRGB's `conv + bias + upsample(skip)` (stylegan2-pytorch/model.py:371-377) and the
    discriminator ResBlock's `(out + skip) / sqrt(2)` (model.py:693-697) as one pass on the GPU."""
    if a.is_cuda and a.dtype == torch.float32 and a.dim() == 4:
        return _AddBiasScale.apply(a, b, bias, scale)
    out = a if b is None else a + b
    if bias is not None:
        out = out + bias.reshape(1, -1, 1, 1)
    return out if scale == 1.0 else out * scale

