"""Mirror of GAN2Shape/stylegan2/stylegan2-pytorch/op/fused_act.py (FusedLeakyReLU,
fused_leaky_relu, :20-92) on top of the `fused` plugin (libg2s.so).  Unlike the reference there
is no native-PyTorch fallback: CPU tensors raise."""
import torch
from torch import nn
from torch.autograd import Function

from gan2shape_amd import lib as _lib
from gan2shape_amd.plugins import fused


class FusedLeakyReLUFunctionBackward(Function):
    """fused_act.py:20-49."""

    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale, need_bias_grad):
        ctx.save_for_backward(out)
        ctx.negative_slope = negative_slope
        ctx.scale = scale
        empty = grad_output.new_empty(0)
        grad_input = fused.fused_bias_act(grad_output, empty, out, 3, 1, negative_slope, scale)
        if need_bias_grad:
            dim = [0]
            if grad_input.dim() > 2:
                dim += list(range(2, grad_input.dim()))
            grad_bias = grad_input.sum(dim).detach()
        else:  # frozen G/D biases (GAN2Shape never optimises them): skip the reduction
            grad_bias = grad_output.new_empty(0)
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        gradgrad_out = fused.fused_bias_act(gradgrad_input, gradgrad_bias, out, 3, 1,
                                            ctx.negative_slope, ctx.scale)
        return gradgrad_out, None, None, None, None


class FusedLeakyReLUFunction(Function):
    """fused_act.py:52-71."""

    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        empty = input.new_empty(0)
        out = fused.fused_bias_act(input, bias, empty, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope = negative_slope
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        need_b = ctx.needs_input_grad[1]
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(
            grad_output, out, ctx.negative_slope, ctx.scale, need_b)
        return grad_input, (grad_bias if need_b else None), None, None


class FusedLeakyReLU(nn.Module):
    """fused_act.py:74-83."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    """fused_act.py:86-92 — sqrt(2) * leaky_relu(input + bias[c])."""
    _lib.require_cuda(input, bias)
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)


class _NoiseBiasAct(Function):
    """StyledConv tail (stylegan2-pytorch/model.py:349-355) in one pass:
    lrelu(x + noise_w * noise + bias) * sqrt(2).  Gradient only w.r.t. x (noise weight, noise map
    and bias belong to the frozen generator)."""

    @staticmethod
    def forward(ctx, x, noise, noise_w, bias, negative_slope, scale):
        x = x.contiguous()
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        nz = None if noise is None else noise.contiguous().view(-1)
        if nz is not None and nz.numel() != H * W:
            raise RuntimeError("noise must be one (1,1,H,W) map")
        L = _lib.load()
        _lib.check(L.g2s_noise_bias_act(_lib.ptr(x), _lib.ptr(nz),
                                        _lib.ptr(None if nz is None else noise_w.contiguous()),
                                        _lib.ptr(bias.contiguous()), _lib.ptr(y), B, C, H * W,
                                        float(negative_slope), float(scale), _lib.stream()))
        ctx.save_for_backward(y)
        ctx.negative_slope = negative_slope
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        empty = grad_output.new_empty(0)
        gx = fused.fused_bias_act(grad_output, empty, out, 3, 1, ctx.negative_slope, ctx.scale)
        return gx, None, None, None, None, None


def fused_noise_bias_act(x, noise, noise_w, bias, negative_slope=0.2, scale=2 ** 0.5):
    _lib.require_cuda(x, bias)
    if x.dtype != torch.float32:
        raise RuntimeError("fused_noise_bias_act: float32 only")
    return _NoiseBiasAct.apply(x, noise, noise_w, bias, negative_slope, scale)
