"""Mirror of GAN2Shape/stylegan2/stylegan2-pytorch/op/upfirdn2d.py (UpFirDn2d, UpFirDn2dBackward,
upfirdn2d, :18-154) on top of the `upfirdn2d_op` plugin (libg2s.so).  No native fallback."""
import torch
from torch.autograd import Function

from gan2shape_amd import lib as _lib
from gan2shape_amd.plugins import upfirdn2d_op


class UpFirDn2dBackward(Function):
    """upfirdn2d.py:18-84 — the gradient is the same op with up/down swapped, the flipped kernel
    and g_pad."""

    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        up_x, up_y = up
        down_x, down_y = down
        g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1 = g_pad
        grad_output = grad_output.reshape(-1, out_size[0], out_size[1], 1)
        grad_input = upfirdn2d_op.upfirdn2d(grad_output, grad_kernel, down_x, down_y, up_x, up_y,
                                            g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)
        grad_input = grad_input.view(in_size[0], in_size[1], in_size[2], in_size[3])
        ctx.save_for_backward(kernel)
        ctx.up, ctx.down, ctx.pad = up, down, pad
        ctx.in_size, ctx.out_size = in_size, out_size
        return grad_input

    @staticmethod
    def backward(ctx, gradgrad_input):
        kernel, = ctx.saved_tensors
        gradgrad_input = gradgrad_input.reshape(-1, ctx.in_size[2], ctx.in_size[3], 1)
        gradgrad_out = upfirdn2d_op.upfirdn2d(gradgrad_input, kernel, ctx.up[0], ctx.up[1],
                                              ctx.down[0], ctx.down[1], *ctx.pad)
        gradgrad_out = gradgrad_out.view(ctx.in_size[0], ctx.in_size[1], ctx.out_size[0],
                                         ctx.out_size[1])
        return gradgrad_out, None, None, None, None, None, None, None, None


_FLIPPED = {}  # (data_ptr, version, shape) -> flipped FIR kernel: constant buffers, flipped once


def _flipped(kernel):
    key = (kernel.data_ptr(), kernel._version, tuple(kernel.shape), kernel.device)
    hit = _FLIPPED.get(key)
    if hit is None:
        if len(_FLIPPED) > 64:
            _FLIPPED.clear()
        hit = _FLIPPED[key] = (kernel, torch.flip(kernel, [0, 1]))  # holds `kernel`: the ptr stays unique
    return hit[1]


class UpFirDn2d(Function):
    """upfirdn2d.py:87-141."""

    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        pad_x0, pad_x1, pad_y0, pad_y1 = pad
        kernel_h, kernel_w = kernel.shape
        batch, channel, in_h, in_w = input.shape
        ctx.in_size = input.shape
        input = input.reshape(-1, in_h, in_w, 1)
        ctx.save_for_backward(kernel, _flipped(kernel))
        out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h) // down_y + 1
        out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w) // down_x + 1
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = (up_x, up_y), (down_x, down_y), (pad_x0, pad_x1, pad_y0, pad_y1)
        g_pad_x0 = kernel_w - pad_x0 - 1
        g_pad_y0 = kernel_h - pad_y0 - 1
        g_pad_x1 = in_w * up_x - out_w * down_x + pad_x0 - up_x + 1
        g_pad_y1 = in_h * up_y - out_h * down_y + pad_y0 - up_y + 1
        ctx.g_pad = (g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)
        out = upfirdn2d_op.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                     pad_y0, pad_y1)
        return out.view(-1, channel, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        grad_input = UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, ctx.up, ctx.down,
                                             ctx.pad, ctx.g_pad, ctx.in_size, ctx.out_size)
        return grad_input, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d.py:144-154."""
    _lib.require_cuda(input, kernel)
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
