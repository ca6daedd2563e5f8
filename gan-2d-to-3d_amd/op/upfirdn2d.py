"""`upfirdn2d(input, kernel, up, down, pad)` — the Python op API of
GAN2Shape/stylegan2/stylegan2-pytorch/op/upfirdn2d.py:144-154 — on the `upfirdn2d_op` plugin
(libg2s.so).  CUDA tensors always take the plugin (no fallback: a missing library raises); CPU
tensors get the reference's plain-PyTorch answer (op/cpu_tensors.py), as in the reference's `op`.

Autograd structure (own): the adjoint of "upsample by u, pad, FIR with k, downsample by d" is the
same operation with u and d swapped, the flipped kernel and complementary padding, so a single
function (`_Resample`) is its own backward (and double backward)."""
import torch
from torch.autograd import Function

from gan2shape_amd import lib as _lib
from gan2shape_amd.plugins import upfirdn2d_op

_FLIPPED = {}  # (data_ptr, version, shape, device) -> (kernel, flipped kernel): constant FIR buffers


def _flip(kernel):
    key = (kernel.data_ptr(), kernel._version, tuple(kernel.shape), kernel.device)
    hit = _FLIPPED.get(key)
    if hit is None:
        if len(_FLIPPED) > 64:
            _FLIPPED.clear()
        hit = _FLIPPED[key] = (kernel, torch.flip(kernel, [0, 1]))  # holds `kernel`: the ptr stays unique
    return hit[1]


def _out_len(n, up, down, pad0, pad1, taps):
    return (n * up + pad0 + pad1 - taps) // down + 1


class _Resample(Function):
    """x (N, C, H, W) -> upfirdn(x).  rates = (up, down) per axis pair, pad = (x0, x1, y0, y1)."""

    @staticmethod
    def forward(ctx, x, kernel, up, down, pad, out_hw):
        N, C, H, W = x.shape
        planes = x.reshape(N * C, H, W, 1)
        y = upfirdn2d_op.upfirdn2d(planes, kernel, up[0], up[1], down[0], down[1], *pad)
        kh, kw = kernel.shape
        oh = _out_len(H, up[1], down[1], pad[2], pad[3], kh)
        ow = _out_len(W, up[0], down[0], pad[0], pad[1], kw)
        if out_hw is not None and (oh, ow) != tuple(out_hw):
            raise RuntimeError("upfirdn2d adjoint: size mismatch")
        # padding of the adjoint: what is left of the kernel support on either side
        ctx.adjoint_pad = (kw - pad[0] - 1, W * up[0] - ow * down[0] + pad[0] - up[0] + 1,
                           kh - pad[2] - 1, H * up[1] - oh * down[1] + pad[2] - up[1] + 1)
        ctx.rates, ctx.in_hw = (up, down), (H, W)
        ctx.save_for_backward(kernel)
        return y.view(N, C, oh, ow)

    @staticmethod
    def backward(ctx, gy):
        kernel, = ctx.saved_tensors
        up, down = ctx.rates
        gx = _Resample.apply(gy.contiguous(), _flip(kernel), down, up, ctx.adjoint_pad, ctx.in_hw)
        return gx, None, None, None, None, None


def upfirdn2d_adjoint(gy, kernel, up, down, pad, in_hw):
    """Adjoint of y = upfirdn2d(x, kernel, up, down, pad) applied to `gy` (x was in_hw = (H, W)): the
    data-gradient, as one launch (what _Resample.backward computes), for hand-written backward passes."""
    H, W = in_hw
    kh, kw = kernel.shape
    oh, ow = gy.shape[2], gy.shape[3]
    apad = (kw - pad[0] - 1, W * up - ow * down + pad[0] - up + 1,
            kh - pad[0] - 1, H * up - oh * down + pad[0] - up + 1)
    return _Resample.apply(gy.contiguous(), _flip(kernel), (down, down), (up, up), apad, (H, W))


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """Upsample by `up` (zero insertion), pad by `pad` = (before, after) on both axes (negative =
    crop), filter with the 2-D FIR `kernel`, keep every `down`-th sample."""
    if not input.is_cuda:      # the reference's device split (upfirdn2d.py:145)
        from . import cpu_tensors
        return cpu_tensors.upfirdn2d(input, kernel, up, down, pad)
    _lib.require_cuda(input, kernel)
    return _Resample.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]), None)
