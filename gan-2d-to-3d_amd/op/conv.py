"""nn.Conv2d / nn.ConvTranspose2d of the trained nets (GAN2Shape/networks.py:23-244) on libg2s.so:
forward and data-gradient on the fp32-MFMA implicit-GEMM kernel (g2s_conv2d), weight gradient on
the pixel-reduction GEMM (g2s_conv2d_wgrad), optional fused (leaky-)ReLU.  One launch per
direction instead of MIOpen's solver + layout-transpose + workspace-fill sequences at batch 1.
No native fallback."""
import torch
from torch.autograd import Function

from gan2shape_amd import lib as _lib


def _conv2d_raw(x, w, bias, Cr, M, k, stride, pad, adjoint, m_major, out_hw, act, slope):
    B, _, H, W = x.shape
    if adjoint:
        oh, ow = out_hw if out_hw else ((H - 1) * stride - 2 * pad + k, (W - 1) * stride - 2 * pad + k)
    else:
        oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((B, M, oh, ow), dtype=torch.float32, device=x.device)
    L = _lib.load()
    from gan2shape_amd.modconv import profiled
    sp = H * W if adjoint else oh * ow  # every (input pixel, tap) pair of the strided side once
    with profiled(2.0 * B * Cr * M * k * k * sp, 4.0 * (x.numel() + w.numel() + y.numel())):
        _lib.check(L.g2s_conv2d(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), B, Cr, M, H, W, k,
                                stride, pad, int(adjoint), int(m_major), oh if adjoint else 0,
                                ow if adjoint else 0, 1 if act else 0, float(slope), 1.0, _lib.stream()))
    return y


def _wgrad(A, G, k, stride, pad):
    B, Ca, PH, PW = A.shape
    _, Cg, GH, GW = G.shape
    dw = torch.empty((Ca, Cg, k, k), dtype=torch.float32, device=A.device)
    L = _lib.load()
    _lib.check(L.g2s_conv2d_wgrad(_lib.ptr(A), _lib.ptr(G), _lib.ptr(dw), B, Ca, Cg, PH, PW, GH, GW, k,
                                  stride, pad, _lib.stream()))
    return dw


class ConvFunction(Function):
    """transposed = False: F.conv2d(x, w [Cout,Cin,k,k], bias, stride, pad); True:
    F.conv_transpose2d(x, w [Cin,Cout,k,k], bias, stride, pad); then leaky-ReLU(slope) if
    slope is not None (0 = ReLU)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, transposed, slope):
        _lib.require_cuda(x, w, bias)
        if x.dtype != torch.float32 or w.dtype != torch.float32:
            raise RuntimeError("conv: float32 only")
        x, w = x.contiguous(), w.contiguous()
        k = w.shape[2]
        cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        if x.shape[1] != cin:
            raise RuntimeError(f"conv: input has {x.shape[1]} channels, weight expects {cin}")
        b = None if bias is None else bias.contiguous()
        y = _conv2d_raw(x, w, b, cin, cout, k, stride, pad, transposed, not transposed, None,
                        slope is not None, slope or 0.0)
        ctx.save_for_backward(x, w, y if slope is not None else None)
        ctx.cfg = (stride, pad, transposed, slope, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, transposed, slope, has_bias = ctx.cfg
        k = w.shape[2]
        gy = gy.contiguous()
        if slope is not None:  # gradient through the fused activation: slope taken from sign(y)
            from gan2shape_amd.plugins import fused
            gy = fused.fused_bias_act(gy, gy.new_empty(0), y, 3, 1, float(slope), 1.0)
        cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _conv2d_raw(gy, w, None, cout, cin, k, stride, pad, not transposed, transposed,
                             (x.shape[2], x.shape[3]), False, 0.0)
        if ctx.needs_input_grad[1]:
            gw = _wgrad(x, gy, k, stride, pad) if transposed else _wgrad(gy, x, k, stride, pad)
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3))
        return gx, gw, gb, None, None, None, None


def supported(mod, x):
    """True if `mod` (nn.Conv2d / nn.ConvTranspose2d) on input x can run on g2s_conv2d."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    k = mod.kernel_size
    return (k[0] == k[1] and 1 <= k[0] <= 5 and mod.stride[0] == mod.stride[1] and mod.stride[0] in (1, 2)
            and mod.padding[0] == mod.padding[1] and isinstance(mod.padding[0], int) and 0 <= mod.padding[0] < k[0]
            and mod.dilation == (1, 1) and mod.groups == 1 and mod.padding_mode == 'zeros'
            and getattr(mod, 'output_padding', (0, 0)) == (0, 0)
            and x.shape[2] + 2 * mod.padding[0] >= k[0] and x.shape[3] + 2 * mod.padding[0] >= k[0])


def conv_module(mod, x, slope=None):
    """Run nn.Conv2d / nn.ConvTranspose2d `mod` (its parameters, stride, padding) on libg2s.so."""
    transposed = isinstance(mod, torch.nn.ConvTranspose2d)
    return ConvFunction.apply(x, mod.weight, mod.bias, mod.stride[0], mod.padding[0], transposed, slope)
