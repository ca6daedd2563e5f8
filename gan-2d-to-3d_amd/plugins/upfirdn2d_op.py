"""Drop-in for the reference's pybind module `upfirdn2d_op`
(GAN2Shape/stylegan2/stylegan2-pytorch/op/upfirdn2d.cpp:12-22, bound at op/upfirdn2d.py:12-16).

    upfirdn2d_op.upfirdn2d(input[M,H,W,1], kernel[kh,kw], up_x, up_y, down_x, down_y,
                           pad_x0, pad_x1, pad_y0, pad_y1) -> [M,H',W',1]

Negative pads crop.  Non-CUDA tensors raise RuntimeError (CHECK_CUDA, upfirdn2d.cpp:7,16-17)."""
import torch

from gan2shape_amd import lib as _lib

_DT = {torch.float32: _lib.G2S_F32, torch.float16: _lib.G2S_F16}


def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    if not input.is_cuda:
        raise RuntimeError("input must be a CUDA tensor")
    if not kernel.is_cuda:
        raise RuntimeError("kernel must be a CUDA tensor")
    if input.dim() != 4:
        raise RuntimeError("input must be [major, in_h, in_w, minor]")
    if input.dtype not in _DT:
        raise RuntimeError(f"upfirdn2d: unsupported dtype {input.dtype}")
    major, in_h, in_w, minor = input.shape
    x = input.contiguous()
    if minor != 1:  # never reached from GAN2Shape (op/upfirdn2d.py:98 always reshapes to minor 1)
        x = x.permute(0, 3, 1, 2).contiguous()
    k = kernel.contiguous().float()
    kh, kw = k.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) // down_y
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) // down_x
    y = torch.empty((major * minor, out_h, out_w), dtype=x.dtype, device=x.device)
    L = _lib.load()
    _lib.check(L.g2s_upfirdn2d(_lib.ptr(x), _lib.ptr(k), _lib.ptr(y), major * minor, in_h, in_w,
                               kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1,
                               _DT[x.dtype], _lib.stream()))
    if minor != 1:
        return y.view(major, minor, out_h, out_w).permute(0, 2, 3, 1).contiguous()
    return y.view(major, out_h, out_w, 1)
