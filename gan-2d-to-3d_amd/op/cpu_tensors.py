"""The reference's behaviour for CPU TENSORS in the Python op API: its `op` package answers a CPU
tensor with plain PyTorch (op/fused_act.py:86-92, op/upfirdn2d.py:144-154) and a CUDA tensor with
the native module.  Same split here — and only that: a CUDA tensor NEVER comes this way (it goes to
libg2s.so, and a missing library raises in lib.load()); this module is what lets the minimal config
(BASELINE.json configs[0]: CPU tensors, plumbing only) and CPU-side tooling call the op API.

Own formulation: the resampler is zero-insertion + padding/cropping + one depthwise correlation with
the flipped FIR + strided slicing, all differentiable torch ops."""
import torch
import torch.nn.functional as F


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    shape = [1, -1] + [1] * (x.dim() - 2)
    return F.leaky_relu(x + bias.view(*shape), negative_slope) * scale


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    n, c, h, w = x.shape
    planes = x.reshape(n * c, 1, h, w)
    if up > 1:                                   # zero insertion: sample (i, j) -> (i * up, j * up)
        z = planes.new_zeros(n * c, 1, h * up, w * up)
        z[:, :, ::up, ::up] = planes
        planes = z
    p0, p1 = pad
    planes = F.pad(planes, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    hh, ww = planes.shape[-2:]
    planes = planes[:, :, max(-p0, 0):hh - max(-p1, 0), max(-p0, 0):ww - max(-p1, 0)]   # negative pad = crop
    taps = torch.flip(kernel, [0, 1])[None, None].to(planes.dtype)                       # FIR = correlation with the flip
    out = F.conv2d(planes, taps)
    out = out[:, :, ::down, ::down]
    return out.reshape(n, c, out.shape[-2], out.shape[-1])
