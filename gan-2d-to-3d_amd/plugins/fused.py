"""Drop-in for the reference's pybind module `fused`
(GAN2Shape/stylegan2/stylegan2-pytorch/op/fused_bias_act.cpp:11-20, bound at op/fused_act.py:13-17).

    fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale) -> Tensor

Same argument meaning and error behaviour: `bias` / `refer` with 0 elements mean "absent"
(fused_act.py:27,55; fused_bias_act_kernel.cu:62-63); non-CUDA input raises RuntimeError
(CHECK_CUDA, fused_bias_act.cpp:7,13-14); inputs are made contiguous; the output is a fresh
tensor; the launch goes to the current stream without synchronising."""
import torch

from gan2shape_amd import lib as _lib

_DT = {torch.float32: _lib.G2S_F32, torch.float16: _lib.G2S_F16}


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    if not input.is_cuda:
        raise RuntimeError("input must be a CUDA tensor")
    if not bias.is_cuda:
        raise RuntimeError("bias must be a CUDA tensor")
    if input.dtype not in _DT:
        raise RuntimeError(f"fused_bias_act: unsupported dtype {input.dtype}")
    x = input.contiguous()
    b = bias.contiguous().to(x.dtype) if bias.numel() else None
    r = refer.contiguous().to(x.dtype) if refer.numel() else None
    if r is not None and r.numel() != x.numel():
        raise RuntimeError("refer must have as many elements as input")
    y = torch.empty_like(x)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    L = _lib.load()
    _lib.check(L.g2s_fused_bias_act(_lib.ptr(x), _lib.ptr(b), _lib.ptr(r), _lib.ptr(y), x.numel(),
                                    step_b, b.numel() if b is not None else 0, int(act), int(grad),
                                    float(alpha), float(scale), _DT[x.dtype], _lib.stream()))
    return y
