"""GroupNorm + ReLU / LeakyReLU as one op (g2s_groupnorm_act_fwd / _bwd, libg2s.so): the
nn.GroupNorm -> activation pairs of the depth / albedo nets (GAN2Shape/networks.py:88-127).
No native fallback."""
import torch
from torch.autograd import Function

from gan2shape_amd import lib as _lib


class GroupNormActFunction(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, act, alpha):
        _lib.require_cuda(x, gamma, beta)
        if x.dtype != torch.float32:
            raise RuntimeError("groupnorm_act: float32 only")
        x, gamma, beta = x.contiguous(), gamma.contiguous(), beta.contiguous()
        B, C = x.shape[:2]
        HW = x[0, 0].numel()
        L = _lib.load()
        y = torch.empty_like(x)
        mean = torch.empty((B, groups), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = torch.empty(L.g2s_groupnorm_workspace_floats(B, C, HW, groups), dtype=torch.float32,
                         device=x.device)
        _lib.check(L.g2s_groupnorm_act_fwd(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y),
                                           _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(ws), B, C, HW,
                                           groups, float(eps), int(act), float(alpha), _lib.stream()))
        ctx.save_for_backward(x, y, gamma, mean, rstd)
        ctx.cfg = (groups, int(act), float(alpha))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, mean, rstd = ctx.saved_tensors
        groups, act, alpha = ctx.cfg
        B, C = x.shape[:2]
        HW = x[0, 0].numel()
        L = _lib.load()
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(L.g2s_groupnorm_workspace_floats(B, C, HW, groups), dtype=torch.float32,
                         device=x.device)
        _lib.check(L.g2s_groupnorm_act_bwd(_lib.ptr(gy.contiguous()), _lib.ptr(y), _lib.ptr(x),
                                           _lib.ptr(gamma), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(dx),
                                           _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws), B, C, HW,
                                           groups, act, alpha, _lib.stream()))
        return dx, dgamma, dbeta, None, None, None, None


def groupnorm_act(x, gamma, beta, groups, eps=1e-5, act=True, negative_slope=0.0):
    """act(group_norm(x, groups, gamma, beta, eps)); act = leaky-ReLU(negative_slope), 0 = ReLU."""
    return GroupNormActFunction.apply(x, gamma, beta, groups, eps, 1 if act else 0, negative_slope)


def supported(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x[0, 0].numel() % 4 == 0
