"""numpy (float64) restatement of network glue used inside the step — TEST INFRASTRUCTURE ONLY.

group_norm_act: torch.nn.GroupNorm (biased variance, eps inside the square root, per-channel affine)
followed by ReLU / LeakyReLU, as the depth / albedo nets chain them
(GAN2Shape/networks.py:88-127).  The reference has no fixture for this pair; the restatement is
pinned against torch's own CPU GroupNorm in tests/test_oracle_ops.py.
"""
import numpy as np


def group_norm_act(x, gamma, beta, groups, eps=1e-5, act=True, slope=0.0):
    x = np.asarray(x, np.float64)
    B, C = x.shape[:2]
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(-1, keepdims=True)
    var = xg.var(-1, keepdims=True)
    xh = ((xg - mean) / np.sqrt(var + eps)).reshape(x.shape)
    shape = (1, C) + (1,) * (x.ndim - 2)
    y = xh * np.asarray(gamma, np.float64).reshape(shape) + np.asarray(beta, np.float64).reshape(shape)
    if act:
        y = np.where(y > 0, y, y * slope)
    return y


def group_norm_act_grad(x, gamma, beta, gy, groups, eps=1e-5, act=True, slope=0.0):
    """Analytic gradients (dx, dgamma, dbeta) of sum(gy * group_norm_act(x))."""
    x = np.asarray(x, np.float64)
    gy = np.asarray(gy, np.float64)
    B, C = x.shape[:2]
    shape = (1, C) + (1,) * (x.ndim - 2)
    ga = np.asarray(gamma, np.float64).reshape(shape)
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(xg.var(-1, keepdims=True) + eps)
    xh = ((xg - mean) * rstd).reshape(x.shape)
    y = xh * ga + np.asarray(beta, np.float64).reshape(shape)
    g = gy * (np.where(y > 0, 1.0, slope) if act else 1.0)
    axes = (0,) + tuple(range(2, x.ndim))
    dgamma, dbeta = (g * xh).sum(axes), g.sum(axes)
    gg = (g * ga).reshape(B, groups, -1)
    xhg = xh.reshape(B, groups, -1)
    dx = rstd * (gg - gg.mean(-1, keepdims=True) - xhg * (gg * xhg).mean(-1, keepdims=True))
    return dx.reshape(x.shape), dgamma, dbeta
