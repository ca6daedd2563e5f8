"""Pin the C oracle (oracle/g2s_oracle.c) against golden vectors generated from the reference's
own Python (tests/golden/make_golden.py): op/upfirdn2d.py:157-198, op/fused_act.py:86-92,
stylegan2-pytorch/model.py:250-291."""
import math

import numpy as np
import pytest

from oracle import capi

UP_CASES = ["blur_up", "rgb_up", "d_blur3", "d_blur1", "down2", "crop"]


@pytest.mark.parametrize("name", UP_CASES)
def test_upfirdn2d_forward(golden, name):
    g = golden("ops")
    up, down, p0, p1 = (int(v) for v in g[f"upfirdn2d.{name}.args"])
    y = capi.upfirdn2d(g[f"upfirdn2d.{name}.x"], g[f"upfirdn2d.{name}.k"], (up, up), (down, down),
                       (p0, p1, p0, p1))
    assert y.shape == g[f"upfirdn2d.{name}.y"].shape
    np.testing.assert_allclose(y, g[f"upfirdn2d.{name}.y"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", UP_CASES)
def test_upfirdn2d_backward_is_adjoint_op(golden, name):
    """op/upfirdn2d.py:100-115: the gradient is the same op with up/down swapped, the flipped
    kernel and g_pad."""
    g = golden("ops")
    up, down, p0, p1 = (int(v) for v in g[f"upfirdn2d.{name}.args"])
    x, k, gy = g[f"upfirdn2d.{name}.x"], g[f"upfirdn2d.{name}.k"], g[f"upfirdn2d.{name}.gy"]
    kh, kw = k.shape
    in_h, in_w = x.shape[2:]
    out_h, out_w = gy.shape[2:]
    gp_x0 = kw - p0 - 1
    gp_y0 = kh - p0 - 1
    gp_x1 = in_w * up - out_w * down + p0 - up + 1
    gp_y1 = in_h * up - out_h * down + p0 - up + 1
    gx = capi.upfirdn2d(gy, k[::-1, ::-1].copy(), (down, down), (up, up), (gp_x0, gp_x1, gp_y0, gp_y1))
    np.testing.assert_allclose(gx, g[f"upfirdn2d.{name}.gx"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["4d", "2d"])
def test_fused_bias_act(golden, name):
    g = golden("ops")
    x, b, gy = g[f"fused.{name}.x"], g[f"fused.{name}.b"], g[f"fused.{name}.gy"]
    y = capi.fused_bias_act(x, b, None, 3, 0, 0.2, math.sqrt(2))
    np.testing.assert_allclose(y, g[f"fused.{name}.y"], rtol=1e-6, atol=1e-7)
    # backward form (op/fused_act.py:20-38): act=3, grad=1, ref = forward output, no bias
    gx = capi.fused_bias_act(gy, None, y, 3, 1, 0.2, math.sqrt(2))
    np.testing.assert_allclose(gx, g[f"fused.{name}.gx"], rtol=1e-6, atol=1e-7)
    dims = (0,) + tuple(range(2, gx.ndim))
    np.testing.assert_allclose(gx.sum(dims), g[f"fused.{name}.gb"], rtol=1e-5, atol=1e-6)


def test_fused_bias_act_other_modes():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 4)).astype(np.float32)
    b = rng.standard_normal(3).astype(np.float32)
    xb = x + b[None, :, None]
    np.testing.assert_array_equal(capi.fused_bias_act(x, b, None, 1, 0, 0.2, 2.0), xb * 2.0)
    np.testing.assert_array_equal(capi.fused_bias_act(x, b, None, 3, 2, 0.2, 2.0), np.zeros_like(x))
    np.testing.assert_array_equal(capi.fused_bias_act(x, None, None, 1, 2, 0.2, 2.0), np.zeros_like(x))


def _style_mod(g, name):
    """EqualLinear(style_dim, cin, bias_init=1) — model.py:157-167, lr_mul=1."""
    w, b, s = g[f"{name}.mod_weight"], g[f"{name}.mod_bias"], g[f"{name}.s"]
    return (s @ (w * (1 / math.sqrt(w.shape[1]))).T + b).astype(np.float32)


@pytest.mark.parametrize("name,mode,demod", [("plain", 0, True), ("rgb", 0, False)])
def test_modconv_plain(golden, name, mode, demod):
    g = golden("modconv")
    w = g[f"{name}.weight"][0]
    scale = 1 / math.sqrt(w.shape[1] * w.shape[2] ** 2)
    y = capi.modconv(g[f"{name}.x"], w, _style_mod(g, name), scale, demod, mode)
    np.testing.assert_allclose(y, g[f"{name}.y"], rtol=1e-4, atol=1e-5)


def test_modconv_upsample_then_blur(golden):
    """model.py:264-275: conv_transpose2d stride 2, then Blur(pad=(1,1), kernel*4)."""
    g = golden("modconv")
    w = g["up.weight"][0]
    scale = 1 / math.sqrt(w.shape[1] * 9)
    style = _style_mod(g, "up")
    np.testing.assert_allclose(style, g["up.style_mod"], rtol=1e-5, atol=1e-6)
    y = capi.modconv(g["up.x"], w, style, scale, True, 1)
    assert y.shape[2] == 2 * g["up.x"].shape[2] + 1
    k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32)
    k = k / k.sum() * 4
    y = capi.upfirdn2d(y, k, pad=(1, 1, 1, 1))
    np.testing.assert_allclose(y, g["up.y"], rtol=1e-4, atol=1e-5)


def test_modconv_downsample_after_blur(golden):
    """model.py:277-283: Blur(pad=(2,2)) then conv2d stride 2."""
    g = golden("modconv")
    w = g["down.weight"][0]
    scale = 1 / math.sqrt(w.shape[1] * 9)
    k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32)
    k = k / k.sum()
    xb = capi.upfirdn2d(g["down.x"], k, pad=(2, 2, 2, 2))
    y = capi.modconv(xb, w, _style_mod(g, "down"), scale, True, 2)
    np.testing.assert_allclose(y, g["down.y"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape,groups,slope", [((2, 8, 6, 6), 4, 0.2), ((1, 32, 16, 16), 8, 0.0), ((3, 6, 4, 4), 6, 0.0)])
def test_group_norm_act_oracle_matches_torch_cpu(shape, groups, slope):
    """Pins oracle/nets.py against torch's CPU GroupNorm + (leaky-)ReLU, forward and gradients."""
    import torch
    from oracle import nets
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape) * 2 + 0.5
    gamma, beta = rng.standard_normal(shape[1]), rng.standard_normal(shape[1])
    gy = rng.standard_normal(shape)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.leaky_relu(torch.nn.functional.group_norm(xt, groups, gt, bt, 1e-5), slope)
    y.backward(torch.tensor(gy))
    np.testing.assert_allclose(nets.group_norm_act(x, gamma, beta, groups, 1e-5, True, slope), y.detach().numpy(),
                               rtol=1e-10, atol=1e-12)
    dx, dg, db = nets.group_norm_act_grad(x, gamma, beta, gy, groups, 1e-5, True, slope)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dg, gt.grad.numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("mode,h", [(0, 7), (1, 5), (2, 9)])
def test_adjoint_weight_identities(mode, h):
    """tests/conv_cases.py: the data-gradient of each convolution geometry written as a forward
    call of the oracle with adjoint weights equals torch's autograd of the reference formulation
    (F.conv2d pad k//2 / F.conv_transpose2d stride 2 / F.conv2d stride 2, model.py:264-289)."""
    import torch
    import torch.nn.functional as F
    from conv_cases import expected_modconv
    gen = torch.Generator().manual_seed(mode)
    B, cin, cout, k = 2, 5, 7, 3
    x = torch.randn(B, cin, h, h, generator=gen, requires_grad=True)
    w = torch.randn(cout, cin, k, k, generator=gen) / 5
    s = 1 + 0.3 * torch.randn(B, cin, generator=gen)
    d = 1 + 0.3 * torch.randn(B, cout, generator=gen)
    xs = x * s[:, :, None, None]
    if mode == 0:
        y = F.conv2d(xs, w, padding=1)
    elif mode == 1:
        y = F.conv_transpose2d(xs, w.transpose(0, 1), stride=2)
    else:
        y = F.conv2d(xs, w, stride=2)
    y = y * d[:, :, None, None]
    fwd = expected_modconv(x.detach().numpy(), w.numpy(), s.numpy(), d.numpy(), mode, 0)
    np.testing.assert_allclose(fwd, y.detach().numpy(), rtol=1e-5, atol=1e-5)
    gy = torch.randn(y.shape, generator=gen)
    (gxs,) = torch.autograd.grad(y, xs, gy)
    # g2s_modconv(transpose=1): in_scale = demod on the incoming gradient, no output scale
    bwd = expected_modconv(gy.numpy(), w.numpy(), d.numpy(), None, mode, 1)
    np.testing.assert_allclose(bwd, gxs.numpy(), rtol=1e-5, atol=1e-5)
