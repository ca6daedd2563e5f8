"""Autograd wrappers of csrc/geometry.hip: the renderer-geometry / loss glue of a GAN2Shape step as
single kernels (forward and analytic backward) instead of chains of elementwise launches.

Each function restates a piece of the reference (cited per class); the device-agnostic torch
implementations in renderer/renderer.py and losses.py stay the specification — the -m gpu tests
compare these kernels against them."""
import torch
from torch.autograd import Function

from . import lib as _lib
from . import zeropool as _zp


def _cleared(shape, device):
    """(tensor, from_pool): an accumulator the library call would otherwise clear with a memset of its own — a
    slice of the step's cleared pool when one is active (then the call runs under lib.precleared), else plain
    memory that the call clears itself."""
    t = _zp.take(tuple(shape), device)
    if t is not None:
        return t, True
    return torch.empty(shape, dtype=torch.float32, device=device), False


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


class ViewTransformFunction(Function):
    """view [B,6] -> (R = Rz Ry Rx [B,3,3], t [B,1,3]) with the scalings of
    GAN2Shape/model.py:330-335 folded in (renderer/utils.py:33-73)."""

    @staticmethod
    def forward(ctx, view, rot_scale, txy_scale, tz_scale):
        view = _f32c(view)
        B = view.shape[0]
        R = torch.empty((B, 3, 3), dtype=torch.float32, device=view.device)
        t = torch.empty((B, 1, 3), dtype=torch.float32, device=view.device)
        L = _lib.load()
        _lib.check(L.g2s_view_transform_fwd(_lib.ptr(view), rot_scale, txy_scale, tz_scale,
                                            _lib.ptr(R), _lib.ptr(t), B, _lib.stream()))
        ctx.save_for_backward(view)
        ctx.scales = (rot_scale, txy_scale, tz_scale)
        return R, t

    @staticmethod
    def backward(ctx, gR, gt):
        view, = ctx.saved_tensors
        B = view.shape[0]
        gR = torch.zeros((B, 3, 3), device=view.device) if gR is None else _f32c(gR)
        gt = torch.zeros((B, 1, 3), device=view.device) if gt is None else _f32c(gt)
        gview = torch.empty_like(view)
        L = _lib.load()
        _lib.check(L.g2s_view_transform_bwd(_lib.ptr(view), *ctx.scales, _lib.ptr(gR), _lib.ptr(gt),
                                            _lib.ptr(gview), B, _lib.stream()))
        return gview, None, None, None


class WarpVertsFunction(Function):
    """renderer.py:90-95 get_warped_3d_grid: R (d * ray - c) + c + t, vertices [B, H*W, 3]."""

    @staticmethod
    def forward(ctx, depth, rays, R, t, rcd):
        depth, R, t = _f32c(depth), _f32c(R), _f32c(t)
        B, H, W = depth.shape
        verts = torch.empty((B, H * W, 3), dtype=torch.float32, device=depth.device)
        L = _lib.load()
        _lib.check(L.g2s_warp_verts_fwd(_lib.ptr(depth), _lib.ptr(rays), _lib.ptr(R), _lib.ptr(t),
                                        float(rcd), _lib.ptr(verts), B, H * W, _lib.stream()))
        ctx.save_for_backward(depth, rays, R)
        ctx.rcd = float(rcd)
        return verts

    @staticmethod
    def backward(ctx, gverts):
        depth, rays, R = ctx.saved_tensors
        B, H, W = depth.shape
        need_rt = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        gdepth = torch.empty_like(depth)
        grt, pre = _cleared((B, 12), depth.device) if need_rt else (None, False)
        L = _lib.load()
        with _lib.precleared(pre):
            _lib.check(L.g2s_warp_verts_bwd(_lib.ptr(depth), _lib.ptr(rays), _lib.ptr(R),
                                            _lib.ptr(_f32c(gverts)), ctx.rcd, _lib.ptr(gdepth),
                                            _lib.ptr(grt), B, H * W, _lib.stream()))
        gR = grt[:, :9].reshape(B, 3, 3) if need_rt else None
        gt = grt[:, 9:].reshape(B, 1, 3) if need_rt else None
        return gdepth, None, gR, gt, None


class InvWarpGridFunction(Function):
    """renderer.py:110-114 get_inv_warped_2d_grid: R^T (d * ray - t - c) + c, projected with K and
    normalised to [-1, 1] (align_corners=True convention), grid [B, H, W, 2]."""

    @staticmethod
    def forward(ctx, depth, rays, R, t, K9, rcd):
        depth, R, t = _f32c(depth), _f32c(R), _f32c(t)
        B, H, W = depth.shape
        grid = torch.empty((B, H, W, 2), dtype=torch.float32, device=depth.device)
        Kc = (_lib.C.c_float * 9)(*K9)
        L = _lib.load()
        _lib.check(L.g2s_inv_warp_grid_fwd(_lib.ptr(depth), _lib.ptr(rays), _lib.ptr(R), _lib.ptr(t),
                                           Kc, float(rcd), _lib.ptr(grid), B, H, W, _lib.stream()))
        ctx.save_for_backward(depth, rays, R, t)
        ctx.meta = (K9, float(rcd))
        return grid

    @staticmethod
    def backward(ctx, ggrid):
        depth, rays, R, t = ctx.saved_tensors
        K9, rcd = ctx.meta
        B, H, W = depth.shape
        need_rt = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        gdepth = torch.empty_like(depth)
        grt, pre = _cleared((B, 12), depth.device) if need_rt else (None, False)
        Kc = (_lib.C.c_float * 9)(*K9)
        L = _lib.load()
        with _lib.precleared(pre):
            _lib.check(L.g2s_inv_warp_grid_bwd(_lib.ptr(depth), _lib.ptr(rays), _lib.ptr(R), _lib.ptr(t),
                                               Kc, rcd, _lib.ptr(_f32c(ggrid)), _lib.ptr(gdepth),
                                               _lib.ptr(grt), B, H, W, _lib.stream()))
        gR = grt[:, :9].reshape(B, 3, 3) if need_rt else None
        gt = grt[:, 9:].reshape(B, 1, 3) if need_rt else None
        return gdepth, None, gR, gt, None, None


class SmoothLossFunction(Function):
    """losses.py:54-79 SmoothLoss for one map ([N,H,W] or [B,C,H,W])."""

    @staticmethod
    def forward(ctx, pred):
        p = _f32c(pred).reshape(-1, pred.shape[-2], pred.shape[-1])
        N, H, W = p.shape
        loss, pre = _cleared((), p.device)
        L = _lib.load()
        with _lib.precleared(pre):
            _lib.check(L.g2s_smooth_loss_fwd(_lib.ptr(p), _lib.ptr(loss), N, H, W, _lib.stream()))
        ctx.save_for_backward(p)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, gloss):
        p, = ctx.saved_tensors
        N, H, W = p.shape
        gp = torch.empty_like(p)
        L = _lib.load()
        _lib.check(L.g2s_smooth_loss_bwd(_lib.ptr(p), _lib.ptr(_f32c(gloss)), _lib.ptr(gp), N, H, W,
                                         _lib.stream()))
        return gp.reshape(ctx.shape)


class NormalFunction(Function):
    """renderer.py:127-139 get_normal_from_depth."""

    @staticmethod
    def forward(ctx, depth, rays):
        depth = _f32c(depth)
        B, H, W = depth.shape
        normal = torch.empty((B, H, W, 3), dtype=torch.float32, device=depth.device)
        L = _lib.load()
        _lib.check(L.g2s_normal_fwd(_lib.ptr(depth), _lib.ptr(rays), _lib.ptr(normal), B, H, W,
                                    _lib.stream()))
        ctx.save_for_backward(depth, rays)
        return normal

    @staticmethod
    def backward(ctx, gnormal):
        depth, rays = ctx.saved_tensors
        B, H, W = depth.shape
        gdepth = torch.empty_like(depth)
        L = _lib.load()
        _lib.check(L.g2s_normal_bwd(_lib.ptr(depth), _lib.ptr(rays), _lib.ptr(_f32c(gnormal)),
                                    _lib.ptr(gdepth), B, H, W, _lib.stream()))
        return gdepth, None


class ShadingFunction(Function):
    """model.py:347-360 get_lighting_directions + get_shading: light [B,4] (after the mean is added),
    normal [Bn,H,W,3], albedo [Ba,3,H,W] with Bn, Ba in {1, B} -> diffuse [B,1,H,W], texture [B,3,H,W]."""

    @staticmethod
    def forward(ctx, normal, light, albedo):
        normal, light, albedo = _f32c(normal), _f32c(light), _f32c(albedo)
        B = light.shape[0]
        Bn, H, W, _ = normal.shape
        Ba = albedo.shape[0]
        diffuse = torch.empty((B, 1, H, W), dtype=torch.float32, device=light.device)
        texture = torch.empty((B, 3, H, W), dtype=torch.float32, device=light.device)
        L = _lib.load()
        _lib.check(L.g2s_shading_fwd(_lib.ptr(normal), _lib.ptr(light), _lib.ptr(albedo),
                                     _lib.ptr(diffuse), _lib.ptr(texture), B, Bn, Ba, H * W,
                                     _lib.stream()))
        ctx.save_for_backward(normal, light, albedo)
        return diffuse, texture

    @staticmethod
    def backward(ctx, gdiffuse, gtexture):
        normal, light, albedo = ctx.saved_tensors
        B = light.shape[0]
        Bn, H, W, _ = normal.shape
        Ba = albedo.shape[0]
        dev = light.device
        if gtexture is None:
            gtexture = torch.zeros((B, 3, H, W), dtype=torch.float32, device=dev)
        gnormal = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
        galbedo = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
        glight, pre = _cleared((B, 4), dev)
        gd = None if gdiffuse is None else _f32c(gdiffuse)
        L = _lib.load()
        with _lib.precleared(pre):
            _lib.check(L.g2s_shading_bwd(_lib.ptr(normal), _lib.ptr(light), _lib.ptr(albedo), _lib.ptr(gd),
                                         _lib.ptr(_f32c(gtexture)), _lib.ptr(gnormal), _lib.ptr(galbedo),
                                         _lib.ptr(glight), B, Bn, Ba, H * W, _lib.stream()))
        if Bn == 1 and B > 1:
            gnormal = gnormal.sum(0, keepdim=True)
        if Ba == 1 and B > 1:
            galbedo = galbedo.sum(0, keepdim=True)
        return gnormal, glight, galbedo


def normal_from_depth(depth, rays):
    return NormalFunction.apply(depth, rays)


class DepthHeadFunction(Function):
    """model.py:337-345 get_clamped_depth (+ rescale_depth): centre over the whole batch, tanh, rescale to
    [lo, hi], blend the border columns — two launches forward (the mean, then everything else), three
    backward, where the op-by-op chain takes 14 each way."""

    @staticmethod
    def forward(ctx, raw, W, lo, hi, clamp_border, border_depth):
        raw = _f32c(raw)
        mean = raw.view(1, -1).mean(1)
        out = torch.empty_like(raw)
        ctx.args = (raw.numel(), int(W), float(lo), float(hi), int(bool(clamp_border)), float(border_depth))
        _lib.check(_lib.load().g2s_depth_head_fwd(_lib.ptr(raw), _lib.ptr(mean), _lib.ptr(out), *ctx.args, _lib.stream()))
        ctx.save_for_backward(raw, mean)
        return out

    @staticmethod
    def backward(ctx, g):
        raw, mean = ctx.saved_tensors
        g_raw = torch.empty_like(raw)
        gsum, pre = _cleared((1,), raw.device)
        with _lib.precleared(pre):
            _lib.check(_lib.load().g2s_depth_head_bwd(_lib.ptr(raw), _lib.ptr(mean), _lib.ptr(_f32c(g)), _lib.ptr(g_raw),
                                                      _lib.ptr(gsum), *ctx.args, _lib.stream()))
        return g_raw, None, None, None, None, None


def depth_head(raw, W, lo, hi, clamp_border, border_depth):
    return DepthHeadFunction.apply(raw, W, lo, hi, clamp_border, border_depth)


def shading(normal, light, albedo):
    return ShadingFunction.apply(normal, light, albedo)


def view_transform(view, rot_scale=1.0, txy_scale=1.0, tz_scale=1.0):
    return ViewTransformFunction.apply(view, float(rot_scale), float(txy_scale), float(tz_scale))


def warp_verts(depth, rays, R, t, rcd):
    return WarpVertsFunction.apply(depth, rays, R, t, rcd)


def inv_warp_grid(depth, rays, R, t, K9, rcd):
    return InvWarpGridFunction.apply(depth, rays, R, t, K9, rcd)


def smooth_loss(pred):
    return SmoothLossFunction.apply(pred)
