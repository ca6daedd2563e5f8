"""F.grid_sample(x, grid, 'bilinear', align_corners=True) (+ .clamp(lo, hi)) of the reconstruction warp
(GAN2Shape/model.py:147-150, 267-270) on libg2s.so: one launch forward, one backward, and — unlike
ATen's grid_sampler_2d_backward — a bit-reproducible texture gradient in deterministic mode
(csrc/grid_sample.hip)."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import lib as _lib
from .. import zeropool as _zp


class _GridSample(Function):
    @staticmethod
    def forward(ctx, x, grid, lo, hi):
        x, grid = x.contiguous(), grid.contiguous()
        B, C, IH, IW = x.shape
        _, H, W, _ = grid.shape
        y = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        clamp = lo is not None
        ctx.args = (B, C, IH, IW, H, W, int(clamp), float(lo) if clamp else 0.0, float(hi) if clamp else 0.0)
        _lib.check(_lib.load().g2s_grid_sample_fwd(_lib.ptr(x), _lib.ptr(grid), _lib.ptr(y), *ctx.args, _lib.stream()))
        ctx.save_for_backward(x, grid)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, grid = ctx.saved_tensors
        L = _lib.load()
        need_x, need_grid = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gg = torch.empty_like(grid) if need_grid else None
        if not need_x and gg is None:
            return None, None, None, None
        # the scatter target (default mode: gx itself; deterministic mode: the fixed-point workspace) must start
        # at zero: a slice of the step's cleared pool when there is one, else the call clears it with a memset
        gx, ws, ws_bytes, pre = None, None, 0, False
        if need_x and L.g2s_get_deterministic():
            ws_bytes = L.g2s_grid_sample_bwd_workspace_bytes(*ctx.args[:4])
            ws = _zp.take(((ws_bytes + 3) // 4,), x.device)
            pre = ws is not None
            if ws is None:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            gx = torch.empty_like(x)
        elif need_x:
            gx = _zp.take(tuple(x.shape), x.device)
            pre = gx is not None
            if gx is None:
                gx = torch.empty_like(x)
        with _lib.precleared(pre):
            _lib.check(L.g2s_grid_sample_bwd(_lib.ptr(gy.contiguous()), _lib.ptr(x), _lib.ptr(grid), _lib.ptr(gx), _lib.ptr(gg),
                                             *ctx.args, _lib.ptr(ws), ws_bytes, _lib.stream()))
        return gx, gg, None, None


def grid_sample(x, grid, lo=None, hi=None):
    """F.grid_sample(x, grid, mode='bilinear', align_corners=True), then .clamp(lo, hi) when bounds are
    given.  CUDA float32: libg2s; anything else: the torch ops."""
    if x.is_cuda and x.dtype == torch.float32 and grid.dtype == torch.float32 and x.dim() == 4 and x.numel() > 0 \
            and grid.shape[0] == x.shape[0] and min(x.shape[2:]) > 1:
        return _GridSample.apply(x, grid, lo, hi)
    y = F.grid_sample(x, grid, mode='bilinear', align_corners=True)
    return y if lo is None else y.clamp(min=lo, max=hi)
