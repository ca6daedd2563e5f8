"""The three losses of the step (behaviour of GAN2Shape/losses.py:6-79): discriminator-feature L1,
masked photometric L1, second-order smoothness.  On the GPU the smoothness loss is one fused
kernel per direction (fused_geometry.smooth_loss)."""
import torch
import torch.nn.functional as F


def _weighted_mean(err, weight):
    """sum(err * w) / sum(w) with w broadcast to err; plain mean when w is None."""
    if weight is None:
        return err.mean()
    weight = weight.expand_as(err)
    return (err * weight).sum() / weight.sum()


class _WeightedL1(torch.autograd.Function):
    """sum(|x - y| * w.expand_as(x)) / sum(w.expand_as(x)) for x, y (B, C, H, W) and a mask w
    (B, 1, H, W) — one kernel per direction (g2s_weighted_l1_fwd / _bwd) instead of sub, abs, expand,
    mul and two full-size reductions.  Gradient w.r.t. x only (y is the detached real branch / the
    target image, w a detached mask)."""

    @staticmethod
    def forward(ctx, x, y, w):
        from . import lib as _lib
        x, y = x.contiguous(), y.contiguous()
        B, C, H, W = x.shape
        L = _lib.load()
        num = torch.zeros((), dtype=torch.float32, device=x.device)
        wc = None if w is None else w.contiguous()
        _lib.check(L.g2s_weighted_l1_fwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(wc), _lib.ptr(num), B, C, H * W,
                                         _lib.stream()))
        den = (wc.sum() * C) if wc is not None else float(x.numel())
        ctx.save_for_backward(x, y, wc, den if torch.is_tensor(den) else None)
        ctx.den = None if torch.is_tensor(den) else den
        return num / den

    @staticmethod
    def backward(ctx, g):
        from . import lib as _lib
        x, y, wc, den = ctx.saved_tensors
        coef = (g / (den if den is not None else ctx.den)).reshape(1).float().contiguous()
        gx = torch.empty_like(x)
        B, C, H, W = x.shape
        L = _lib.load()
        _lib.check(L.g2s_weighted_l1_bwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(wc), _lib.ptr(coef), _lib.ptr(gx), B, C,
                                         H * W, _lib.stream()))
        return gx, None, None


def _masked_l1(a, b, weight):
    """(|a - b| * w).sum() / w.sum() with w (B,1,H,W) or (B,C,H,W)-expandable; fused on the GPU."""
    if (a.is_cuda and a.dtype == torch.float32 and a.dim() == 4 and a.shape == b.shape and not b.requires_grad
            and (a.shape[2] * a.shape[3]) % 4 == 0
            and (weight is None or (weight.dim() == 4 and weight.shape[1] == 1 and weight.shape[0] == a.shape[0]
                                    and weight.shape[2:] == a.shape[2:] and not weight.requires_grad))):
        return _WeightedL1.apply(a, b, weight)
    return _weighted_mean((a - b).abs(), weight)


class DiscriminatorLoss():
    """Sum over the first `ftr_num` discriminator feature maps of the L1 distance between the
    features of `fake_img` and `real_img`; a mask is box-averaged down to each feature resolution
    and used as the weight (losses.py:6-36).  The real branch runs without autograd."""

    def __init__(self, ftr_num=4, data_parallel=False):
        if data_parallel:
            raise NotImplementedError("data_parallel is disabled in the reference (model.py:82)")
        self.ftr_num = ftr_num

    def _features(self, D, image):
        return D(image, self.ftr_num)[1]

    def __call__(self, D, fake_img, real_img, mask=None):
        with torch.no_grad():
            real = self._features(D, real_img.detach())
        fake = self._features(D, fake_img)
        count = len(fake) if self.ftr_num is None else self.ftr_num
        total = 0
        pooled = {}     # mask averaged down to a feature resolution, each level from the previous one
        for f_fake, f_real in zip(fake[:count], real[:count]):
            weight = None
            if mask is not None:
                step = (mask.shape[2] // f_fake.shape[2], mask.shape[3] // f_fake.shape[3])
                weight = pooled.get(step)
                if weight is None:
                    # box average over step x step (losses.py:27-31); a power-of-two step is reached by
                    # 2x2 averages of the previous level (same value, sums re-associated)
                    prev = pooled.get((step[0] // 2, step[1] // 2)) if step[0] % 2 == 0 and step[1] % 2 == 0 else None
                    weight = F.avg_pool2d(prev, 2, 2) if prev is not None else \
                        F.avg_pool2d(mask, kernel_size=step, stride=step)
                    pooled[step] = weight
            total = total + _masked_l1(f_fake, f_real, weight)
        return total


class PhotometricLoss():
    """L1 between two images, optionally confidence-weighted (sigma) and mask-averaged
    (losses.py:39-51)."""
    EPS = 1e-7

    def __call__(self, image1, image2, mask=None, conf_sigma=None):
        if conf_sigma is None:
            return _masked_l1(image1, image2, mask)
        err = (image1 - image2).abs()
        sigma = conf_sigma + self.EPS
        err = err * 2 ** 0.5 / sigma + sigma.log()
        return _weighted_mean(err, mask)


class SmoothLoss():
    """mean|d2/dx2| + mean|d2/dxdy| + mean|d2/dydx| + mean|d2/dy2| of a (N,H,W) / (N,C,H,W) map;
    a list of maps is a pyramid weighted 1, 1/2.3, 1/2.3^2, ... (losses.py:54-79)."""

    @staticmethod
    def gradient(pred):
        """Forward differences along x and y: (D_dx, D_dy)."""
        planes = pred.reshape(-1, pred.shape[-2], pred.shape[-1]) if pred.dim() == 4 else pred
        return torch.diff(planes, dim=2), torch.diff(planes, dim=1)

    def _one(self, pred):
        first = self.gradient(pred)
        second = [d for g in first for d in self.gradient(g)]     # dx2, dxdy, dydx, dy2
        return sum(d.abs().mean() for d in second)

    def __call__(self, pred_map):
        if torch.is_tensor(pred_map) and pred_map.is_cuda and pred_map.dtype == torch.float32 \
                and pred_map.dim() in (3, 4):
            from .fused_geometry import smooth_loss  # one kernel forward, one backward
            return smooth_loss(pred_map)
        pyramid = pred_map if isinstance(pred_map, (tuple, list)) else [pred_map]
        loss, weight = 0, 1
        for level in pyramid:
            loss = loss + self._one(level) * weight
            weight /= 2.3
        return loss
