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
        for f_fake, f_real in zip(fake[:count], real[:count]):
            weight = None
            if mask is not None:
                step = (mask.shape[2] // f_fake.shape[2], mask.shape[3] // f_fake.shape[3])
                weight = F.avg_pool2d(mask, kernel_size=step, stride=step)
            total = total + _weighted_mean((f_fake - f_real).abs(), weight)
        return total


class PhotometricLoss():
    """L1 between two images, optionally confidence-weighted (sigma) and mask-averaged
    (losses.py:39-51)."""
    EPS = 1e-7

    def __call__(self, image1, image2, mask=None, conf_sigma=None):
        err = (image1 - image2).abs()
        if conf_sigma is not None:
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
