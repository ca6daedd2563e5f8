"""Mirror of GAN2Shape/losses.py:6-79 (DiscriminatorLoss, PhotometricLoss, SmoothLoss)."""
import torch
import torch.nn as nn


class DiscriminatorLoss():
    """L1 between the first `ftr_num` discriminator feature maps, mask average-pooled to each
    feature resolution (losses.py:6-36).  The real branch runs without autograd."""

    def __init__(self, ftr_num=4, data_parallel=False):
        if data_parallel:
            raise NotImplementedError("data_parallel is disabled in the reference (model.py:82)")
        self.ftr_num = ftr_num

    def __call__(self, D, fake_img, real_img, mask=None):
        with torch.no_grad():
            _, real_feature = D(real_img.detach(), self.ftr_num)
        _, fake_feature = D(fake_img, self.ftr_num)
        losses = []
        ftr_num = self.ftr_num if self.ftr_num is not None else len(fake_feature)
        for i in range(ftr_num):
            loss = torch.abs(fake_feature[i] - real_feature[i])
            if mask is not None:
                b, c, h, w = loss.shape
                _, _, hm, wm = mask.shape
                sh, sw = hm // h, wm // w
                mask0 = nn.functional.avg_pool2d(mask, kernel_size=(sh, sw),
                                                 stride=(sh, sw)).expand_as(loss)
                loss = (loss * mask0).sum() / mask0.sum()
            else:
                loss = loss.mean()
            losses += [loss]
        return sum(losses)


class PhotometricLoss():
    EPS = 1e-7

    def __call__(self, image1, image2, mask=None, conf_sigma=None):
        loss = (image1 - image2).abs()
        if conf_sigma is not None:
            loss = loss * 2 ** 0.5 / (conf_sigma + self.EPS) + (conf_sigma + self.EPS).log()
        if mask is not None:
            mask = mask.expand_as(loss)
            loss = (loss * mask).sum() / mask.sum()
        else:
            loss = loss.mean()
        return loss


class SmoothLoss():
    """Second-order smoothness (losses.py:54-79)."""

    def __call__(self, pred_map):
        if torch.is_tensor(pred_map) and pred_map.is_cuda and pred_map.dtype == torch.float32 \
                and pred_map.dim() in (3, 4):
            from .fused_geometry import smooth_loss  # one kernel forward, one backward
            return smooth_loss(pred_map)
        if type(pred_map) not in [tuple, list]:
            pred_map = [pred_map]
        loss = 0
        weight = 1
        for scaled_map in pred_map:
            dx, dy = self.gradient(scaled_map)
            dx2, dxdy = self.gradient(dx)
            dydx, dy2 = self.gradient(dy)
            loss += (dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()) * weight
            weight /= 2.3
        return loss

    def gradient(self, pred):
        if pred.dim() == 4:
            pred = pred.reshape(-1, pred.size(2), pred.size(3))
        D_dy = pred[:, 1:] - pred[:, :-1]
        D_dx = pred[:, :, 1:] - pred[:, :, :-1]
        return D_dx, D_dy
