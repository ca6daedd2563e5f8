"""Depth priors for the depth-net pre-training (GAN2Shape/priors.py:7-107).

The reference derives the object mask from a parsing network (MaskingModel, model.py:473-551:
BiSeNet / PSPNet checkpoints that are not available offline).  Here the mask is an argument
(`masking_model(image) -> (1,1,H,W) soft mask in [0,1]`); with none given a centred elliptical
synthetic mask is used (benchmarks, tests).  ellipsoid / masked_box / smoothed_box / box are
implemented; the confidence priors need the parsing net's confidence map and accept it through
the same callable.
"""
import math

import torch

from . import utils


def synthetic_mask(image, rx=0.33, ry=0.42):
    """Centred soft ellipse, (1, 1, H, W)."""
    h, w = image.shape[-2:]
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    r = torch.sqrt((xx / (2 * rx)) ** 2 + (yy / (2 * ry)) ** 2)
    return (1 - r).clamp(0, 1).mul(4).clamp(0, 1)[None, None]


class PriorGenerator():
    def __init__(self, image_size, category, prior, noise_threshold=0.7, near=0.91, far=1.02,
                 masking_model=None):
        self.image_size = image_size
        self.category = category
        self.prior = prior
        if not hasattr(self, f'_{prior}_prior'):
            raise NotImplementedError()
        self.noise_threshold = noise_threshold
        self.near = near
        self.far = far
        self.base_prior = torch.Tensor(1, self.image_size, self.image_size).fill_(far)
        self.masking_model = masking_model if masking_model is not None else synthetic_mask

    def __call__(self, image, device='cuda', *args, **kwargs):
        with torch.no_grad():
            prior = getattr(self, f'_{self.prior}_prior')(image, *args, **kwargs)
            return prior.to(device)

    def _mask(self, image):
        return self.masking_model(image)[0].float().cpu()

    def _box_prior(self, _):  # priors.py:26-33
        c = int(self.image_size / 2)
        bh, bw = int(self.image_size * 0.5 * 0.5), int(self.image_size * 0.8 * 0.5)
        prior = torch.zeros([1, self.image_size, self.image_size])
        prior[0, c - bw: c + bw, c - bh: c + bh] = 1
        return prior

    def _masked_box_prior(self, image):  # priors.py:35-45
        mask = self._mask(image).clone()
        mask[mask < self.noise_threshold] = 0
        mask = (mask - self.noise_threshold) / (1 - self.noise_threshold)
        return self.far - self.base_prior * mask

    def _smooth(self, prior):  # priors.py:47-67: 3 x (11x11 box filter, rescale to [near, far], pad)
        kernel_size, pad, n_convs = 11, 5, 3
        filt = torch.ones(1, 1, kernel_size, kernel_size)
        filt = filt / torch.norm(filt)
        prior = prior.unsqueeze(0)
        for _ in range(n_convs):
            prior = torch.nn.functional.conv2d(prior, filt)
            prior = self.near + ((prior - torch.min(prior)) * (self.far - self.near)) \
                / (torch.max(prior) - torch.min(prior))
            prior = torch.nn.functional.pad(prior, tuple([pad] * 4), value=self.far)
        return prior.squeeze(0)

    def _smoothed_box_prior(self, image):
        return self._smooth(self._masked_box_prior(image))

    def _ellipsoid_prior(self, image):  # priors.py:74-97
        radius = 0.4
        mask = self._mask(image)[0] >= self.noise_threshold
        max_y, min_y, max_x, min_x = utils.get_mask_range(mask)
        r_pixel = (max_x - min_x) / 2
        ratio = (max_y - min_y) / (max_x - min_x)
        c_x = (max_x + min_x) / 2
        c_y = (max_y + min_y) / 2
        i, j = torch.meshgrid(torch.linspace(0, self.image_size - 1, self.image_size),
                              torch.linspace(0, self.image_size - 1, self.image_size), indexing="ij")
        i = (i - self.image_size / 2) / ratio + self.image_size / 2
        temp = math.sqrt(radius ** 2 - (radius - (self.far - self.near)) ** 2)
        dist = torch.sqrt((i - c_y) ** 2 + (j - c_x) ** 2)
        area = dist <= r_pixel
        dist_rescale = dist / r_pixel * temp
        depth = radius - torch.sqrt(torch.abs(radius ** 2 - dist_rescale ** 2)) + self.near
        prior = torch.clone(self.base_prior)
        prior[0, area] = depth[area]
        return prior

    def _confidence_prior(self, image):  # priors.py:99-103 (the callable returns the confidence map)
        return self.far - self.base_prior * self._mask(image)

    def _smoothed_confidence_prior(self, image):
        return self._smooth(self._confidence_prior(image))
