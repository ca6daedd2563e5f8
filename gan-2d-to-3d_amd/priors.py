"""Depth priors for the depth-net pre-training (behaviour of GAN2Shape/priors.py:7-107).

The reference derives the object mask from a parsing network (MaskingModel, model.py:473-551:
BiSeNet / PSPNet checkpoints that are not available offline).  Here the mask source is an argument
(`masking_model(image) -> (1,1,H,W) soft mask in [0,1]`); with none given a centred elliptical
synthetic mask is used (benchmarks, tests).  The confidence priors take the parsing net's
confidence map through the same callable.

Priors are maps in [near, far] (smaller = closer): `far` outside the object.
"""
import math

import torch
import torch.nn.functional as F

from . import utils


def synthetic_mask(image, rx=0.33, ry=0.42):
    """Centred soft ellipse, (1, 1, H, W)."""
    h, w = image.shape[-2:]
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    r = torch.sqrt((xx / (2 * rx)) ** 2 + (yy / (2 * ry)) ** 2)
    return (1 - r).clamp(0, 1).mul(4).clamp(0, 1)[None, None]


class PriorGenerator():
    """prior_name in {box, masked_box, smoothed_box, ellipsoid, confidence, smoothed_confidence}."""

    SMOOTH_TAPS = 11      # box-filter side of the smoothing passes (priors.py:49-51)
    SMOOTH_PASSES = 3
    ELLIPSOID_RADIUS = 0.4

    def __init__(self, image_size, category, prior, noise_threshold=0.7, near=0.91, far=1.02,
                 masking_model=None):
        self._build = getattr(self, f'_{prior}_prior', None)
        if self._build is None:
            raise NotImplementedError()
        self.image_size, self.category, self.prior = image_size, category, prior
        self.noise_threshold, self.near, self.far = noise_threshold, near, far
        self.base_prior = torch.full((1, image_size, image_size), float(far))
        self.masking_model = synthetic_mask if masking_model is None else masking_model

    def __call__(self, image, device='cuda', *args, **kwargs):
        with torch.no_grad():
            return self._build(image, *args, **kwargs).to(device)

    # ---- mask sources
    def _mask(self, image):
        return self.masking_model(image)[0].float().cpu()

    # ---- priors
    def _box_prior(self, _image):
        """1 inside a centred 0.8 S x 0.5 S box, 0 outside (priors.py:26-33)."""
        S = self.image_size
        centre, half_rows, half_cols = S // 2, int(S * 0.8 * 0.5), int(S * 0.5 * 0.5)
        box = torch.zeros(1, S, S)
        box[0, centre - half_rows:centre + half_rows, centre - half_cols:centre + half_cols] = 1
        return box

    def _masked_box_prior(self, image):
        """far * (1 - m) with the mask re-normalised above the noise threshold (priors.py:35-45)."""
        t = self.noise_threshold
        mask = self._mask(image)
        mask = torch.where(mask < t, torch.zeros_like(mask), mask)
        return self.far - self.base_prior * ((mask - t) / (1 - t))

    def _smooth(self, prior):
        """SMOOTH_PASSES x (valid box filter, rescale to [near, far], pad back with far)
        (priors.py:47-67; the filter is ones / ||ones||, i.e. 1/11 per tap)."""
        taps = self.SMOOTH_TAPS
        box = torch.full((1, 1, taps, taps), 1.0 / taps)
        x = prior[None]
        for _ in range(self.SMOOTH_PASSES):
            x = F.conv2d(x, box)
            lo, hi = x.min(), x.max()
            x = self.near + (x - lo) * (self.far - self.near) / (hi - lo)
            x = F.pad(x, (taps // 2,) * 4, value=self.far)
        return x[0]

    def _smoothed_box_prior(self, image):
        return self._smooth(self._masked_box_prior(image))

    def _ellipsoid_prior(self, image):
        """Spherical cap of radius 0.4 over the mask's bounding ellipse: depth `near` at the centre,
        `far` at the rim and outside (priors.py:74-97)."""
        S, R = self.image_size, self.ELLIPSOID_RADIUS
        inside = self._mask(image)[0] >= self.noise_threshold
        top, bottom, right, left = utils.get_mask_range(inside)   # max_y, min_y, max_x, min_x
        half_width = (right - left) / 2
        aspect = (top - bottom) / (right - left)
        cx, cy = (right + left) / 2, (top + bottom) / 2
        axis = torch.arange(S, dtype=torch.float32)
        rows = (axis[:, None] - S / 2) / aspect + S / 2           # rows squeezed to a circle
        cols = axis[None, :]
        dist = torch.sqrt((rows - cy) ** 2 + (cols - cx) ** 2)
        rim = math.sqrt(R ** 2 - (R - (self.far - self.near)) ** 2)   # cap half-width at depth far
        rho = dist / half_width * rim
        cap = R - torch.sqrt(torch.abs(R ** 2 - rho ** 2)) + self.near
        prior = self.base_prior.clone()
        prior[0] = torch.where(dist <= half_width, cap, prior[0])
        return prior

    def _confidence_prior(self, image):
        """far * (1 - confidence) (priors.py:99-103; the callable returns the confidence map)."""
        return self.far - self.base_prior * self._mask(image)

    def _smoothed_confidence_prior(self, image):
        return self._smooth(self._confidence_prior(image))
