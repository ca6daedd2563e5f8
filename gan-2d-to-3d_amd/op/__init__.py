"""Host-side mirror of GAN2Shape/stylegan2/stylegan2-pytorch/op/__init__.py:1-2."""
from .fused_act import FusedLeakyReLU, add_bias_scale, clamp, fused_leaky_relu, fused_noise_bias_act
from .grid_sample import grid_sample
from .upfirdn2d import upfirdn2d

__all__ = ["FusedLeakyReLU", "add_bias_scale", "fused_leaky_relu", "fused_noise_bias_act", "grid_sample", "upfirdn2d"]
