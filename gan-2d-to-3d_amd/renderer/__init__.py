"""Mirror of GAN2Shape/renderer/__init__.py."""
from .renderer import Renderer

__all__ = ["Renderer"]
