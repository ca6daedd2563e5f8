"""MI355X-native GAN2Shape inner loop (hand-written HIP kernels behind a C ABI + the host-side
mirror of the reference's operator / model-step interfaces).  Import as `gan2shape_amd`."""
from . import lib  # noqa: F401

__all__ = ["lib"]
