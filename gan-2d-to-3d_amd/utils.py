"""Small tensor helpers with the behaviour of GAN2Shape/utils.py:12-41 (resize, crop,
get_mask_range), written for this package."""
import torch
import torch.nn.functional as F


def resize(image, size):
    """Resize the last two dims to `size` = (H, W): bilinear when the target is taller than the
    source, area (box) averaging when it is shorter, untouched when the heights agree.  A 3-D
    tensor is treated as a batch of single-channel maps (utils.py:12-23)."""
    single_channel = image.dim() == 3
    maps = image[:, None] if single_channel else image
    src_h = maps.shape[-2]
    if size[0] != src_h:
        maps = F.interpolate(maps, size, mode='bilinear' if size[0] > src_h else 'area')
    return maps[:, 0] if single_channel else maps


def crop(tensor, crop_size):
    """Centred square crop of an (N, C, S, S) tensor (utils.py:26-30)."""
    lo = (tensor.shape[2] - crop_size) // 2
    hi = lo + crop_size
    return tensor[:, :, lo:hi, lo:hi]


def get_mask_range(mask):
    """(max_y, min_y, max_x, min_x) of the True entries of a 2-D boolean mask, as float scalars
    (utils.py:33-41)."""
    rows, cols = torch.nonzero(mask, as_tuple=True)
    rows, cols = rows.float(), cols.float()
    return rows.max(), rows.min(), cols.max(), cols.min()
