"""Mirror of GAN2Shape/utils.py:12-41 (resize, crop, get_mask_range)."""
import torch
import torch.nn.functional as F


def resize(image, size):
    """bilinear up / area down / identity when sizes are equal (utils.py:12-23)."""
    dim = image.dim()
    if dim == 3:
        image = image.unsqueeze(1)
    b, _, h, w = image.shape
    if size[0] > h:
        image = F.interpolate(image, size, mode='bilinear')
    elif size[0] < h:
        image = F.interpolate(image, size, mode='area')
    if dim == 3:
        image = image.squeeze(1)
    return image


def crop(tensor, crop_size):
    size = tensor.size(2)   # assume h=w
    margin = (size - crop_size) // 2
    return tensor[:, :, margin:margin + crop_size, margin:margin + crop_size]


def get_mask_range(mask):
    h_range = torch.arange(0, mask.size(0))
    w_range = torch.arange(0, mask.size(1))
    grid = torch.stack(torch.meshgrid([h_range, w_range], indexing="ij"), 0).float()
    max_y = torch.max(grid[0, mask])
    min_y = torch.min(grid[0, mask])
    max_x = torch.max(grid[1, mask])
    min_x = torch.min(grid[1, mask])
    return max_y, min_y, max_x, min_x
