"""On-disk layout of a GAN2Shape dataset (behaviour of GAN2Shape/dataset.py:8-79): `root/list.txt`
names the image files, one per line; `root/latents/<stem>.pt` holds each image's StyleGAN2 latent.
Items are `(image in [-1, 1] of shape (3,H,W), latent (512,) or (n_latent, 512), index)`.

Written for this package: `list.txt` is read with the csv module (no pandas); latents are loaded
with `torch.load(weights_only=True)` (tensor or dict-of-tensor files only — nothing is unpickled);
an invalid subset raises `IndexError` instead of exiting the process.  `default_transform` restates
main.py:98-103 (`transforms.Resize(size)` + `ToTensor()`) without torchvision."""
import csv
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


def default_transform(image_size):
    """PIL image -> float tensor (3,H,W) in [0,1]; the smaller edge is resized to `image_size`
    (bilinear), as torchvision's Resize(int) + ToTensor do."""
    def transform(image):
        w, h = image.size
        if w <= h:
            size = (image_size, max(1, int(image_size * h / w)))
        else:
            size = (max(1, int(image_size * w / h)), image_size)
        if (w, h) != size:
            image = image.resize(size, Image.BILINEAR)
        a = np.asarray(image.convert('RGB'), dtype=np.uint8)
        return torch.from_numpy(a.copy()).permute(2, 0, 1).float().div(255)
    return transform


class _ListedFiles(Dataset):
    """The entries of `root/list.txt` (first column), optionally restricted to `subset` indices."""

    def __init__(self, root_dir, list_filename, subset):
        self.root_dir = root_dir
        with open(os.path.join(root_dir, list_filename), newline='') as f:
            names = [row[0] for row in csv.reader(f) if row]
        self.file_list = names if subset is None else [names[i] for i in subset]

    def __len__(self):
        return len(self.file_list)


class ImageDataset(_ListedFiles):
    """Images scaled to [-1, 1] (the transform is expected to produce [0, 1])."""

    def __init__(self, root_dir, list_filename='list.txt', transform=None, subset=None):
        super().__init__(root_dir, list_filename, subset)
        self.transform = transform

    def __getitem__(self, index):
        with Image.open(os.path.join(self.root_dir, self.file_list[index])) as image:
            data = image if self.transform is None else self.transform(image)
            return data * 2 - 1


class LatentDataset(_ListedFiles):
    """`latents/<stem>.pt`: a tensor, {'latent': w}, or {name: {'latent': w}} (dataset.py:50-58)."""

    def __init__(self, root_dir, list_filename='list.txt', latent_folder='latents', subset=None):
        super().__init__(root_dir, list_filename, subset)
        self.latent_folder = latent_folder

    def __getitem__(self, index):
        stem = self.file_list[index].split('.')[0]
        obj = torch.load(os.path.join(self.root_dir, self.latent_folder, stem + '.pt'),
                         map_location='cpu', weights_only=True)
        if isinstance(obj, dict):
            inner = obj if 'latent' in obj else obj.popitem()[1]
            obj = inner['latent']
        latent = obj.detach()
        return latent.squeeze(0) if latent.dim() == 2 else latent   # (1, 512) -> (512,)


class ImageLatentDataset(Dataset):
    """(image, latent, index) triples over the same list."""

    def __init__(self, root_dir, list_filename='list.txt', transform=None, latent_folder='latents',
                 subset=None):
        self.image_dataset = ImageDataset(root_dir, list_filename, transform, subset)
        self.latent_dataset = LatentDataset(root_dir, list_filename, latent_folder, subset)
        assert len(self.image_dataset) == len(self.latent_dataset)

    def __len__(self):
        return len(self.image_dataset)

    def __getitem__(self, index):
        return self.image_dataset[index], self.latent_dataset[index], index
