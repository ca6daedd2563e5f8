"""Image / latent dataset of GAN2Shape (GAN2Shape/dataset.py:8-79): `root/list.txt` names the image
files, `root/latents/<stem>.pt` holds each image's StyleGAN2 latent.  Same three classes, same item
layout `(image in [-1, 1] (3,H,W), latent (512,) or (n_latent,512), index)`.

Differences: `list.txt` is read with the csv module (no pandas); latents are loaded with
`torch.load(weights_only=True)` (tensor or dict-of-tensor files only, nothing is unpickled);
invalid subsets raise `IndexError` instead of exiting the process.  `default_transform` restates
main.py:98-103 (`transforms.Resize(size)` + `ToTensor()`) without torchvision."""
import csv
from os import path

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


def _read_list(root_dir, list_filename, subset):
    with open(path.join(root_dir, list_filename), newline='') as f:
        names = [row[0] for row in csv.reader(f) if row]
    if subset is not None:
        names = [names[i] for i in subset]  # IndexError on an invalid subset
    return names


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


class ImageDataset(Dataset):
    def __init__(self, root_dir, list_filename='list.txt', transform=None, subset=None):
        self.root_dir = root_dir
        self.transform = transform
        self.file_list = _read_list(root_dir, list_filename, subset)

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, index):
        with Image.open(path.join(self.root_dir, self.file_list[index])) as image:
            if self.transform is not None:
                image = self.transform(image)
            return image * 2 - 1


class LatentDataset(Dataset):
    def __init__(self, root_dir, list_filename='list.txt', latent_folder='latents', subset=None):
        self.root_dir = root_dir
        self.latent_folder = latent_folder
        self.file_list = _read_list(root_dir, list_filename, subset)

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, index):
        latent_file = self.file_list[index].split('.')[0] + '.pt'
        latent = torch.load(path.join(self.root_dir, self.latent_folder, latent_file),
                            map_location='cpu', weights_only=True)
        if isinstance(latent, dict):  # dataset.py:53-56: {'latent': w} or {name: {'latent': w}}
            if 'latent' not in latent:
                latent = latent.popitem()[1]
            latent = latent['latent']
        latent = latent.detach()
        return latent.squeeze(0) if latent.dim() == 2 else latent


class ImageLatentDataset(Dataset):
    def __init__(self, root_dir, list_filename='list.txt', transform=None, latent_folder='latents',
                 subset=None):
        self.image_dataset = ImageDataset(root_dir, list_filename, transform, subset)
        self.latent_dataset = LatentDataset(root_dir, list_filename, latent_folder, subset)
        assert len(self.image_dataset) == len(self.latent_dataset)

    def __len__(self):
        return len(self.image_dataset)

    def __getitem__(self, index):
        return self.image_dataset[index], self.latent_dataset[index], index
