"""ORACLE (test infrastructure, not product code).

The reference's loader transform (data/base_dataset.py:24-64, `resize_and_crop` / `crop`) and the gray mix of its pair
dataset (data/wsgan_emb_dataset.py:46-49), restated with the third-party code the reference itself runs: torchvision's
`Resize(osize, Image.BICUBIC)` on a PIL image IS Pillow's `Image.resize` (torchvision is absent from this image, Pillow
12.2 is here), `ToTensor` is `uint8 -> float32, .div(255)` on the CHW permutation, `Normalize` is `.sub_(mean).div_(std)`.
The random draws (crop offsets, flip) are arguments, so a caller can replay the product's draws.
Only tests/ may import this module.
"""
import numpy as np
import torch
from PIL import Image


def transform(img, load_size, fine_size, x0, y0, flip, resize=True):
    """PIL RGB image -> float32 (3, fine, fine) in [-1, 1]"""
    if resize:
        img = img.resize((load_size, load_size), Image.BICUBIC)     # transforms.Resize([loadSize, loadSize], BICUBIC)
    img = img.crop((x0, y0, x0 + fine_size, y0 + fine_size))        # transforms.RandomCrop(fineSize) at (y0, x0)
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)                  # transforms.RandomHorizontalFlip
    t = torch.from_numpy(np.array(img, np.uint8, copy=True)).permute(2, 0, 1).contiguous()
    t = t.to(dtype=torch.float32).div(255)                          # transforms.ToTensor
    mean = torch.tensor([0.5, 0.5, 0.5]).view(3, 1, 1)
    std = torch.tensor([0.5, 0.5, 0.5]).view(3, 1, 1)
    return t.sub_(mean).div_(std)                                   # transforms.Normalize


def to_gray(t):
    """data/wsgan_emb_dataset.py:46-49"""
    return (t[0, ...] * 0.299 + t[1, ...] * 0.587 + t[2, ...] * 0.114).unsqueeze(0)


def scale_width(img, target_width):
    """reference data/base_dataset.py:87-104 (__scale_width), on Pillow itself"""
    ow, oh = img.size
    mult = 4
    assert target_width % mult == 0, "the target width needs to be multiple of %d." % mult
    if ow == target_width and oh % mult == 0:
        return img
    w = target_width
    target_height = int(target_width * oh / ow)
    m = (target_height - 1) // mult
    h = (m + 1) * mult
    return img.resize((w, h), Image.BICUBIC)


def adjust(img):
    """reference data/base_dataset.py:66-84 (__adjust): width and height up to multiples of 4"""
    ow, oh = img.size
    mult = 4
    if ow % mult == 0 and oh % mult == 0:
        return img
    w = ((ow - 1) // mult + 1) * mult
    h = ((oh - 1) // mult + 1) * mult
    return img.resize((w, h), Image.BICUBIC)


def transform_mode(img, mode, load_size, fine_size, x0=0, y0=0, flip=False):
    """the non-affine `--transforms` modes of get_transform (reference data/base_dataset.py:24-40, 54-64) with the random draws as
    arguments: PIL RGB image -> float32 (3, H, W) in [-1, 1]"""
    crop = True
    if mode == 'resize_and_crop':
        img = img.resize((load_size, load_size), Image.BICUBIC)
    elif mode == 'crop':
        pass
    elif mode == 'scale_width':
        img, crop = scale_width(img, fine_size), False
    elif mode == 'scale_width_and_crop':
        img = scale_width(img, load_size)
    elif mode == 'none':
        img, crop = adjust(img), False
    else:
        raise ValueError(mode)
    if crop:
        img = img.crop((x0, y0, x0 + fine_size, y0 + fine_size))
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    t = torch.from_numpy(np.array(img, np.uint8, copy=True)).permute(2, 0, 1).contiguous().to(dtype=torch.float32).div(255)
    return t.sub_(torch.tensor([0.5, 0.5, 0.5]).view(3, 1, 1)).div_(torch.tensor([0.5, 0.5, 0.5]).view(3, 1, 1))
