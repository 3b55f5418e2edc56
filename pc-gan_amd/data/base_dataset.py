"""reference data/base_dataset.py:7-64, without torchvision: the resize(bicubic) -> random
crop -> random flip -> [0,1] -> (x-0.5)/0.5 pipeline is restated on PIL + numpy."""
import random

import numpy as np
import torch
import torch.utils.data as data


class BaseDataset(data.Dataset):
    def name(self):
        return 'BaseDataset'

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def initialize(self, opt):
        pass

    def __len__(self):
        return 0


MULT = 4      # reference data/base_dataset.py:70-73, 91-95: sizes are made multiples of 4 ("going through generator network may change img size")
TRANSFORM_MODES = ('resize_and_crop', 'crop', 'scale_width', 'scale_width_and_crop', 'none', 'resize_affine_crop', 'resize_affine_center')


def _up4(v):
    return ((v - 1) // MULT + 1) * MULT


def scale_width_size(w, h, target_width):
    """reference __scale_width (data/base_dataset.py:87-104): (w, h) after the scaling, or None when the image is kept"""
    assert target_width % MULT == 0, 'the target width needs to be multiple of %d.' % MULT
    if w == target_width and h % MULT == 0:
        return None
    return target_width, _up4(int(target_width * h / w))


def adjust_size(w, h):
    """reference __adjust (data/base_dataset.py:66-84): width and height rounded up to multiples of 4, or None"""
    if w % MULT == 0 and h % MULT == 0:
        return None
    return _up4(w), _up4(h)


def resize_plan(opt, w, h):
    """What `--transforms` does to a (w, h) image before flip / ToTensor / Normalize (reference data/base_dataset.py:24-64):
    returns (resized, crop, centre) -- `resized` = (RW, RH) of the bicubic resize or None if Pillow's resize is skipped, `crop` =
    the side of the square crop or None, `centre` = True for a centre crop (resize_affine_center) instead of a random one."""
    m = opt.transforms
    if m in ('resize_and_crop', 'resize_affine_crop'):
        return (opt.loadSize, opt.loadSize), opt.fineSize, False
    if m == 'resize_affine_center':
        return (opt.loadSize, opt.loadSize), opt.fineSize, True
    if m == 'crop':
        return None, opt.fineSize, False
    if m == 'scale_width':
        return scale_width_size(w, h, opt.fineSize), None, False
    if m == 'scale_width_and_crop':
        return scale_width_size(w, h, opt.loadSize), opt.fineSize, False
    if m == 'none':
        return adjust_size(w, h), None, False
    raise ValueError('--resize_or_crop %s is not a valid option.' % m)      # (the reference's message, :49)


def _inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision.transforms.functional._get_inverse_affine_matrix as published with the torchvision releases of the reference's
    era (0.2 - 0.4; RandomAffine without translate / shear here): inverse of T * C * RSS * C^-1 for PIL's Image.transform."""
    import math
    angle, shear = math.radians(angle), math.radians(shear)
    scale = 1.0 / scale
    d = math.cos(angle + shear) * math.cos(angle) + math.sin(angle + shear) * math.sin(angle)
    matrix = [math.cos(angle + shear), math.sin(angle + shear), 0, -math.sin(angle), math.cos(angle), 0]
    matrix = [scale / d * v for v in matrix]
    matrix[2] += matrix[0] * (-center[0] - translate[0]) + matrix[1] * (-center[1] - translate[1])
    matrix[5] += matrix[3] * (-center[0] - translate[0]) + matrix[4] * (-center[1] - translate[1])
    matrix[2] += center[0]
    matrix[5] += center[1]
    return matrix


def random_affine(img, degrees, scale_range):
    """transforms.RandomAffine(degrees, scale=scale_range, resample=BICUBIC, fillcolor=127) of the reference's two affine modes
    (data/base_dataset.py:41-52): angle ~ U(-degrees, degrees), scale ~ U(scale_range), no translation / shear, about the image
    centre.  Draws from `random` (angle first, then scale).  torchvision is absent here: this restates its published algorithm and
    is NOT pinned by a golden vector."""
    from PIL import Image
    angle = random.uniform(-degrees, degrees)
    scale = random.uniform(scale_range[0], scale_range[1])
    w, h = img.size
    matrix = _inverse_affine_matrix((w * 0.5 + 0.5, h * 0.5 + 0.5), angle, (0, 0), scale, 0.0)
    return img.transform((w, h), Image.AFFINE, matrix, Image.BICUBIC, fillcolor=127)


def get_transform(opt):
    """PIL.Image -> float tensor (3, H, W) in [-1, 1]: every `--transforms` mode of the reference (data/base_dataset.py:24-64).
    `--use_color_jitter` adds transforms.ColorJitter() with its default arguments (brightness = contrast = saturation = hue = 0):
    the identity, so the flag is accepted and changes nothing -- as in the reference."""
    from PIL import Image
    if opt.transforms not in TRANSFORM_MODES:
        raise ValueError('--resize_or_crop %s is not a valid option.' % opt.transforms)

    def tf(img):
        w, h = img.size
        resized, fs, centre = resize_plan(opt, w, h)
        if resized is not None:
            img = img.resize(resized, Image.BICUBIC)
        if opt.transforms in ('resize_affine_crop', 'resize_affine_center'):
            img = random_affine(img, opt.affineDegrees, tuple(opt.affineScale))
        if fs is not None:
            w, h = img.size
            if w < fs or h < fs:    # torchvision's RandomCrop refuses too (no pad_if_needed in the reference's pipeline)
                raise ValueError('Required crop size %s is larger than input image size %s' % ((fs, fs), (h, w)))
            if centre:              # transforms.CenterCrop: round((size - crop) / 2)
                x0, y0 = int(round((w - fs) / 2.0)), int(round((h - fs) / 2.0))
            else:
                x0 = random.randint(0, w - fs) if w > fs else 0
                y0 = random.randint(0, h - fs) if h > fs else 0
            img = img.crop((x0, y0, x0 + fs, y0 + fs))
        if opt.isTrain and not opt.no_flip and random.random() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        arr = np.asarray(img, dtype=np.float32) / 255.0
        t = torch.from_numpy(arr.transpose(2, 0, 1).copy())
        return (t - 0.5) / 0.5
    return tf


def decode_raw(img, opt):
    """--gpu_transform: what a worker hands to the loader instead of the transformed tensor -- the decoded image as a
    uint8 (H, W, 3) tensor and the (x0, y0, flip) draws of get_transform, taken from `random` in the same order."""
    from .gpu_transform import draw_augmentation
    if opt.transforms in ('resize_affine_crop', 'resize_affine_center'):
        raise NotImplementedError('pcgan_amd: --gpu_transform does not cover the affine modes (--transforms %s); the loader\'s PIL '
                                  'path (no --gpu_transform) does' % opt.transforms)
    w, h = img.size
    resized, fs, _ = resize_plan(opt, w, h)
    if resized is not None:
        w, h = resized
    aug = draw_augmentation(w, h, fs, opt.isTrain and not opt.no_flip)
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()), torch.tensor(aug, dtype=torch.int32)
