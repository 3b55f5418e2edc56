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


def get_transform(opt):
    """PIL.Image -> float tensor (3,fineSize,fineSize) in [-1,1]; `resize_and_crop` and `crop`
    of the reference (the other modes raise)."""
    from PIL import Image

    def tf(img):
        if opt.transforms == 'resize_and_crop':
            img = img.resize((opt.loadSize, opt.loadSize), Image.BICUBIC)
        elif opt.transforms != 'crop':
            raise NotImplementedError('pcgan_amd: --transforms %s is outside the hot path' % opt.transforms)
        w, h = img.size
        fs = opt.fineSize
        if w < fs or h < fs:    # torchvision's RandomCrop refuses too (no pad_if_needed in the reference's pipeline)
            raise ValueError('Required crop size %s is larger than input image size %s' % ((fs, fs), (h, w)))
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
    if opt.transforms == 'resize_and_crop':
        w = h = opt.loadSize
    elif opt.transforms == 'crop':
        w, h = img.size
    else:
        raise NotImplementedError('pcgan_amd: --transforms %s is outside the hot path' % opt.transforms)
    aug = draw_augmentation(w, h, opt.fineSize, opt.isTrain and not opt.no_flip)
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()), torch.tensor(aug, dtype=torch.int32)
