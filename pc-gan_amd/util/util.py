"""Host utilities of the hot path (reference util/util.py), written fresh.

Tensor helpers route through the HIP functional layer; the integer helpers
are plain Python and must be bit-exact with the reference.
"""
import argparse
import ast
import os

import numpy as np
import torch

from ..hip import functional as HF


def mkdirs(paths):
    for p in ([paths] if isinstance(paths, str) else list(paths)):
        os.makedirs(p, exist_ok=True)


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def get_attr_value(fname, dlm='_'):
    """reference util/util.py:71-73"""
    return float(fname.split(dlm)[0])


def get_attr_label(attr, bins):
    """reference util/util.py:76-81: index of the first bin [bins[L], bins[L+1]) containing attr.
    Quirks kept: no match (attr < bins[0], NaN) falls through to the LAST bin index
    len(bins)-2; fewer than two bin edges gives None."""
    label = None
    for label in range(len(bins) - 1):
        if bins[label] <= attr < bins[label + 1]:
            break
    return label


def str2list(text):
    """reference util/util.py:84-93: python-literal list or a .npy/.npz path."""
    assert isinstance(text, str)
    text = text.strip()
    if text.endswith(('.npy', '.npz')):
        return np.load(text)
    assert text.startswith('[') and text.endswith(']')
    return ast.literal_eval(text)


def str2bool(v):
    """reference util/util.py:96-108"""
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def upsample2d(t, size):
    """reference util/util.py:111-117 on the HIP bilinear kernel."""
    return HF.upsample2d(t, size)


_inject = {'eps': None}


def inject_noise(iterator):
    """Parity-test hook: tensors to use instead of torch.randn_like, consumed in call order."""
    _inject['eps'] = iterator


def _randn_like(t):
    if _inject['eps'] is not None:
        return next(_inject['eps']).to(device=t.device, dtype=t.dtype).view_as(t)
    return torch.randn_like(t)


def reparameterize(mu, logvar):
    """reference util/util.py:130-133: mu + eps * exp(logvar / 2), one randn_like per call"""
    std = torch.exp(0.5 * logvar)
    return mu + _randn_like(std) * std


def resample(mu=0., var=0.):
    """reference util/util.py:136-139 (tiny (B,1,1,1) tensors: host-side scalar plumbing)."""
    std = torch.sqrt(var)
    return mu + _randn_like(std) * std


def compute_mu_and_var(E, x, T, noisy=False):
    """reference util/util.py:153-171: T stochastic passes of E, running mean and mean of squares."""
    y_mu, y_sq, s2_mu = 0., 0., 0.
    for _ in range(T):
        if noisy:
            y, logs2 = E(x)
            s2_mu = s2_mu + 1. / T * torch.exp(logs2)
        else:
            y = E(x)
        y_mu = y_mu + 1. / T * y
        y_sq = y_sq + 1. / T * y ** 2
    y_var = y_sq - y_mu ** 2
    return (y_mu, y_var, s2_mu) if noisy else (y_mu, y_var)


def tensor2im(image, imtype=np.uint8):
    """reference util/util.py:14-28: first image of the batch, [-1,1] -> [0,255] HWC."""
    if not isinstance(image, torch.Tensor):
        return image
    t = image.detach()
    arr = (t[0] if t.dim() == 4 else t).cpu().float().numpy()
    if arr.shape[0] == 1:
        arr = np.tile(arr, (3, 1, 1))
    return ((np.transpose(arr, (1, 2, 0)) + 1) / 2.0 * 255.0).astype(imtype)


def save_image(image_numpy, image_path):
    """reference util/util.py:43-45"""
    from PIL import Image
    Image.fromarray(image_numpy).save(image_path)
