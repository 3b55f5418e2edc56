"""The loader's image pipeline on the GPU: decoded uint8 RGB images in, the fp32 NCHW batch the
model's set_input reads out -- `Resize([loadSize, loadSize], BICUBIC) -> RandomCrop(fineSize) -> RandomHorizontalFlip
-> ToTensor -> Normalize(.5, .5)` of the reference's get_transform (data/base_dataset.py:24-64) in ONE launch of
`pcgan_image_transform` per batch of equally sized images (include/pcgan_hip.h).

torchvision's Resize on a PIL image is Pillow's `Image.resize`; its resampling is a separable two-pass filter in 22-bit
fixed point.  `resample_table` rebuilds Pillow's coefficient tables (libImaging/Resample.c: bicubic_filter,
precompute_coeffs, normalize_coeffs_8bpc) in the same double arithmetic, the kernel applies them in integers, so the
batch equals the PIL path bit for bit (tests/test_gpu_transform.py holds it to `Image.resize` itself).

JPEG/PNG decoding stays in the DataLoader workers; what moves to the GPU is the per-pixel work (4.7 ms per image and
core with PIL at 200x200 -> 143x143 -> 128x128 on the MI355X box's host, scripts/bench_transform.py).
"""
import ctypes
import math
import random

import numpy as np
import torch

from ..hip import lib as _L

PRECISION_BITS = 32 - 8 - 2     # Resample.c


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_table(in_size, out_size):
    """(coefficients int32 [out][ksize], bounds int32 [out][2], ksize) of Pillow's bicubic resampling of a whole
    axis from in_size to out_size; the identity table when the size does not change (Pillow skips that pass)."""
    if in_size == out_size:
        k = np.full((out_size, 1), 1 << PRECISION_BITS, dtype=np.int32)
        b = np.stack([np.arange(out_size, dtype=np.int32), np.ones(out_size, dtype=np.int32)], axis=1)
        return k, np.ascontiguousarray(b), 1
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
    return kk, bounds, ksize


def draw_augmentation(w, h, fine, flip_enabled):
    """crop offsets and flip of ONE image, drawn from `random` in the order the PIL path draws them (fine = None: no crop,
    nothing drawn for it)"""
    x0 = random.randint(0, w - fine) if (fine is not None and w > fine) else 0
    y0 = random.randint(0, h - fine) if (fine is not None and h > fine) else 0
    flip = 1 if (flip_enabled and random.random() < 0.5) else 0
    return x0, y0, flip


class _Geometry(object):
    """device-resident tables of one (source size -> resized size -> crop size)"""

    def __init__(self, H, W, RH, RW, FH, FW, out_channels, device):
        kh, bh, ksh = resample_table(W, RW)
        kv, bv, ksv = resample_table(H, RH)
        self.RH, self.RW, self.FH, self.FW = RH, RW, FH, FW
        self.desc = _L.ImageDesc(H, W, RH, RW, FH, FW, ksh, ksv, out_channels)
        band, rows = ctypes.c_int(0), ctypes.c_int(0)
        bv = np.ascontiguousarray(bv)
        _L.check(_L.load().pcgan_image_transform_band(ctypes.byref(self.desc), bv.ctypes.data_as(ctypes.c_void_p),
                                                      ctypes.byref(band), ctypes.byref(rows)), 'image_transform_band')
        self.band, self.rows = band.value, rows.value
        self.kh, self.bh, self.kv, self.bv = (torch.from_numpy(np.ascontiguousarray(t)).to(device) for t in (kh, bh, kv, bv))


class GpuTransform(object):
    """callable: list of uint8 (H, W, 3) tensors + per-image (x0, y0, flip) -> float (n, C, fine, fine) on `device`"""

    def __init__(self, opt, device):
        if opt.transforms not in ('resize_and_crop', 'crop', 'scale_width', 'scale_width_and_crop', 'none'):
            raise NotImplementedError('pcgan_amd: the GPU image pipeline does not cover --transforms %s (the affine modes run on the '
                                      'loader\'s PIL path: drop --gpu_transform)' % opt.transforms)
        self.opt = opt
        self.fine = opt.fineSize
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('pcgan_amd: the GPU image pipeline needs a GPU device (got %s); there is no fallback' % device)
        self._geo = {}
        self._stage = {}        # (H, W) -> reusable host buffer uint8 (cap, H, W, 3)

    def geometry(self, H, W, out_channels):
        """resized size and output size of an H x W source under the loader's --transforms mode (base_dataset.resize_plan)"""
        key = (H, W, out_channels)
        if key not in self._geo:
            from .base_dataset import resize_plan
            resized, fs, _ = resize_plan(self.opt, W, H)
            RW, RH = resized if resized is not None else (W, H)
            FH, FW = (fs, fs) if fs is not None else (RH, RW)
            if RH < FH or RW < FW:
                raise ValueError('image %dx%d (resized %dx%d) is smaller than --fineSize %d' % (H, W, RH, RW, self.fine))
            self._geo[key] = _Geometry(H, W, RH, RW, FH, FW, out_channels, self.device)
        return self._geo[key]

    def _upload(self, images, H, W):
        """decoded bytes -> one reusable host buffer per source size (numpy row copies: torch's copy_ would fork its
        whole intra-op thread pool for every 120 KB image, 0.24 ms each on a 128-thread host) -> one copy to the GPU
        (measured on the MI355X box: 0.2 ms + 0.15 ms for 64 images of 200x200, scripts/diag/diag_h2d.py)"""
        n = len(images)
        buf = self._stage.get((H, W))
        if buf is None or buf.shape[0] < n:
            buf = self._stage[(H, W)] = np.empty((n, H, W, 3), dtype=np.uint8)
        for j, im in enumerate(images):
            buf[j] = im.numpy()
        return torch.from_numpy(buf[:n]).to(self.device)     # blocking copy: the buffer is free again on return

    def __call__(self, images, aug, out_channels=3):
        n = len(images)
        aug = torch.as_tensor(aug, dtype=torch.int32).reshape(n, 3)
        groups = {}
        for i, im in enumerate(images):
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
                raise ValueError('GpuTransform: image %d is not a uint8 (H, W, 3) tensor' % i)
            groups.setdefault((int(im.shape[0]), int(im.shape[1])), []).append(i)
        # one output size per batch: the crop size, or (scale_width / none: no crop) the common size of the resized images
        sizes = {(g.FH, g.FW) for g in (self.geometry(H, W, out_channels) for (H, W) in groups)}
        if len(sizes) != 1:
            raise ValueError('GpuTransform: --transforms %s leaves images of different sizes %s in one batch (the reference\'s '
                             'DataLoader cannot stack them either)' % (self.opt.transforms, sorted(sizes)))
        FH, FW = next(iter(sizes))
        out = torch.empty((n, out_channels, FH, FW), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for (H, W), idx in groups.items():
            g = self.geometry(H, W, out_channels)
            RH, RW = g.RH, g.RW
            a = torch.empty((len(idx), 4), dtype=torch.int32)
            a[:, :3] = aug[idx]
            a[:, 3] = torch.as_tensor(idx, dtype=torch.int32)
            # the kernel trusts the offsets: check them where they are still host data
            if int(a[:, 0].min()) < 0 or int(a[:, 0].max()) > RW - FW or int(a[:, 1].min()) < 0 \
                    or int(a[:, 1].max()) > RH - FH:
                raise ValueError('GpuTransform: crop offset outside the resized image')
            src = self._upload([images[i] for i in idx], H, W)
            a_dev = a.to(self.device, non_blocking=True)
            _L.check(_L.load().pcgan_image_transform(
                ctypes.byref(g.desc), src.data_ptr(), g.kh.data_ptr(), g.bh.data_ptr(), g.kv.data_ptr(), g.bv.data_ptr(),
                a_dev.data_ptr(), out.data_ptr(), len(idx), g.band, g.rows, stream), 'image_transform')
            a_dev.record_stream(torch.cuda.current_stream(self.device))
        from ..hip import ops
        ops.mark_ready(out)      # producer's event: the step's ahead-of-time passes over this batch wait for THIS, not for the main stream
        return out
