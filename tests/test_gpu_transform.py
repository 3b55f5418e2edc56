"""GPU: the loader's image pipeline kernel (pcgan_image_transform, through the C-ABI) against the oracle's restatement of
the reference transform on Pillow -- BIT-EXACT (integer resampling, three correctly rounded fp32 operations), for
down- and up-scaling, crop-only, gray mix, flips, images of different sizes in one batch, every crop corner; the loader
end to end with and without --gpu_transform; full-size property checks at the UTKFace geometry (200 -> 143 -> 128)."""
import random
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import transform_ref as R

pytestmark = pytest.mark.gpu


class _O(object):
    def __init__(self, load, fine, transforms='resize_and_crop'):
        self.loadSize, self.fineSize, self.transforms = load, fine, transforms
        self.isTrain, self.no_flip = True, False


def _images(sizes, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i, (h, w) in enumerate(sizes):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i % 3 == 2:
            a = (a > 127).astype(np.uint8) * 255       # overshoot of the negative lobes must saturate like Pillow
        out.append(a)
    return out


def _run(dev, imgs, load, fine, aug, transforms='resize_and_crop', channels=3):
    from pcgan_amd.data.gpu_transform import GpuTransform
    tf = GpuTransform(_O(load, fine, transforms), dev)
    got = tf([torch.from_numpy(a) for a in imgs], aug, out_channels=channels)
    torch.cuda.synchronize()
    return got.cpu()


def _want(imgs, load, fine, aug, resize=True, channels=3):
    res = []
    for a, (x0, y0, flip) in zip(imgs, aug):
        t = R.transform(Image.fromarray(a), load, fine, x0, y0, flip, resize=resize)
        res.append(R.to_gray(t) if channels == 1 else t)
    return torch.stack(res)


@pytest.mark.parametrize('src,load,fine', [((200, 200), 143, 128), ((250, 250), 140, 128), ((64, 48), 143, 128), ((28, 28), 32, 32),
                                           ((300, 301), 36, 32), ((5, 7), 64, 64), ((128, 128), 128, 128), ((200, 200), 286, 256),
                                           ((37, 53), 47, 1)])
def test_kernel_matches_pillow_path(dev, src, load, fine):
    random.seed(load * 7 + fine)
    imgs = _images([src] * 5, seed=load)
    m = load - fine
    aug = [(0, 0, 0), (m, m, 1), (m, 0, 0), (0, m, 1)] + [(random.randint(0, m), random.randint(0, m), random.randint(0, 1))]
    assert torch.equal(_run(dev, imgs, load, fine, aug), _want(imgs, load, fine, aug))


@pytest.mark.parametrize('mode,src,load,fine', [('scale_width', (70, 50), 40, 32), ('scale_width', (120, 200), 160, 128),
                                                ('scale_width_and_crop', (70, 50), 40, 32), ('scale_width_and_crop', (200, 150), 160, 128),
                                                ('none', (70, 50), 40, 32), ('none', (53, 47), 40, 32), ('none', (48, 64), 40, 32)])
def test_scale_width_and_none_modes_match_pillow(dev, mode, src, load, fine):
    """the loader modes round 2 left out (reference data/base_dataset.py:33-40, 66-104) through the same integer resampler: bit-exact
    against the oracle on Pillow; `scale_width` / `none` have no crop (the output is the resized image, sides multiples of 4)"""
    from pcgan_amd.data.base_dataset import resize_plan
    o = _O(load, fine, mode)
    resized, fs, _ = resize_plan(o, src[1], src[0])
    w, h = resized if resized is not None else (src[1], src[0])
    random.seed(load + fine + len(mode))
    imgs = _images([src] * 4, seed=load + len(mode))
    mx, my = (w - fs, h - fs) if fs is not None else (0, 0)
    aug = [(0, 0, 0), (mx, my, 1), (random.randint(0, mx), random.randint(0, my), 0), (random.randint(0, mx), random.randint(0, my), 1)]
    got = _run(dev, imgs, load, fine, aug, transforms=mode)
    want = torch.stack([R.transform_mode(Image.fromarray(a), mode, load, fine, x0, y0, bool(fl)) for a, (x0, y0, fl) in zip(imgs, aug)])
    assert got.shape == want.shape and torch.equal(got, want)
    if fs is None:
        assert got.shape[2] % 4 == 0 and got.shape[3] % 4 == 0


def test_crop_only_and_gray(dev):
    imgs = _images([(40, 52)] * 4, seed=1)
    aug = [(0, 0, 0), (20, 8, 1), (5, 3, 0), (20, 8, 0)]
    assert torch.equal(_run(dev, imgs, None, 32, aug, transforms='crop'), _want(imgs, 0, 32, aug, resize=False))
    got = _run(dev, imgs, 36, 32, [(1, 2, 1)] * 4, channels=1)
    assert got.shape == (4, 1, 32, 32) and torch.equal(got, _want(imgs, 36, 32, [(1, 2, 1)] * 4, channels=1))


def test_ragged_batch_keeps_order(dev):
    sizes = [(50, 50), (44, 61), (50, 50), (200, 180), (44, 61), (33, 33), (50, 50)]
    imgs = _images(sizes, seed=2)
    random.seed(5)
    aug = [(random.randint(0, 8), random.randint(0, 8), random.randint(0, 1)) for _ in sizes]
    assert torch.equal(_run(dev, imgs, 40, 32, aug), _want(imgs, 40, 32, aug))


def test_rejects_bad_input(dev):
    from pcgan_amd.data.gpu_transform import GpuTransform
    tf = GpuTransform(_O(40, 32), dev)
    img = torch.zeros(50, 50, 3, dtype=torch.uint8)
    with pytest.raises(ValueError, match='crop offset'):
        tf([img], [(9, 0, 0)])
    with pytest.raises(ValueError, match='uint8'):
        tf([img.float()], [(0, 0, 0)])
    with pytest.raises(ValueError, match='smaller'):
        GpuTransform(_O(16, 32), dev)([img], [(0, 0, 0)])


def test_full_size_properties(dev):
    """UTKFace geometry, one training batch of pairs (64 images): constant images stay constant, a flip of the crop is the
    mirrored output, shifting the crop window shifts the output, and a sample of the batch equals the Pillow path"""
    n, load, fine = 64, 143, 128
    imgs = _images([(200, 200)] * n, seed=3)
    imgs[0][:] = 255
    imgs[1][:] = 0
    imgs[2][:] = 77
    random.seed(11)
    aug = [(random.randint(0, 15), random.randint(0, 15), 0) for _ in range(n)]
    base = _run(dev, imgs, load, fine, aug)
    assert torch.equal(base[0], torch.full((3, fine, fine), 1.0)) and torch.equal(base[1], torch.full((3, fine, fine), -1.0))
    assert torch.equal(base[2], torch.full((3, fine, fine), (np.float32(77) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)))
    flipped = _run(dev, imgs, load, fine, [(x, y, 1) for x, y, _ in aug])
    assert torch.equal(flipped, base.flip(3))
    shifted = _run(dev, imgs, load, fine, [(x - 1 if x else x, y, 0) for x, y, _ in aug])
    for i, (x, _, _) in enumerate(aug):
        if x:
            assert torch.equal(shifted[i][:, :, 1:], base[i][:, :, :-1])
    pick = list(range(0, n, 7))
    assert torch.equal(base[pick], _want([imgs[i] for i in pick], load, fine, [aug[i] for i in pick]))


def _opt(tmp_path, extra):
    from pcgan_amd.options.train_options import TrainOptions
    argv = ['train.py', '--dataroot', str(tmp_path), '--model', 'wsgan_emb', '--gpu_ids', '0', '--checkpoints_dir',
            str(tmp_path / 'ck'), '--sourcefile_A', str(tmp_path / 'pairs.txt'), '--loadSize', '40', '--fineSize', '32',
            '--nThreads', '0', '--batchSize', '4', '--serial_batches'] + list(extra)
    old, sys.argv = sys.argv, argv
    try:
        return TrainOptions().parse()
    finally:
        sys.argv = old


def test_loader_end_to_end(dev, tmp_path):
    from pcgan_amd.data import CreateDataLoader
    imgs = _images([(50, 50), (44, 61)] * 4, seed=4)
    for i, a in enumerate(imgs):
        Image.fromarray(a).save(tmp_path / ('img_%d.png' % i))
    with open(tmp_path / 'pairs.txt', 'w') as f:
        for i in range(8):
            f.write('img_%d.png img_%d.png %d\n' % (i, (i + 3) % 8, (0, 2)[i % 2]))
    random.seed(21)
    pil = list(CreateDataLoader(_opt(tmp_path, [])).load_data())
    random.seed(21)
    gpu = list(CreateDataLoader(_opt(tmp_path, ['--gpu_transform'])).load_data())
    assert len(pil) == len(gpu) == 2
    for a, b in zip(pil, gpu):
        assert sorted(a) == sorted(b)
        assert b['A'].is_cuda and torch.equal(a['A'], b['A'].cpu()) and torch.equal(a['B'], b['B'].cpu())
        assert torch.equal(a['label'], b['label']) and a['A_paths'] == b['A_paths'] and a['B_paths'] == b['B_paths']


def test_cycle_loader_end_to_end(dev, tmp_path):
    """the unpaired image + attribute dataset (wsgan_cycle) through both loader modes: identical batches"""
    from pcgan_amd.data import CreateDataLoader
    from pcgan_amd.options.train_options import TrainOptions
    imgs = _images([(50, 50)] * 6, seed=8)
    with open(tmp_path / 'a.txt', 'w') as fa, open(tmp_path / 'b.txt', 'w') as fb:
        for i, a in enumerate(imgs):
            Image.fromarray(a).save(tmp_path / ('%d_img.png' % (20 + i)))
            fa.write('%d_img.png\n' % (20 + i))
            fb.write('%d_img.png\n' % (20 + i))

    def opt(extra):
        argv = ['train.py', '--dataroot', str(tmp_path), '--model', 'wsgan_cycle', '--gpu_ids', '0', '--checkpoints_dir', str(tmp_path / 'ck'),
                '--sourcefile_A', str(tmp_path / 'a.txt'), '--sourcefile_B', str(tmp_path / 'b.txt'), '--loadSize', '40', '--fineSize', '32',
                '--nThreads', '0', '--batchSize', '3', '--which_model_netG', 'resnet_9blocks'] + extra
        old, sys.argv = sys.argv, argv
        try:
            return TrainOptions().parse()
        finally:
            sys.argv = old
    random.seed(33)
    torch.manual_seed(5)        # the DataLoader's shuffle
    pil = list(CreateDataLoader(opt([])).load_data())
    random.seed(33)
    torch.manual_seed(5)
    gpu = list(CreateDataLoader(opt(['--gpu_transform'])).load_data())
    assert len(pil) == len(gpu) == 2
    for a, b in zip(pil, gpu):
        assert sorted(a) == sorted(b) == ['A', 'A_paths', 'B_attr', 'B_paths']
        assert torch.equal(a['A'], b['A'].cpu()) and torch.equal(a['B_attr'], b['B_attr']) and a['A_paths'] == b['A_paths']
