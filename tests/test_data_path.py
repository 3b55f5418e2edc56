"""CPU: the loader side of the path -- pair-file parsing, dict layout, the PIL transform against
the oracle's restatement of torchvision-on-Pillow, and the host half of the GPU image pipeline: the fixed-point
coefficient tables must reproduce Pillow's `Image.resize(BICUBIC)` bit for bit when applied in integers."""
import os
import random
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import transform_ref as R


def _opt(tmp_path, extra=()):
    from pcgan_amd.options.train_options import TrainOptions
    argv = ['train.py', '--dataroot', str(tmp_path), '--model', 'wsgan_emb', '--gpu_ids', '-1', '--checkpoints_dir',
            str(tmp_path / 'ck'), '--sourcefile_A', str(tmp_path / 'pairs.txt'), '--loadSize', '40', '--fineSize', '32',
            '--nThreads', '0'] + list(extra)
    old, sys.argv = sys.argv, argv
    try:
        return TrainOptions().parse()
    finally:
        sys.argv = old


def _make_images(tmp_path, n=6, sizes=((50, 50),)):
    rng = np.random.default_rng(7)
    names = []
    for i in range(n):
        h, w = sizes[i % len(sizes)]
        name = 'img_%d.png' % i
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / name)
        names.append(name)
    with open(tmp_path / 'pairs.txt', 'w') as f:
        for i in range(n):
            f.write('%s %s %d\n' % (names[i], names[(i + 1) % n], (0, 2, 1)[i % 3]))
    return names


def _int_resize(arr, out_h, out_w):
    """the kernel's arithmetic in numpy: horizontal pass, uint8 rounding, vertical pass"""
    from pcgan_amd.data.gpu_transform import resample_table, PRECISION_BITS

    def one_axis(a, out, axis):
        k, b, _ = resample_table(a.shape[axis], out)
        a = np.moveaxis(a.astype(np.int64), axis, 0)
        res = np.zeros((out,) + a.shape[1:], dtype=np.int64)
        for o in range(out):
            x0, c = b[o]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(k[o, :c].astype(np.int64), a[x0:x0 + c], axes=(0, 0))
            res[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
        return np.moveaxis(res, 0, axis).astype(np.uint8)
    return one_axis(one_axis(arr, out_w, 1), out_h, 0)


@pytest.mark.parametrize('h,w,oh,ow', [(200, 200, 143, 143), (200, 200, 128, 128), (250, 250, 140, 140), (64, 48, 143, 143),
                                       (37, 53, 53, 37), (200, 200, 200, 100), (28, 28, 32, 32), (300, 301, 10, 11), (5, 5, 64, 64),
                                       (1, 1, 8, 8)])
def test_resample_tables_reproduce_pillow(h, w, oh, ow):
    rng = np.random.default_rng(h * 1000 + w)
    for kind in ('noise', 'extremes'):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if kind == 'extremes':                       # saturating overshoot of the negative bicubic lobes
            a = (a > 127).astype(np.uint8) * 255
        ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(_int_resize(a, oh, ow), ref)


def test_identity_table_when_size_is_kept():
    from pcgan_amd.data.gpu_transform import resample_table, PRECISION_BITS
    k, b, ks = resample_table(17, 17)
    assert ks == 1 and (k == 1 << PRECISION_BITS).all() and (b[:, 0] == np.arange(17)).all() and (b[:, 1] == 1).all()


def test_pil_transform_matches_oracle(tmp_path):
    from pcgan_amd.data.base_dataset import get_transform
    _make_images(tmp_path)
    opt = _opt(tmp_path)
    tf = get_transform(opt)
    img = Image.open(tmp_path / 'img_0.png').convert('RGB')
    for seed in range(6):
        random.seed(seed)
        got = tf(img)
        random.seed(seed)
        x0, y0 = random.randint(0, 8), random.randint(0, 8)
        flip = random.random() < 0.5
        assert torch.equal(got, R.transform(img, 40, 32, x0, y0, flip))
        assert got.shape == (3, 32, 32) and got.dtype == torch.float32 and -1.0 <= float(got.min()) and float(got.max()) <= 1.0


@pytest.mark.parametrize('mode,size', [('scale_width', (50, 70)), ('scale_width', (64, 32)), ('scale_width_and_crop', (70, 50)),
                                       ('scale_width_and_crop', (90, 40)), ('none', (50, 70)), ('none', (53, 47)), ('none', (48, 64)),
                                       ('crop', (50, 70))])
def test_every_non_affine_transform_mode_matches_the_oracle(tmp_path, mode, size):
    """f3 leftovers of round 2: `scale_width`, `scale_width_and_crop`, `none` (reference data/base_dataset.py:33-40, 66-104) next to
    `resize_and_crop` / `crop` -- the PIL path against the oracle's restatement on Pillow, draws replayed"""
    from pcgan_amd.data.base_dataset import get_transform, resize_plan
    _make_images(tmp_path, n=2, sizes=(size,))
    opt = _opt(tmp_path, ['--transforms', mode])
    tf = get_transform(opt)
    img = Image.open(tmp_path / 'img_0.png').convert('RGB')
    resized, fs, _ = resize_plan(opt, img.size[0], img.size[1])
    w, h = resized if resized is not None else img.size
    for seed in range(4):
        random.seed(seed)
        got = tf(img)
        random.seed(seed)
        x0 = random.randint(0, w - fs) if (fs is not None and w > fs) else 0
        y0 = random.randint(0, h - fs) if (fs is not None and h > fs) else 0
        flip = random.random() < 0.5
        assert torch.equal(got, R.transform_mode(img, mode, 40, 32, x0, y0, flip))
        if fs is None:
            assert got.shape[1] % 4 == 0 and got.shape[2] % 4 == 0        # "the size needs to be a multiple of 4"
        else:
            assert got.shape == (3, 32, 32)


def test_affine_modes_and_color_jitter(tmp_path):
    """resize_affine_crop / resize_affine_center (reference :41-52) run on the PIL path: identity parameters leave the resized image
    unchanged, a real draw rotates / scales about the centre with fill 127, the centre mode draws no crop offsets; --use_color_jitter
    is torchvision's ColorJitter() with default (zero) arguments, i.e. the identity (:58-59)"""
    from pcgan_amd.data import base_dataset as B
    _make_images(tmp_path, n=2, sizes=((50, 50),))
    img = Image.open(tmp_path / 'img_0.png').convert('RGB').resize((40, 40), Image.BICUBIC)
    # identity parameters: the inverse matrix is the identity and Pillow's bicubic transform at integer positions returns the image
    m = B._inverse_affine_matrix((20.5, 20.5), 0.0, (0, 0), 1.0, 0.0)
    assert np.allclose(m, [1, 0, 0, 0, 1, 0], atol=1e-12)
    opt = _opt(tmp_path, ['--transforms', 'resize_affine_center', '--affineDegrees', '0', '--affineScale', '1.0', '1.0', '--no_flip'])
    src = Image.open(tmp_path / 'img_0.png').convert('RGB')
    got = B.get_transform(opt)(src)
    assert torch.equal(got, R.transform(src, 40, 32, 4, 4, False))        # CenterCrop: round((40 - 32) / 2) = 4
    # a real draw: rotation by 90 degrees about the centre maps the image onto its rotation (up to the half-pixel centre convention)
    opt2 = _opt(tmp_path, ['--transforms', 'resize_affine_crop', '--use_color_jitter'])
    random.seed(3)
    out = B.get_transform(opt2)(src)
    assert out.shape == (3, 32, 32) and torch.isfinite(out).all() and -1.0 <= float(out.min()) and float(out.max()) <= 1.0
    random.seed(3)
    angle = random.uniform(-5, 5)
    scale = random.uniform(0.95, 1.05)
    assert -5 <= angle <= 5 and 0.95 <= scale <= 1.05
    with pytest.raises(ValueError, match='not a valid option'):
        B.get_transform(_opt(tmp_path, ['--transforms', 'bogus']))
    from pcgan_amd.data.gpu_transform import GpuTransform
    with pytest.raises(NotImplementedError, match='affine'):
        GpuTransform(opt2, 'cuda:0')


def test_crop_larger_than_image_is_refused(tmp_path):
    from pcgan_amd.data.base_dataset import get_transform
    _make_images(tmp_path)
    tf = get_transform(_opt(tmp_path, ['--transforms', 'crop', '--fineSize', '64']))
    with pytest.raises(ValueError, match='larger than input image'):
        tf(Image.open(tmp_path / 'img_0.png').convert('RGB'))


def test_pair_dataset_layout(tmp_path):
    from pcgan_amd.data import CreateDataLoader
    names = _make_images(tmp_path)
    opt = _opt(tmp_path, ['--batchSize', '3', '--serial_batches', '--no_flip'])
    loader = CreateDataLoader(opt).load_data()
    batches = list(loader)
    assert len(batches) == 2
    seen = set()
    for b in batches:
        assert sorted(b) == ['A', 'A_paths', 'B', 'B_paths', 'label']
        assert b['A'].shape == (3, 3, 32, 32) and b['B'].shape == (3, 3, 32, 32) and b['label'].dtype == torch.int64
        for pa, pb, lab in zip(b['A_paths'], b['B_paths'], b['label']):
            i = names.index(os.path.basename(pa))
            assert os.path.basename(pb) == names[(i + 1) % 6] and int(lab) == (0, 2, 1)[i % 3]
            seen.add(i)
    assert seen == set(range(6))


def test_pair_dataset_per_label_variant_and_gray(tmp_path):
    from pcgan_amd.data import create_dataset
    _make_images(tmp_path)
    opt = _opt(tmp_path, ['--no_mixed_label_D', '--input_nc', '1', '--output_nc', '1', '--loadSize', '32', '--no_flip'])
    ds = create_dataset(opt)
    item = ds[0]
    assert sorted(item) == sorted(['%d_%s' % (L, k) for L in (0, 1, 2) for k in ('A', 'B', 'A_paths', 'B_paths')])
    for L in (0, 1, 2):         # loadSize == fineSize, no flip: nothing is drawn
        for side in 'AB':
            img = Image.open(item['%d_%s_paths' % (L, side)]).convert('RGB')
            want = R.to_gray(R.transform(img, 32, 32, 0, 0, False))
            assert item['%d_%s' % (L, side)].shape == (1, 32, 32) and torch.equal(item['%d_%s' % (L, side)], want)


def test_raw_mode_hands_over_the_same_draws(tmp_path):
    """--gpu_transform: a worker returns the decoded bytes and the draws; replaying them through the integer pipeline
    (numpy stand-in for the kernel) gives the PIL path's tensor bit for bit"""
    from pcgan_amd.data import create_dataset
    _make_images(tmp_path, sizes=((50, 50), (44, 61)))
    pil = create_dataset(_opt(tmp_path))
    raw = create_dataset(_opt(tmp_path, ['--gpu_transform']))
    for index in range(4):
        random.seed(100 + index)
        want = pil[index]
        random.seed(100 + index)
        got = raw[index]
        assert sorted(got) == ['A_aug', 'A_paths', 'A_raw', 'B_aug', 'B_paths', 'B_raw', 'label']
        for side in 'AB':
            img, (x0, y0, flip) = got[side + '_raw'].numpy(), [int(v) for v in got[side + '_aug']]
            res = _int_resize(img, 40, 40)[y0:y0 + 32, x0:x0 + 32]
            res = res[:, ::-1] if flip else res
            t = torch.from_numpy(np.ascontiguousarray(res.transpose(2, 0, 1))).to(torch.float32).div(255).sub(0.5).div(0.5)
            assert torch.equal(t, want[side])
        assert got['label'] == want['label'] and got['A_paths'] == want['A_paths']


def test_gpu_transform_refuses_cpu(tmp_path):
    from pcgan_amd.data.gpu_transform import GpuTransform
    _make_images(tmp_path)
    with pytest.raises(RuntimeError, match='no fallback'):
        GpuTransform(_opt(tmp_path, ['--gpu_transform']), 'cpu')


def test_image_transform_argument_checks():
    import ctypes
    from pcgan_amd.hip import lib
    h = lib.load()
    d = lib.ImageDesc(50, 50, 40, 40, 48, 48, 5, 5, 3)          # crop larger than the resized image
    band, rows = ctypes.c_int(0), ctypes.c_int(0)
    bv = np.zeros((40, 2), dtype=np.int32)
    assert h.pcgan_image_transform_band(ctypes.byref(d), bv.ctypes.data_as(ctypes.c_void_p), ctypes.byref(band), ctypes.byref(rows)) != 0
    assert b'crop' in h.pcgan_last_error()
    from pcgan_amd.data.gpu_transform import resample_table
    _, bv, _ = resample_table(200, 143)
    d = lib.ImageDesc(200, 200, 143, 143, 128, 128, 7, 7, 3)
    bv = np.ascontiguousarray(bv)
    assert h.pcgan_image_transform_band(ctypes.byref(d), bv.ctypes.data_as(ctypes.c_void_p), ctypes.byref(band), ctypes.byref(rows)) == 0
    assert band.value == 16 and 16 * 200 / 143 <= rows.value <= 16 * 200 / 143 + 8
