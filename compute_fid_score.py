#!/usr/bin/env python
"""Frechet distance between two image sets (reference compute_fid_score.py).

    python compute_fid_score.py gen_dir real_dir --features alexnet --pretrained_model_path_IP alexnet.pth
    python compute_fid_score.py gen.txt real.txt --dataroot rootA rootB --features alexnet ...
    python compute_fid_score.py stats_a.npz stats_b.npz            # precomputed 'mu' + 'sigma', or raw 'act'

Each path is a directory of *.jpg / *.png, a .txt list of file names under the matching --dataroot (both as in the
reference), or an .npz of statistics.  The reference extracts pool3 features with torchvision's Inception-v3 and
weights it downloads (models/inception.py:60); neither exists offline, so this build cannot reproduce FID VALUES.  What
it provides is the metric (pcgan_amd/util/fid.py, held to the reference's own function by tests/test_fid.py) and a
feature extractor that runs here: `--features alexnet` = the AlexNet identity network of the training step on the HIP
path (conv5 features, spatially averaged, 256 dims).  Numbers from it are Frechet-AlexNet distances, not FID.
"""
import argparse
import os
import pathlib

import numpy as np


def list_images(path, dataroot):
    if path.endswith('.txt'):
        with open(path, 'r') as f:
            return [os.path.join(dataroot, line.rstrip('\n')) for line in f.readlines()]
    p = pathlib.Path(path)
    return sorted(str(x) for x in list(p.glob('*.jpg')) + list(p.glob('*.png')))


def load_images(files):
    """(n, 3, H, W) float32 in [0, 1] (compute_fid_score.py:208-220)"""
    from PIL import Image
    imgs = np.array([np.asarray(Image.open(fn).convert('RGB'), dtype=np.float32) for fn in files])
    return imgs.transpose((0, 3, 1, 2)) / 255.0


def alexnet_features(weights, device):
    import torch
    from pcgan_amd.models import networks
    net = networks.define_IP('alexnet', 3, [device.index])
    if weights:
        getattr(net, 'module', net).load_pretrained(weights)
    else:
        print('WARNING: --pretrained_model_path_IP not given: random AlexNet features')
    net.eval()
    norm = networks.Normalize((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010))
    from pcgan_amd.util.util import upsample2d
    return lambda batch: net(norm(upsample2d(batch.to(device) * 2 - 1, 224)))


def statistics_of(path, dataroot, model, batch_size):
    from pcgan_amd.util.fid import activation_statistics, get_activations
    if path.endswith('.npz'):
        z = np.load(path)
        return (z['mu'], z['sigma']) if 'mu' in z else activation_statistics(z['act'])
    if model is None:
        raise RuntimeError('image paths need a feature extractor: pass --features alexnet (Inception-v3 weights are not available offline)')
    import torch
    files = list_images(path, dataroot)
    if not files:
        raise RuntimeError('Invalid path: %s' % path)
    return activation_statistics(get_activations(torch.from_numpy(load_images(files)), model, batch_size))


def main():
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument('path', type=str, nargs=2, help='generated / real images: directory, .txt list or .npz statistics')
    ap.add_argument('--dataroot', type=str, nargs=2, default=['', ''])
    ap.add_argument('--batch-size', type=int, default=64)
    ap.add_argument('--features', choices=['none', 'alexnet'], default='none')
    ap.add_argument('--pretrained_model_path_IP', type=str, default='')
    ap.add_argument('-c', '--gpu', default='0', type=str)
    ap.add_argument('--result_path', type=str, default='')
    args = ap.parse_args()
    from pcgan_amd.util.fid import frechet_distance
    model = None
    if args.features == 'alexnet':
        import torch
        model = alexnet_features(args.pretrained_model_path_IP, torch.device('cuda:%d' % int(args.gpu)))
    m1, s1 = statistics_of(args.path[0], args.dataroot[0], model, args.batch_size)
    m2, s2 = statistics_of(args.path[1], args.dataroot[1], model, args.batch_size)
    value = frechet_distance(m1, s1, m2, s2)
    print('Frechet distance (%s features): %.6f' % (args.features if model else 'precomputed', value))
    if args.result_path:
        with open(args.result_path, 'a') as f:
            f.write('%s %s %.6f\n' % (args.path[0], args.path[1], value))
    return value


if __name__ == '__main__':
    main()
