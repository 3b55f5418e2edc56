#!/usr/bin/env python
"""Elo-rating encoder trainer on the HIP path (reference siamese.py:41-154, 422-769, `--mode train`).

Trains `networks.SiameseNetwork` (ResNet-18 trunk + 3x3 conv head, global pooling) on image pairs with a three-way
label (0: first < second, 1: draw, 2: first > second) with the draw-aware binary NLL on sigmoid(rating1 - rating2) and
Adam; writes `init_net.pth`, `latest_net.pth`, `<epoch>_net.pth` and `loss.txt` under <checkpoint_dir>/<name>/ -- the
`<epoch>_net.pth` file is what `train.py --model wsgan_emb --pretrained_model_path_E` loads.

Same flag names and defaults as the reference for everything implemented: the deterministic recipe, MC dropout (`--bayesian true
--bnn_dropout p --T_train T`) and the noisy-rating variants (`--noisy true` with `--rsample`, `--lb_or_mc`, `--M`, a second Adam on
the log-variance head with `--lr_sigma` and its epoch schedule); no `fc_dim`, no `use_cxn`, no `--finetune_fc_only`; image
augmentation is resize + random crop + flip (torchvision's affine /
colour jitter are not available here); `--dataroot synthetic` trains on seeded synthetic pairs whose label is decided
by a hidden per-image score, so the loss must fall.

    python siamese.py --dataroot synthetic --name elo --batch_size 32 --num_epochs 2 --pretrained_model_path ''
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 siamese.py ...        # one process per GPU, RCCL
"""
import argparse
import math
import os

import numpy as np
import torch

from pcgan_amd.data.base_dataset import get_transform
from pcgan_amd.hip import parallel
from pcgan_amd.hip.optim import FusedAdam
from pcgan_amd.models import networks
from pcgan_amd.util.util import reparameterize, str2bool


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    add = ap.add_argument
    add('--mode', type=str, default='train')
    add('--name', type=str, default='exp')
    add('--dataroot', required=True)
    add('--datafile', type=str, default='')
    add('--dataroot_val', type=str, default='')
    add('--datafile_val', type=str, default='')
    add('--pretrained_model_path', type=str, default='pretrained_models/resnet18-5c106cde.pth')
    add('--checkpoint_dir', type=str, default='checkpoints')
    add('--save_epoch_freq', type=int, default=5)
    add('--num_workers', type=int, default=4)
    add('--num_epochs', type=int, default=50)
    add('--batch_size', type=int, default=100)
    add('--lr', type=float, default=0.0002)
    add('--which_model', type=str, default='resnet18')
    add('--pooling', type=str, default='avg')
    add('--loadSize', type=int, default=240)
    add('--fineSize', type=int, default=224)
    add('--gpu_ids', type=str, default='0')
    add('--fc_dim', type=int, nargs='*', default=[])
    add('--cnn_dim', type=int, nargs='*', default=[32, 1])
    add('--no_cnn', action='store_true')
    add('--cnn_pad', type=int, default=1)
    add('--cnn_relu_slope', type=float, default=0.7)
    add('--use_cxn', action='store_true')
    add('--finetune_fc_only', action='store_true')
    add('--print_freq', type=int, default=10)
    add('--display_id', type=int, default=-1)
    add('--transforms', type=str, default='resize_and_crop')
    add('--no_flip', action='store_true')
    add('--continue_train', action='store_true')
    add('--which_epoch', type=str, default='latest')
    add('--epoch_count', type=int, default=1)
    add('--save_latest_freq', type=int, default=100)
    add('--serial_batches', action='store_true')
    add('--draw_prob_thresh', type=float, default=0.16)
    add('--noisy', type=str2bool, default=False)
    add('--bayesian', type=str2bool, default=False)
    add('--bnn_dropout', type=float, default=0.)
    add('--M', type=int, default=1, help='number of reparameterization samples')
    add('--T_train', type=int, default=1, help='MC-dropout passes per iteration (--bayesian)')
    add('--T', type=int, default=10, help='number of Bayesian samples (--mode embedding)')
    add('--lr_sigma', type=float, default=0.0000002, help='learning rate of the log-variance head (--noisy)')
    add('--noisy_sigma_updating_epochs', nargs='*', type=int, default=[0, 1], help='starts ends (-1: --num_epochs)')
    add('--rsample', type=str2bool, default=True)
    add('--lb_or_mc', type=str, default='lb', choices=['lb', 'mc'])
    add('--seed', type=int, default=0)
    add('--max_dataset_size', type=int, default=1 << 30)
    return ap


class PairDataset(torch.utils.data.Dataset):
    """lines "<imageA> <imageB> <label>" under --dataroot (reference siamese.py:158-180), or synthetic pairs"""

    def __init__(self, opt, root, listing):
        self.opt = opt
        self.root = root
        self.synthetic = root == 'synthetic'
        if self.synthetic:
            self.n = min(opt.max_dataset_size, 64 * opt.batch_size)
            return
        with open(listing) as f:
            self.lines = [line.split() for line in f if line.strip()][:opt.max_dataset_size]
        self.n = len(self.lines)
        opt.isTrain = True
        self.transform = get_transform(opt)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if self.synthetic:
            g = torch.Generator().manual_seed(9000 + i)
            s = self.opt.fineSize
            level = torch.rand(2, generator=g)                                  # the hidden rating: mean brightness
            imgs = [(torch.rand(3, s, s, generator=g) * 0.5 + level[k] - 0.75) for k in range(2)]
            d = float(level[0] - level[1])
            label = 1 if abs(d) < 0.1 else (2 if d > 0 else 0)
            return imgs[0], imgs[1], torch.tensor(label)
        from PIL import Image
        a, b, lab = self.lines[i][:3]
        A = self.transform(Image.open(os.path.join(self.root, a)).convert('RGB'))
        B = self.transform(Image.open(os.path.join(self.root, b)).convert('RGB'))
        return A, B, torch.tensor(int(lab))


def init_like_reference(net):
    """weights_init of the reference (siamese.py:288-296): every conv N(0, 0.02), BatchNorm weight N(1, 0.02), bias 0"""
    for m in net.modules():
        kind = m.__class__.__name__
        if 'Conv' in kind:
            m.weight.data.normal_(0.0, 0.02)
        elif 'BatchNorm2d' in kind:
            m.weight.data.normal_(1.0, 0.02)
            m.bias.data.fill_(0)


def build_net(opt, device):
    if 'resnet' not in opt.which_model:
        raise NotImplementedError('pcgan_amd: rating trunk [%s] is outside the MI355X hot path' % opt.which_model)
    if opt.finetune_fc_only:
        raise NotImplementedError('pcgan_amd: siamese.py --finetune_fc_only is outside the MI355X hot path')
    base = networks.ResNetFeature(input_nc=3, which_model=opt.which_model, dropout=opt.bnn_dropout)
    net = networks.SiameseNetwork(base, pooling=opt.pooling, cnn_dim=[] if opt.no_cnn else opt.cnn_dim, cnn_pad=opt.cnn_pad,
                                  cnn_relu_slope=opt.cnn_relu_slope, fc_dim=opt.fc_dim, use_cxn=opt.use_cxn, noisy=opt.noisy,
                                  drop_layer=networks.get_dropout_layer(dropout=opt.bnn_dropout), rsample=opt.rsample)
    save_dir = os.path.join(opt.checkpoint_dir, opt.name)
    if opt.continue_train:
        net.load_state_dict(torch.load(os.path.join(save_dir, '%s_net.pth' % opt.which_epoch), map_location='cpu'), strict=False)
    else:
        init_like_reference(net)
        if opt.pretrained_model_path:
            net.load_pretrained(opt.pretrained_model_path)
    return net.to(device)


MAGIC_EPS = 1e-20


def iteration_loss(opt, net, criterion, img0, img1, label):
    """loss and probabilities of one training iteration (reference siamese.py:596-665): the deterministic recipe, T_train
    MC-dropout passes (--bayesian), and with --noisy either the score / score_std form or M reparameterised ratings per pass,
    combined as a Monte-Carlo estimate of the probability (`mc`) or of the loss (`lb`)."""
    T = opt.T_train if opt.bayesian else 1
    loss, prob = 0.0, None
    for _ in range(T):
        if opt.noisy and opt.rsample:
            y1, y2, logvar1, logvar2 = net(img0, img1)
            if opt.lb_or_mc == 'mc':
                prob = 0.0
                for _m in range(opt.M):
                    score = reparameterize(y1, logvar1) - reparameterize(y2, logvar2)
                    prob = prob + 1. / opt.M * torch.sigmoid(score)
                loss = loss + 1. / T * criterion(prob, label)
            else:
                for _m in range(opt.M):
                    score = reparameterize(y1, logvar1) - reparameterize(y2, logvar2)
                    prob = torch.sigmoid(score)
                    loss = loss + 1. / (T * opt.M) * criterion(prob, label)
        elif opt.noisy:
            _, _, score, score_std = net(img0, img1)
            prob = torch.sigmoid(score / (score_std + MAGIC_EPS))
            loss = loss + 1. / T * criterion(prob, label)
        else:
            _, _, score = net(img0, img1)
            prob = torch.sigmoid(score)
            loss = loss + 1. / T * criterion(prob, label)
    return loss, prob


def sigma_lr_rule(opt):
    """LambdaLR factor of the log-variance head's optimizer (reference siamese.py:555-565): lr_sigma until the first epoch of
    --noisy_sigma_updating_epochs, ramped linearly to --lr at the second"""
    e0, e1 = [opt.num_epochs if e == -1 else e for e in opt.noisy_sigma_updating_epochs]

    def rule(epoch):
        if epoch + opt.epoch_count <= e0:
            return 1.0
        if epoch + opt.epoch_count >= e1:
            return 1.0 * opt.lr / opt.lr_sigma
        return 1.0 * opt.lr / opt.lr_sigma * (float(epoch + opt.epoch_count) - e0) / float(e1 - e0)
    return rule


def predictions(prob, draw_thresh):
    """0 / 1 / 2 per pair from P(first > second) (reference siamese.py:323-331 on (N,1,1,1) probabilities)"""
    p = prob.detach().reshape(prob.size(0), -1)[:, 0]
    draw = (0.5 - p).abs() < draw_thresh
    return torch.where(draw, torch.ones_like(p), torch.where(p > 0.5, torch.full_like(p, 2.0), torch.zeros_like(p))).long()


def save(net, path):
    if not parallel.is_distributed() or torch.distributed.get_rank() == 0:
        torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()}, path)


def train(opt):
    world, rank, local = parallel.init_process_group()
    torch.manual_seed(opt.seed)
    gpu = int(opt.gpu_ids.split(',')[0])
    if gpu < 0 or not torch.cuda.is_available():
        raise RuntimeError('pcgan_amd: siamese.py needs an MI355X (no CPU fallback)')
    device = torch.device('cuda', local % torch.cuda.device_count() if world > 1 else gpu)
    torch.cuda.set_device(device)
    net = build_net(opt, device)
    parallel.broadcast_parameters(net)
    criterion = networks.BinaryNLLLoss()
    params = list(net.base.parameters()) + (list(net.cnn.parameters()) if net.cnn is not None else [])
    optimizer = FusedAdam(params, lr=opt.lr)                                   # optim.Adam(param, lr) of the reference
    optimizer_sigma = scheduler_sigma = None
    if opt.noisy:                                                              # siamese.py:552-566
        optimizer_sigma = FusedAdam(net.cnn_logvar.parameters(), lr=opt.lr_sigma)
        scheduler_sigma = torch.optim.lr_scheduler.LambdaLR(optimizer_sigma, lr_lambda=sigma_lr_rule(opt))
    data = PairDataset(opt, opt.dataroot, opt.datafile)
    loader = torch.utils.data.DataLoader(data, batch_size=opt.batch_size, shuffle=not opt.serial_batches,
                                         num_workers=0 if data.synthetic else opt.num_workers)
    save_dir = os.path.join(opt.checkpoint_dir, opt.name)
    os.makedirs(save_dir, exist_ok=True)
    save(net, os.path.join(save_dir, 'init_net.pth'))
    history, total = [], 0
    for epoch in range(opt.epoch_count, opt.num_epochs + opt.epoch_count):
        wrong = seen = 0
        for img0, img1, label in loader:
            img0, img1, label = (parallel.shard_batch(t, rank, world).to(device) for t in (img0, img1, label))
            total += 1
            optimizer.zero_grad()
            if optimizer_sigma is not None:
                optimizer_sigma.zero_grad()
            loss, prob = iteration_loss(opt, net, criterion, img0, img1, label)
            loss.backward()
            parallel.sync_gradients(optimizer)
            optimizer.step()
            if optimizer_sigma is not None:
                parallel.sync_gradients(optimizer_sigma)
                optimizer_sigma.step()
            wrong += int((predictions(prob, opt.draw_prob_thresh) != label).sum())
            seen += int(label.numel())
            if total % opt.print_freq == 0:
                value = float(loss)
                history.append(value)
                if rank == 0:
                    print('epoch %02d, iter %06d, loss: %.4f' % (epoch, total, value))
            if total % opt.save_latest_freq == 0:
                save(net, os.path.join(save_dir, 'latest_net.pth'))
        if rank == 0:
            print('epoch %02d: train accuracy %.4f' % (epoch, 1.0 - wrong / max(seen, 1)))
        if scheduler_sigma is not None:
            scheduler_sigma.step()
            if rank == 0:
                print('--->> lr      : %g' % optimizer.param_groups[0]['lr'])
                print('--->> lr_sigma: %g' % optimizer_sigma.param_groups[0]['lr'])
        save(net, os.path.join(save_dir, 'latest_net.pth'))
        if epoch % opt.save_epoch_freq == 0:
            save(net, os.path.join(save_dir, '%d_net.pth' % epoch))
    if rank == 0:
        with open(os.path.join(save_dir, 'loss.txt'), 'w') as f:
            f.writelines('%r\n' % v for v in history)
    return history


class SingleImageDataset(torch.utils.data.Dataset):
    """lines "<image> [<attribute>]" of --datafile under --dataroot, or the directory listing (reference siamese.py:182-199);
    yields (image, line).  `--dataroot synthetic`: seeded images named "<brightness>_<i>.png" (the hidden rating as the label)."""

    def __init__(self, opt, root, listing):
        self.opt, self.root = opt, root
        self.synthetic = root == 'synthetic'
        if self.synthetic:
            self.lines = ['%.4f_%d.png' % (0.1 + 0.8 * ((i * 37) % 100) / 100.0, i) for i in range(min(opt.max_dataset_size, 64))]
        elif listing:
            with open(listing) as f:
                self.lines = [line.strip('\n') for line in f.readlines() if line.strip()][:opt.max_dataset_size]
        else:
            self.lines = sorted(os.listdir(root))[:opt.max_dataset_size]
        if not self.synthetic:
            self.transform = get_transform(opt)

    def __len__(self):
        return len(self.lines)

    def __getitem__(self, i):
        if self.synthetic:
            level = float(self.lines[i].split('_')[0])
            g = torch.Generator().manual_seed(7000 + i)
            s = self.opt.fineSize
            return torch.rand(3, s, s, generator=g) * 0.5 + level - 0.75, self.lines[i]
        from PIL import Image
        return self.transform(Image.open(os.path.join(self.root, self.lines[i].split()[0])).convert('RGB')), self.lines[i]


def get_attr_value(fname):
    """reference siamese.py:299-303: the second token of a listing line, else the file name up to its first underscore"""
    return float(fname.split()[1]) if len(fname.split()) > 1 else float(fname.split('_')[0])


def _device(opt):
    gpu = int(opt.gpu_ids.split(',')[0])
    if gpu < 0 or not torch.cuda.is_available():
        raise RuntimeError('pcgan_amd: siamese.py needs an MI355X (no CPU fallback)')
    device = torch.device('cuda', gpu)
    torch.cuda.set_device(device)
    return device


def build_feature_net(opt, device, state_dict=None):
    """the single-image rating net of `--mode embedding` (reference get_model, siamese.py:450-476): SiameseFeature loaded from
    <checkpoint_dir>/<name>/<which_epoch>_net.pth with strict=False (the trainer's comparison head `cxn` / `fc` is not part of it),
    parameters frozen.  Like the reference it is NOT switched to eval(): BatchNorm normalises with the statistics of the (single)
    image and Dropout2d stays active for the MC passes."""
    if 'resnet' not in opt.which_model:
        raise NotImplementedError('pcgan_amd: rating trunk [%s] is outside the MI355X hot path' % opt.which_model)
    base = networks.ResNetFeature(input_nc=3, which_model=opt.which_model, dropout=opt.bnn_dropout)
    net = networks.SiameseFeature(base, pooling=opt.pooling, cnn_dim=[] if opt.no_cnn else opt.cnn_dim, cnn_pad=opt.cnn_pad,
                                  cnn_relu_slope=opt.cnn_relu_slope, noisy=opt.noisy,
                                  drop_layer=networks.get_dropout_layer(dropout=opt.bnn_dropout))
    if state_dict is None:
        state_dict = torch.load(os.path.join(opt.checkpoint_dir, opt.name, '%s_net.pth' % opt.which_epoch), map_location='cpu')
    net.load_state_dict(state_dict, strict=False)
    for p in net.parameters():
        p.requires_grad = False
    return net.to(device)


def embedding(opt, net=None, which_epoch=None):
    """`--mode embedding` (reference siamese.py:791-852): the rating of every image of --datafile, one image per forward pass, written
    as features_<epoch>.npy / labels_<epoch>.npy (+ stds_ with --noisy: exp(logvar / 2); + vars_ with --bayesian: the variance over
    --T MC-dropout passes, the feature being their mean) under <checkpoint_dir>/<name>/ -- the ratings from which the GAN's
    --embedding_mean / --embedding_std / --embedding_bins are derived.  Returns (features, labels)."""
    device = _device(opt)
    opt.isTrain = False if opt.no_flip else True      # (get_transform flips only in train mode; the reference passes its own opt)
    if net is None:
        net = build_feature_net(opt, device)
    which_epoch = which_epoch or opt.which_epoch
    data = SingleImageDataset(opt, opt.dataroot, opt.datafile)
    loader = torch.utils.data.DataLoader(data, shuffle=False, num_workers=0, batch_size=1)
    features, labels, stds, variances = [], [], [], []
    fd = net.feature_dim
    with torch.no_grad():
        for img0, path0 in loader:
            img0 = img0.to(device)
            if opt.bayesian:
                feats, std2 = [], 0.0
                for _ in range(opt.T):
                    out = net(img0)
                    f, logvar = out if opt.noisy else (out, None)
                    feats.append(f.detach().cpu().numpy().reshape(1, fd))
                    if logvar is not None:
                        s_ = torch.exp(0.5 * logvar).detach().cpu().numpy()
                        std2 = std2 + 1.0 / opt.T * s_ * s_
                feats = np.concatenate(feats, axis=0)
                feature = np.mean(feats, axis=0)
                variances.append(np.var(feats, axis=0).reshape(1, fd))
                if opt.noisy:
                    stds.append(np.sqrt(std2).reshape(1, fd))
            elif opt.noisy:
                f, logvar = net(img0)
                feature = f.detach().cpu().numpy()
                stds.append(torch.exp(0.5 * logvar).detach().cpu().numpy().reshape(1, fd))
            else:
                feature = net(img0).detach().cpu().numpy()
            features.append(np.asarray(feature).reshape(1, fd))
            labels.append(get_attr_value(path0[0]))
    X, L = np.concatenate(features, axis=0), np.array(labels)
    out_dir = os.path.join(opt.checkpoint_dir, opt.name)
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, 'features_%s.npy' % which_epoch), X)
    np.save(os.path.join(out_dir, 'labels_%s.npy' % which_epoch), L)
    if opt.noisy:
        np.save(os.path.join(out_dir, 'stds_%s.npy' % which_epoch), np.concatenate(stds, axis=0))
    if opt.bayesian:
        np.save(os.path.join(out_dir, 'vars_%s.npy' % which_epoch), np.concatenate(variances, axis=0))
    return X, L


def test(opt):
    """`--mode test` (reference siamese.py:771-789): accuracy of the pair predictions (0: first < second, 1: draw, 2: first > second)
    over --datafile.  The reference builds a SiameseFeature for this mode and then calls it on pairs (get_model, :450), which cannot
    run; what its `test` evidently means -- the trained SiameseNetwork's P(first > second) against the labels with the draw band
    --draw_prob_thresh -- is what runs here.  Returns the accuracy in percent."""
    device = _device(opt)
    opt.continue_train = True                  # build_net then loads <which_epoch>_net.pth
    net = build_net(opt, device)
    data = PairDataset(opt, opt.dataroot, opt.datafile)
    loader = torch.utils.data.DataLoader(data, shuffle=False, batch_size=opt.batch_size, num_workers=0 if data.synthetic else opt.num_workers)
    wrong = seen = 0
    with torch.no_grad():
        for i, (img0, img1, label) in enumerate(loader):
            out = net(img0.to(device), img1.to(device))
            prob = torch.sigmoid(out[0] - out[1])          # P(first > second) from the two ratings (every head variant returns them first)
            wrong += int((predictions(prob, opt.draw_prob_thresh).cpu() != label.reshape(-1)).sum())
            seen += int(label.numel())
            print('--> batch #%d' % (i + 1))
    acc = 100.0 * (1.0 - wrong / max(seen, 1))
    print('================================================================================')
    print('accuracy: %.6f' % acc)
    return acc


if __name__ == '__main__':
    options = build_parser().parse_args()
    if options.mode == 'train':
        train(options)
    elif options.mode == 'embedding':
        embedding(options)
    elif options.mode == 'test':
        test(options)
    else:
        raise NotImplementedError('pcgan_amd: siamese.py --mode %s is outside the MI355X hot path (train | embedding | test; the '
                                  'optimize / attention modes need visdom / Grad-CAM tooling)' % options.mode)
