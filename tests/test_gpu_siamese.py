"""GPU parity of the Elo-encoder training step: networks.SiameseNetwork + BinaryNLLLoss + FusedAdam on
the HIP path against the reference's golden vectors (first step: identical weights) and the oracle's float64 twin for the
gradients; then siamese.py's own loop on synthetic pairs must reduce the loss.
Tolerances: loss 2e-4; ratings / probabilities 2e-4 of the largest magnitude; gradients |hip - g64| <= 2 |ref32 - g64| +
3e-2 |g64| per tensor (ReLU / max-pool mask flips, see test_gpu_nets.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import weights as W
from test_siamese_oracle_golden import build_oracle, siamese_inputs
from util_cmp import assert_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_siamese_step_matches_reference_and_oracle(dev):
    from pcgan_amd.models import networks
    from pcgan_amd.hip.optim import FusedAdam
    gold = np.load(os.path.join(GOLD, 'siamese_step.npz'))
    net = networks.SiameseNetwork(networks.ResNetFeature(3, 'resnet18'), pooling='avg', cnn_dim=[32, 1], cnn_pad=1,
                                  cnn_relu_slope=0.7)
    net.load_state_dict(W.fill_state_dict({k: v.cpu() for k, v in net.state_dict().items()}, 61))
    net.to(dev)
    crit = networks.BinaryNLLLoss()
    opt = FusedAdam(list(net.base.parameters()) + list(net.cnn.parameters()), lr=2e-4)
    oracle, twin = build_oracle(), build_oracle(torch.float64)
    img0, img1, label = siamese_inputs(0)
    oracle.step(img0, img1, label)
    twin.step(img0.double(), img1.double(), label)
    opt.zero_grad()
    f1, f2, score = net(img0.to(dev), img1.to(dev))
    prob = torch.sigmoid(score)
    loss = crit(prob, label.to(dev))
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    opt.step()
    assert abs(float(loss) - float(gold['it0/loss'])) <= 2e-4
    assert_close(f1, torch.from_numpy(gold['it0/f1']), 2e-4, 'rating of image 0 vs reference')
    assert_close(prob, torch.from_numpy(gold['it0/prob']), 2e-4, 'probability vs reference')
    for k, g64 in twin.grads.items():
        scale = float(g64.norm())
        e_hip = float((grads[k].double().cpu() - g64).norm())
        e_ref = float((oracle.grads[k].double() - g64).norm())
        assert e_hip <= 2 * e_ref + 3e-2 * scale + 1e-6, 'grad %s: |hip-g64| %.3e |ref32-g64| %.3e |g64| %.3e' % (k, e_hip, e_ref, scale)
    for k, v in net.state_dict().items():
        ref = gold['it0/after/' + k]
        assert abs(float(v.double().abs().sum()) - ref[1]) <= 2e-3 * (ref[1] + 1e-3) + 2.02 * 2e-4 * v.numel(), 'after-step ' + k


def test_siamese_script_trains_on_synthetic_pairs(tmp_path, dev):
    sys.path.insert(0, ROOT)
    import siamese
    opt = siamese.build_parser().parse_args(
        ['--dataroot', 'synthetic', '--name', 'elo', '--checkpoint_dir', str(tmp_path), '--batch_size', '16',
         '--num_epochs', '2', '--fineSize', '64', '--max_dataset_size', '256', '--print_freq', '1', '--save_epoch_freq', '1',
         '--pretrained_model_path', '', '--lr', '0.001'])
    history = siamese.train(opt)
    assert len(history) == 32 and all(np.isfinite(history))
    assert np.mean(history[-8:]) < np.mean(history[:8]) - 0.05, 'loss did not fall: %s' % history
    for f in ('init_net.pth', 'latest_net.pth', '1_net.pth', '2_net.pth', 'loss.txt'):
        assert os.path.exists(os.path.join(str(tmp_path), 'elo', f)), f
    # the checkpoint is what wsgan_emb's encoder loads (SiameseFeature.load_pretrained, strict)
    from pcgan_amd.models import networks
    e = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
    e.load_pretrained(os.path.join(str(tmp_path), 'elo', '2_net.pth'))


@pytest.mark.parametrize('name', ['bayesian', 'noisy_std', 'noisy_mc', 'noisy_lb', 'bayesian_noisy_lb', 'bayesian_noisy_std'])
def test_siamese_variant_iteration_matches_reference_and_oracle(dev, name):
    """the trainer's reparameterised / MC-dropout iterations (siamese.iteration_loss, second Adam on cnn_logvar) against the
    reference's golden vectors and the oracle; the oracle's dropout masks and eps draws are replayed on the GPU"""
    sys.path.insert(0, ROOT)
    import siamese
    from oracle import networks_ref as N
    from pcgan_amd.hip import nn as hnn
    from pcgan_amd.hip.optim import FusedAdam
    from pcgan_amd.models import networks
    from pcgan_amd.util import util as hutil
    from test_siamese_oracle_golden import VARIANTS, LR_SIGMA, build_variant_oracle
    noisy, rsample, lb_or_mc, p_drop, T, M = VARIANTS[name]
    gold = np.load(os.path.join(GOLD, 'siamese_variants.npz'))
    opt = siamese.build_parser().parse_args(['--dataroot', 'synthetic', '--noisy', str(noisy), '--bayesian', str(p_drop > 0), '--bnn_dropout', str(p_drop),
                                             '--T_train', str(T), '--M', str(M), '--rsample', str(rsample), '--lb_or_mc', lb_or_mc,
                                             '--lr_sigma', str(LR_SIGMA)])
    net = networks.SiameseNetwork(networks.ResNetFeature(3, 'resnet18', dropout=p_drop), pooling='avg', cnn_dim=[32, 1], cnn_pad=1, cnn_relu_slope=0.7,
                                  noisy=noisy, drop_layer=networks.get_dropout_layer(p_drop), rsample=rsample)
    net.load_state_dict(W.fill_state_dict({k: v.cpu() for k, v in net.state_dict().items()}, 61))
    net.to(dev)
    crit = networks.BinaryNLLLoss()
    optimizer = FusedAdam(list(net.base.parameters()) + list(net.cnn.parameters()), lr=2e-4)
    optimizer_sigma = FusedAdam(net.cnn_logvar.parameters(), lr=LR_SIGMA) if noisy else None
    oracle = build_variant_oracle(name)
    for it in range(2):
        img0, img1, label = siamese_inputs(it)
        # align the weights of the two sides (Adam turns rounding noise of near-zero gradients into +-lr moves)
        if it > 0:
            net.load_state_dict({k: v.to(dev) for k, v in oracle.net.state_dict().items()})
        N.Dropout2dRec.record = []
        oracle.draws = []
        torch.manual_seed(1000 + it)
        oracle.step(img0, img1, label)
        masks, N.Dropout2dRec.record = N.Dropout2dRec.record, None
        hnn.Dropout2d.mask_source = iter(masks)
        hutil.inject_noise(iter(oracle.draws))
        try:
            optimizer.zero_grad()
            if noisy:
                optimizer_sigma.zero_grad()
            loss, prob = siamese.iteration_loss(opt, net, crit, img0.to(dev), img1.to(dev), label.to(dev))
            loss.backward()
            assert next(hnn.Dropout2d.mask_source, None) is None, 'the HIP iteration consumed fewer dropout masks than the oracle drew'
            assert next(iter(hutil._inject['eps']), None) is None, 'the HIP iteration drew fewer eps than the oracle'
        finally:
            hnn.Dropout2d.mask_source = None
            hutil.inject_noise(None)
        q = '%s/it%d' % (name, it)
        assert abs(float(loss) - oracle.loss.item()) <= 2e-4 * max(1.0, abs(oracle.loss.item())), (q, float(loss), oracle.loss.item())
        assert_close(prob, oracle.prob.detach(), 5e-4, q + ' probability vs oracle')
        if it == 0:
            assert abs(float(loss) - float(gold[q + '/loss'])) <= 2e-4, (q, float(loss), float(gold[q + '/loss']))
            assert_close(prob, torch.from_numpy(gold[q + '/prob']), 5e-4, q + ' probability vs reference')
        for k, g in oracle.grads.items():
            hg = dict(net.named_parameters())[k].grad
            scale = float(g.norm())
            if scale < 1e-7:
                continue
            e = float((hg.double().cpu() - g.double()).norm()) / scale
            assert e <= 5e-2, '%s grad %s: relative L2 against the oracle %.3e' % (q, k, e)
        optimizer.step()
        if noisy:
            optimizer_sigma.step()


def test_siamese_script_trains_noisy_bayesian(tmp_path, dev):
    sys.path.insert(0, ROOT)
    import siamese
    opt = siamese.build_parser().parse_args(
        ['--dataroot', 'synthetic', '--name', 'elo_nb', '--checkpoint_dir', str(tmp_path), '--batch_size', '16',
         '--num_epochs', '2', '--fineSize', '64', '--max_dataset_size', '128', '--print_freq', '1', '--save_epoch_freq', '1',
         '--pretrained_model_path', '', '--lr', '0.001', '--noisy', 'true', '--bayesian', 'true', '--bnn_dropout', '0.1',
         '--T_train', '2', '--M', '2', '--lb_or_mc', 'mc', '--lr_sigma', '1e-5', '--noisy_sigma_updating_epochs', '0', '2'])
    history = siamese.train(opt)
    assert len(history) == 16 and all(np.isfinite(history))
    # the checkpoint carries the log-variance head: it is what wsgan_emb --noisy true loads
    from pcgan_amd.models import networks
    e = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7, noisy=True, bnn_dropout=0.1)
    e.load_pretrained(os.path.join(str(tmp_path), 'elo_nb', '2_net.pth'))


def test_embedding_and_test_modes_close_the_elo_pipeline(tmp_path, dev):
    """siamese.py --mode embedding / --mode test (reference siamese.py:771-852, f2 leftovers of round 2): train a rating net on the
    synthetic pairs, extract the ratings of single images with `--mode embedding` -- the .npy files from which the GAN's
    --embedding_mean / --embedding_std / --embedding_bins are derived -- and score pair predictions with `--mode test`.  The
    ratings are held to the ORACLE's SiameseFeature on the same checkpoint (train-mode BatchNorm over the single image, as the
    reference runs it: it never calls eval()), 2e-4; the synthetic rating (mean brightness) must come out monotone."""
    sys.path.insert(0, ROOT)
    import siamese
    from oracle import networks_ref as N
    common = ['--dataroot', 'synthetic', '--name', 'elo_emb', '--checkpoint_dir', str(tmp_path), '--fineSize', '64',
              '--pretrained_model_path', '', '--max_dataset_size', '256']
    opt = siamese.build_parser().parse_args(common + ['--batch_size', '16', '--num_epochs', '3', '--print_freq', '4', '--lr', '0.001'])
    siamese.train(opt)
    # --mode embedding
    eopt = siamese.build_parser().parse_args(common + ['--mode', 'embedding', '--which_epoch', 'latest', '--no_flip', '--max_dataset_size', '12'])
    X, L = siamese.embedding(eopt)
    out_dir = os.path.join(str(tmp_path), 'elo_emb')
    assert X.shape == (12, 1) and L.shape == (12,) and np.isfinite(X).all()
    assert np.array_equal(np.load(os.path.join(out_dir, 'features_latest.npy')), X)
    assert np.array_equal(np.load(os.path.join(out_dir, 'labels_latest.npy')), L)
    assert abs(siamese.get_attr_value('img.png 23.5') - 23.5) < 1e-12 and abs(siamese.get_attr_value('31_abc.png') - 31.0) < 1e-12
    # against the oracle's SiameseFeature on the same checkpoint, image by image, in train mode (batch statistics of ONE image)
    sd = torch.load(os.path.join(out_dir, 'latest_net.pth'), map_location='cpu')
    ref = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
    ref.load_state_dict({k: v for k, v in sd.items() if k in ref.state_dict()}, strict=False)
    ref.train()
    data = siamese.SingleImageDataset(eopt, 'synthetic', '')
    with torch.no_grad():
        for i in range(12):
            img, line = data[i]
            want = float(ref(img.unsqueeze(0)).reshape(-1)[0])
            assert abs(float(X[i, 0]) - want) <= 2e-4 * max(1.0, abs(want)), (i, float(X[i, 0]), want)
            assert abs(L[i] - float(line.split('_')[0])) < 1e-12
    # the trained rating follows the hidden one (mean brightness): rank correlation clearly positive after 3 short epochs
    order_l, order_x = np.argsort(np.argsort(L)), np.argsort(np.argsort(X[:, 0]))
    rho = np.corrcoef(order_l, order_x)[0, 1]
    assert rho > 0.5, rho
    # bayesian + noisy variant of the extraction: T passes, stds / vars files
    bopt = siamese.build_parser().parse_args(['--dataroot', 'synthetic', '--name', 'elo_nb2', '--checkpoint_dir', str(tmp_path), '--fineSize', '64',
                                              '--pretrained_model_path', '', '--batch_size', '16', '--num_epochs', '1', '--max_dataset_size', '64',
                                              '--noisy', 'true', '--bayesian', 'true', '--bnn_dropout', '0.1', '--lr_sigma', '1e-5'])
    siamese.train(bopt)
    bopt.mode, bopt.T, bopt.max_dataset_size, bopt.no_flip = 'embedding', 3, 5, True
    Xb, Lb = siamese.embedding(bopt)
    nb_dir = os.path.join(str(tmp_path), 'elo_nb2')
    S, V = np.load(os.path.join(nb_dir, 'stds_latest.npy')), np.load(os.path.join(nb_dir, 'vars_latest.npy'))
    assert Xb.shape == S.shape == V.shape == (5, 1) and (S > 0).all() and (V >= 0).all() and V.max() > 0
    # --mode test
    topt = siamese.build_parser().parse_args(common + ['--mode', 'test', '--batch_size', '16', '--max_dataset_size', '64'])
    acc = siamese.test(topt)
    assert 50.0 < acc <= 100.0, acc
