"""Elo-encoder trainer: the oracle's restatement of the training step against the vectors captured
from the reference's own SiameseNetwork (oracle/make_golden.py --only siamese -> tests/golden/siamese_step.npz)."""
import os

import numpy as np
import torch

from oracle import siamese_ref as SR
from oracle import weights as W
from util_cmp import assert_close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
LABELS = [[0, 2, 1, 2], [2, 2, 0, 1], [1, 0, 0, 2]]


def siamese_inputs(it):
    return (W.seeded_tensor((4, 3, 64, 64), 900 + it), W.seeded_tensor((4, 3, 64, 64), 950 + it), torch.tensor(LABELS[it]))


def build_oracle(dtype=torch.float32):
    m = SR.SiameseTrainRef()
    m.net.load_state_dict(W.fill_state_dict(m.net.state_dict(), 61))     # same keys as the reference's SiameseNetwork
    m.net.to(dtype)
    return m


def test_siamese_step_matches_reference():
    torch.set_num_threads(4)
    gold = np.load(os.path.join(GOLD, 'siamese_step.npz'))
    m = build_oracle()
    for it in range(3):
        img0, img1, label = siamese_inputs(it)
        loss = m.step(img0, img1, label)
        p = 'it%d' % it
        assert abs(loss - float(gold[p + '/loss'])) <= 2e-5, 'loss it%d: %r vs %r' % (it, loss, float(gold[p + '/loss']))
        assert_close(m.y1, torch.from_numpy(gold[p + '/f1']), 2e-5, 'rating of image 0')
        assert_close(m.prob, torch.from_numpy(gold[p + '/prob']), 2e-5, 'probability')
        for k, g in m.grads.items():
            st = gold['%s/grad/stat/%s' % (p, k)]
            l2 = float(g.double().pow(2).sum().sqrt())
            assert abs(l2 - st[2]) <= 2e-3 * st[2] + 1e-6, 'grad %s l2 %g vs %g' % (k, l2, st[2])
        for k, v in m.net.state_dict().items():
            ref = gold['%s/after/%s' % (p, k)]
            assert abs(float(v.double().abs().sum()) - ref[1]) <= 1e-4 * (ref[1] + 1e-3), 'after-step %s' % k


# name -> (noisy, rsample, lb_or_mc, bnn_dropout, T_train, M): oracle/make_golden.py SIAMESE_VARIANTS
VARIANTS = {
    'bayesian': (False, True, 'lb', 0.2, 2, 1),
    'noisy_std': (True, False, 'lb', 0.0, 1, 1),
    'noisy_mc': (True, True, 'mc', 0.0, 1, 3),
    'noisy_lb': (True, True, 'lb', 0.0, 1, 2),
    'bayesian_noisy_lb': (True, True, 'lb', 0.2, 2, 2),
    'bayesian_noisy_std': (True, False, 'lb', 0.2, 2, 1),
}
LR_SIGMA = 1e-4


def build_variant_oracle(name, dtype=torch.float32):
    noisy, rsample, lb_or_mc, p, T, M = VARIANTS[name]
    m = SR.SiameseTrainRef(noisy=noisy, rsample=rsample, lb_or_mc=lb_or_mc, bnn_dropout=p, T_train=T, M=M, lr_sigma=LR_SIGMA)
    m.net.load_state_dict(W.fill_state_dict(m.net.state_dict(), 61))
    m.net.to(dtype)
    return m


def test_siamese_variants_match_reference():
    """the reparameterised / MC-dropout branches of the trainer's iteration against the reference's SiameseNetwork and
    reparameterize under the same seeds (tests/golden/siamese_variants.npz)"""
    import pytest
    torch.set_num_threads(4)
    gold = np.load(os.path.join(GOLD, 'siamese_variants.npz'))
    for name in VARIANTS:
        m = build_variant_oracle(name)
        for it in range(2):
            img0, img1, label = siamese_inputs(it)
            torch.manual_seed(1000 + it)
            loss = m.step(img0, img1, label)
            q = '%s/it%d' % (name, it)
            assert abs(loss - float(gold[q + '/loss'])) <= 3e-5, '%s loss: %r vs %r' % (q, loss, float(gold[q + '/loss']))
            assert_close(m.y1, torch.from_numpy(gold[q + '/f1']), 3e-5, q + ' rating of image 0 (last pass)')
            assert_close(m.prob, torch.from_numpy(gold[q + '/prob']), 3e-5, q + ' probability')
            for k, g in m.grads.items():
                st = gold['%s/grad/stat/%s' % (q, k)]
                l2 = float(g.double().pow(2).sum().sqrt())
                assert abs(l2 - st[2]) <= 3e-3 * st[2] + 1e-6, '%s grad %s l2 %g vs %g' % (q, k, l2, st[2])
            for k, v in m.net.state_dict().items():
                ref = gold['%s/after/%s' % (q, k)]
                assert abs(float(v.double().abs().sum()) - ref[1]) <= 1e-4 * (ref[1] + 1e-3), '%s after-step %s' % (q, k)
