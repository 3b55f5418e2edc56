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
