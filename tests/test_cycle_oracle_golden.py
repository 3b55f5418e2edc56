"""wsgan_cycle: the oracle's CPU restatement of WSGANCycleModel.optimize_parameters() against
the vectors captured from the reference itself (oracle/make_golden.py --only cycle -> tests/golden/cycle_step.npz)."""
import os

import numpy as np
import torch

from oracle import networks_ref as N
from oracle import step_ref as S
from oracle import weights as W
from util_cmp import assert_close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def build_cycle_oracle(dtype=torch.float32):
    """The oracle-side twin of make_golden.golden_cycle_step."""
    G = N.ResnetGeneratorRef(3, 3, 1, 8, 'instance', 9)
    G.load_state_dict(W.damp_generator_head(W.fill_state_dict(G.state_dict(), 19)))
    D = N.NLayerDiscriminatorRef(3, 0, 8, 3, 'batch', True)
    D.load_state_dict(W.fill_state_dict(D.state_dict(), 20))
    E = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'max', (64, 1), 1, 0.7, False)
    esd = E.state_dict()
    base = W.fill_state_dict({k[len('base.model.'):]: v for k, v in esd.items() if k.startswith('base.model.')}, 31)
    head = W.fill_state_dict(esd, 33)
    for k in esd:
        esd[k] = base[k[len('base.model.'):]] if k.startswith('base.model.') else head[k]
    E.load_state_dict(esd)
    IP = N.AlexNetFeatureRef(3, 'None')
    IP.load_state_dict(W.fill_state_dict(IP.state_dict(), 40))
    for net in (G, D, E, IP):
        net.to(dtype)
    return S.WSGANCycleStepRef(G, D, E, IP, fineSize_E=64, fineSize_IP=64, attr_mean=[35.0], attr_std=[20.0])


def cycle_inputs(it):
    A = W.seeded_tensor((4, 3, 32, 32), 700 + it)
    attr = (W.seeded_tensor((4, 1, 1, 1), 800 + it) + 1.0) * 30.0
    return A, attr


def check_against_golden(gold, p, losses, tensors, grads, nets, tol_loss=2e-5, tol_t=2e-5, tol_g=2e-3, tol_after=1e-4):
    names = list(gold['loss_names'])
    for i, n in enumerate(names):
        ref = gold[p + '/losses'][i]
        assert abs(losses[n] - ref) <= tol_loss * max(1.0, abs(ref)), 'loss %s: %r vs %r' % (n, losses[n], ref)
    for k, t in tensors.items():
        assert_close(t, torch.from_numpy(gold['%s/%s' % (p, k)]), tol_t, k)
    for tag, gd in grads.items():
        for k, g in gd.items():
            if g is None:
                continue
            st = gold['%s/grad%s/stat/%s' % (p, tag, k)]
            l2 = float(g.double().pow(2).sum().sqrt())
            assert abs(l2 - st[2]) <= tol_g * st[2] + 1e-6, 'grad%s %s l2 %g vs %g' % (tag, k, l2, st[2])
    for tag, net in nets.items():
        for k, v in net.state_dict().items():
            ref = gold['%s/after%s/%s' % (p, tag, k)]
            assert abs(float(v.double().abs().sum()) - ref[1]) <= tol_after * (ref[1] + 1e-3), 'after-step %s %s' % (tag, k)


def test_cycle_step_matches_reference():
    torch.set_num_threads(4)
    gold = np.load(os.path.join(GOLD, 'cycle_step.npz'))
    m = build_cycle_oracle()
    for it in range(2):
        A, attr = cycle_inputs(it)
        torch.manual_seed(4321 + it)
        m.set_input(A, attr)
        m.optimize_parameters()
        check_against_golden(gold, 'it%d' % it, m.losses(),
                             {k: getattr(m, k) for k in ('fake_x', 'rec_x', 'fake_y', 'rec_y', 'real_y')},
                             m.grads, {'G': m.netG, 'D': m.netD, 'E': m.netE})
