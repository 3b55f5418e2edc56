"""Pins the ORACLE (oracle/*.py, our CPU restatement) against golden vectors captured from the
reference itself (oracle/make_golden.py imported /root/reference in the build container).
CPU only; no reference access at test time.

Same torch, same ops, same order => the restatement should agree with the reference to
float32 rounding; the tolerance below (1e-5 relative) only allows for thread-count dependent
reduction order inside oneDNN.
"""
import os

import numpy as np
import pytest
import torch

from oracle import networks_ref as N
from oracle import step_ref as S
from oracle import weights as W
from util_cmp import assert_close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-5


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _check_net(net, inputs, seed_dy, gold, prefix, full):
    xs = [i.clone().requires_grad_(True) for i in inputs]
    y = net(*xs)
    ys = list(y) if isinstance(y, (tuple, list)) else [y]
    dys = [W.seeded_normal(tuple(o.shape), seed_dy + j) for j, o in enumerate(ys)]
    torch.autograd.backward(ys, dys)
    for j, o in enumerate(ys):
        assert_close(o, torch.from_numpy(gold['%s/out%d' % (prefix, j)]), TOL, '%s out%d' % (prefix, j))
    for j, x in enumerate(xs):
        key = '%s/din%d' % (prefix, j)
        if key in gold.files:
            assert_close(x.grad, torch.from_numpy(gold[key]), 5 * TOL, key)
    for k, p in net.named_parameters():
        if full:
            assert_close(p.grad, torch.from_numpy(gold['%s/dparam/full/%s' % (prefix, k)]), 1e-4, prefix + ' d' + k, atol=2e-5)
        else:
            samp = p.grad.reshape(-1)[::97]
            assert_close(samp, torch.from_numpy(gold['%s/dparam/samp/%s' % (prefix, k)]), 1e-4, prefix + ' d' + k, atol=2e-5)
    for k, b in net.named_buffers():
        if 'running' in k:
            a = b.double().numpy()
            ref = gold['%s/buf/%s' % (prefix, k)]
            assert abs(a.sum() - ref[0]) <= 1e-5 * (abs(ref[1]) + 1), prefix + ' buffer ' + k


@pytest.mark.parametrize('nb', [2, 9])
def test_generator_matches_reference(nb):
    gold = _load('nets.npz')
    g = N.ResnetGeneratorRef(3, 3, 1, 8, 'instance', nb)
    g.load_state_dict(W.fill_state_dict(g.state_dict(), 10 + nb))
    x = W.seeded_tensor((2, 3, 16, 16), 100 + nb)
    z = W.seeded_normal((2, 1, 1, 1), 200 + nb)
    _check_net(g, [x, z], 300 + nb, gold, 'G%d' % nb, True)
    with torch.no_grad():
        assert_close(g(x, z[:1]), torch.from_numpy(gold['G%d/out_zbroadcast' % nb]), TOL, 'z broadcast')


def test_discriminator_matches_reference():
    gold = _load('nets.npz')
    d = N.NLayerDiscriminatorRef(3, 1, 8, 3, 'batch', True)
    d.load_state_dict(W.fill_state_dict(d.state_dict(), 20))
    _check_net(d, [W.seeded_tensor((3, 3, 32, 32), 101), W.seeded_normal((3, 1, 1, 1), 201)], 301, gold, 'D', True)


@pytest.mark.parametrize('noisy', [False, True])
def test_encoder_matches_reference(noisy):
    gold = _load('nets.npz')
    e = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, noisy)
    e.load_state_dict(W.fill_state_dict(e.state_dict(), 30))
    _check_net(e, [W.seeded_tensor((3, 3, 64, 64), 102)], 302, gold, 'E_noisy%d' % int(noisy), False)


def test_alexnet_feature_matches_reference():
    gold = _load('nets.npz')
    ip = N.AlexNetFeatureRef(3, 'None')
    ip.load_state_dict(W.fill_state_dict(ip.state_dict(), 40))
    _check_net(ip, [W.seeded_tensor((2, 3, 64, 64), 103)], 303, gold, 'IP', False)


def build_oracle_step(variant, extra_args=()):
    """The oracle-side twin of make_golden.golden_steps for one variant."""
    from oracle.make_golden import STEP_VARIANTS
    extra = list(STEP_VARIANTS[variant]) + list(extra_args)
    kv = {}
    i = 0
    while i < len(extra):
        k = extra[i].lstrip('-')
        if i + 1 < len(extra) and not extra[i + 1].startswith('--'):
            kv[k] = extra[i + 1]
            i += 2
        else:
            kv[k] = True
            i += 1
    noisy = str(kv.get('noisy', 'false')).lower() == 'true'
    bayes = str(kv.get('bayesian', 'false')).lower() == 'true'
    drop = float(kv.get('bnn_dropout', 0.0))
    G = N.ResnetGeneratorRef(3, 3, 1, 8, 'instance', 9)
    G.load_state_dict(W.damp_generator_head(W.fill_state_dict(G.state_dict(), 19)))
    D = N.NLayerDiscriminatorRef(3, 1, 8, 3, kv.get('norm_D', 'batch'), True)
    D.load_state_dict(W.fill_state_dict(D.state_dict(), 20))
    E = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18', drop), 'avg', (32, 1), 1, 0.7, noisy, drop)
    E.load_state_dict(W.fill_state_dict(E.state_dict(), 30))
    IP = N.AlexNetFeatureRef(3, 'None')
    IP.load_state_dict(W.fill_state_dict(IP.state_dict(), 40))
    opts = dict(fineSize_E=64, fineSize_IP=64, embedding_mean=[0.1], embedding_std=[0.8], noisy=noisy,
                bayesian=bayes, noisy_var_type=kv.get('noisy_var_type', ''), bnn_T=int(kv.get('bnn_T', 10)),
                lambda_L1=float(kv.get('lambda_L1', 0.0)), lambda_IP=float(kv.get('lambda_IP', 1.0)),
                lambda_z=float(kv.get('lambda_z', 1.0)), lambda_A_GAN=float(kv.get('lambda_A_GAN', 0.0)),
                use_real_A=bool(kv.get('use_real_A', False)), detach_fake_B=bool(kv.get('detach_fake_B', False)),
                lr_E=float(kv.get('lr_E', 0.0)))
    return S.WSGANEmbStepRef(G, D, E, IP, **opts)


def step_inputs(it):
    A = W.seeded_tensor((4, 3, 32, 32), 500 + it)
    B = W.seeded_tensor((4, 3, 32, 32), 600 + it)
    label = [0, 2, 2, 0] if it == 0 else [2, 0, 1, 0]
    return A, B, label


def oracle_set_input(m, variant, it, dtype=torch.float32):
    """feed iteration `it` of a step variant to an oracle step object the way make_golden fed the reference"""
    from oracle.make_golden import step_batch, NP_SEED
    b = step_batch(variant, it)
    if variant == 'no_mixed_label_D':
        np.random.seed(NP_SEED + it)     # the label is drawn from numpy's global generator
        m.set_input_no_mixed({k: (v.to(dtype) if isinstance(v, torch.Tensor) else v) for k, v in b.items()})
    else:
        m.set_input(b['A'].to(dtype), b['B'].to(dtype), [int(v) for v in b['label']])


@pytest.mark.parametrize('variant', ['default', 'noisy_a', 'bayesian_e', 'bayesian_noisy_ae', 'use_real_A',
                                     'lambda_A_GAN', 'detach_fake_B', 'no_ip_no_z', 'no_mixed_label_D', 'norm_D_instance'])
def test_step_matches_reference(variant):
    torch.set_num_threads(4)
    gold = _load('step_%s.npz' % variant)
    m = build_oracle_step(variant)
    names = list(gold['loss_names'])
    for it in range(2):
        torch.manual_seed(1234 + it)     # same CPU random stream as the reference run
        oracle_set_input(m, variant, it)
        m.optimize_parameters()
        p = 'it%d' % it
        if 'it0/label_AB' in gold.files:
            assert [int(v) for v in m.label_AB] == [int(v) for v in gold[p + '/label_AB']], 'label of the batch'
        got = m.losses()
        for i, n in enumerate(names):
            ref = gold[p + '/losses'][i]
            assert abs(got[n] - ref) <= 2e-5 * max(1.0, abs(ref)), '%s loss %s: %r vs %r' % (variant, n, got[n], ref)
        for k in ('fake_B', 'rec_A', 'embedding_A', 'embedding_B', 'y_A', 'y_B'):
            assert_close(getattr(m, k), torch.from_numpy(gold['%s/%s' % (p, k)]), 2e-5, '%s %s' % (variant, k))
        for tag, grads in (('G', m.grads_G), ('D', m.grads_D)):
            for k, g in grads.items():
                if g is None:
                    continue
                st = gold['%s/grad%s/stat/%s' % (p, tag, k)]
                l2 = float(g.double().pow(2).sum().sqrt())
                # IN-cancelled biases have a true gradient of 0: only fp32 noise there
                assert abs(l2 - st[2]) <= 2e-3 * st[2] + 1e-6, '%s grad%s %s l2 %g vs %g' % (variant, tag, k, l2, st[2])
                if variant == 'default':
                    full = torch.from_numpy(gold['%s/grad%s/full/%s' % (p, tag, k)])
                    if full.abs().max() > 1e-5:
                        assert_close(g, full, 2e-3, 'default grad%s %s' % (tag, k))
        for tag, net in (('G', m.netG), ('D', m.netD)):
            for k, v in net.state_dict().items():
                ref = gold['%s/after%s/%s' % (p, tag, k)]
                a = v.double()
                assert abs(float(a.abs().sum()) - ref[1]) <= 1e-4 * (ref[1] + 1e-3), 'after-step %s %s' % (tag, k)


def test_lr_E_step_matches_the_reference_run_with_the_defined_optimizer():
    """a1 / a4 / a6, `--lr_E > 0`: update_G_and_E, backward_GE, backward_G_alone.  The golden vectors are the reference's OWN code for
    that branch, run on torch 2.10 with torch.optim.Adam replaced by oracle.step_ref.AdamThroughData while it built its optimizers
    (make_golden.golden_step_lr_E): the optimizer's semantics are defined by the oracle, the branch's algorithm is pinned here."""
    from oracle.make_golden import LR_E_ARGS
    torch.set_num_threads(4)
    gold = _load('step_lr_E.npz')
    m = build_oracle_step('default', LR_E_ARGS)
    assert isinstance(m.optimizer_E, S.AdamThroughData)
    names = list(gold['loss_names'])
    for it in range(2):
        torch.manual_seed(1234 + it)
        oracle_set_input(m, 'default', it)
        m.optimize_parameters()
        p = 'it%d' % it
        got = m.losses()
        for i, n in enumerate(names):
            ref = gold[p + '/losses'][i]
            assert abs(got[n] - ref) <= 2e-5 * max(1.0, abs(ref)), 'lr_E loss %s: %r vs %r' % (n, got[n], ref)
        for k in ('fake_B', 'rec_A', 'embedding_A', 'embedding_B', 'y_A', 'y_B'):
            assert_close(getattr(m, k), torch.from_numpy(gold['%s/%s' % (p, k)]), 2e-5, 'lr_E %s' % k)
        for tag, grads in (('gradG', m.grads_G), ('gradE', m.grads_E), ('gradG_alone', m.grads_G_alone), ('gradD', m.grads_D)):
            checked = 0
            for k, g in grads.items():
                if g is None:
                    continue
                st = gold['%s/%s/stat/%s' % (p, tag, k)]
                l2 = float(g.double().pow(2).sum().sqrt())
                assert abs(l2 - st[2]) <= 2e-3 * st[2] + 1e-6, 'lr_E %s %s l2 %g vs %g' % (tag, k, l2, st[2])
                if tag == 'gradG_alone':
                    full = torch.from_numpy(gold['%s/%s/full/%s' % (p, tag, k)])
                    if full.abs().max() > 1e-5:
                        assert_close(g, full, 2e-3, 'lr_E %s %s' % (tag, k))
                checked += 1
            assert checked > 0, tag
        for tag, net in (('G', m.netG), ('D', m.netD), ('E', m.netE)):
            for k, v in net.state_dict().items():
                ref = gold['%s/after%s/%s' % (p, tag, k)]
                assert abs(float(v.double().abs().sum()) - ref[1]) <= 1e-4 * (ref[1] + 1e-3), 'lr_E after-step %s %s' % (tag, k)


def test_get_current_visuals_matches_reference():
    """a16: G on real_A[0:1] per fixed rating bin in TRAIN mode -- images and the moved InstanceNorm running statistics"""
    torch.set_num_threads(4)
    gold = _load('visuals.npz')
    m = build_oracle_step('default')
    torch.manual_seed(1234)
    oracle_set_input(m, 'default', 0)
    m.forward()          # (no optimizer step in front: see make_golden.golden_visuals)
    for k, v in m.netG.state_dict().items():
        if 'running' in k:
            assert_close(v, torch.from_numpy(gold['before/' + k]), 1e-5, 'before ' + k, atol=1e-7)
    vis = m.get_current_visuals([-1.0, 0.0, 1.5])
    assert list(vis.keys()) == [str(n) for n in gold['names']]
    for k, v in vis.items():
        assert_close(v, torch.from_numpy(gold['vis/' + k]), 2e-5, 'visual ' + k)
    moved = 0
    for k, v in m.netG.state_dict().items():
        if 'running' in k:
            assert_close(v, torch.from_numpy(gold['after/' + k]), 1e-5, 'after ' + k, atol=1e-7)
            moved += int(not np.allclose(gold['after/' + k], gold['before/' + k]))
    assert moved > 0, 'the visuals pass must move the running statistics (train mode)'
    assert all(p.requires_grad for p in m.netG.parameters()) and bool(gold['requires_grad_after'].all())


def test_integer_helpers_bit_exact():
    g = _load('ints.npz')
    bins = [float(b) for b in g['bins']]
    got = [S.get_attr_label(float(a), bins) for a in g['attrs']]
    assert got == [int(v) for v in g['labels']]
    assert (S.get_attr_label(3.0, [5]) is None) == bool(g['short_is_none'])
    for s, r in zip(g['strs'], g['parsed']):
        assert repr(S.str2list(str(s))) == str(r)
    assert S.relabel([0, 1, 0], [0, 2, 1, 1]) == [0, 0, 1, 1]


def test_lr_schedule_matches_reference():
    g = _load('lr_schedule.npz')
    niter, nd, ec, base = int(g['niter']), int(g['niter_decay']), int(g['epoch_count']), float(g['base_lr'])
    for epoch, lr in enumerate(g['lr']):
        assert abs(base * S.lambda_lr(epoch, ec, niter, nd) - lr) < 1e-12
