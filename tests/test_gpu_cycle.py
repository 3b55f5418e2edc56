"""GPU parity of the wsgan_cycle step: WSGANCycleModel.optimize_parameters() on the HIP path,
built through the unchanged option parser, against the reference's golden vectors (iteration 0: identical weights)
and against the oracle's float64 twin for the gradients (unconditional D, ResNet-18 encoder TRAINED: weight gradients
through train-mode BatchNorm, max pooling and the residual trunk).

Tolerances: losses 2e-4, images / attributes 2e-4 of the largest magnitude; every gradient tensor by relative L2
against the float64 twin, |hip - g64| <= 2 |ref32 - g64| + 3e-2 |g64| (ReLU / max-pool mask flips between fp32 and fp64
runs move single entries, see test_gpu_nets.py) + 1e-6 absolute for the tensors whose true gradient is 0."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import networks_ref as N
from oracle import weights as W
from test_cycle_oracle_golden import build_cycle_oracle, cycle_inputs
from util_cmp import assert_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def build_hip_cycle(tmp_path):
    from pcgan_amd.options.train_options import TrainOptions
    from pcgan_amd.models import create_model
    base = N.ResNetFeatureRef('resnet18')
    base_path = str(tmp_path / 'resnet18_base.pth')
    torch.save(W.fill_state_dict(base.model.state_dict(), 31), base_path)
    ip = N.AlexNetFeatureRef(3, 'None')
    ip_path = str(tmp_path / 'IP.pth')
    torch.save(W.fill_state_dict(ip.state_dict(), 40), ip_path)
    argv = ['train.py', '--dataroot', 'synthetic', '--model', 'wsgan_cycle', '--name', 'g_cycle',
            '--checkpoints_dir', str(tmp_path), '--gpu_ids', '0', '--which_model_netG', 'resnet_9blocks',
            '--which_model_netD', 'n_layers', '--n_layers_D', '3', '--ngf', '8', '--ndf', '8',
            '--fineSize', '32', '--loadSize', '32', '--fineSize_E', '64', '--fineSize_IP', '64',
            '--batchSize', '4', '--pretrained_model_path_E', base_path, '--pretrained_model_path_IP', ip_path,
            '--display_id', '-1', '--attr_bins', '[10, 30, 50]', '--attr_mean', '35.0', '--attr_std', '20.0']
    old, sys.argv = sys.argv, argv
    try:
        opt = TrainOptions().parse()
    finally:
        sys.argv = old
    model = create_model(opt)
    model.setup(opt)
    model.netG.load_state_dict(W.damp_generator_head(W.fill_state_dict(model.netG.state_dict(), 19)))
    model.netD.load_state_dict(W.fill_state_dict(model.netD.state_dict(), 20))
    esd = model.netE.state_dict()
    filled = W.fill_state_dict({k: v.cpu() for k, v in esd.items()}, 33)
    model.netE.load_state_dict({k: (filled[k] if k.startswith('cnn') else v) for k, v in esd.items()})
    return model, opt


def _rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_cycle_step_matches_reference_and_oracle(tmp_path, dev):
    gold = np.load(os.path.join(GOLD, 'cycle_step.npz'))
    names = list(gold['loss_names'])
    model, opt = build_hip_cycle(tmp_path)
    assert model.loss_names == names and model.model_names == ['G', 'E', 'D']
    oracle = build_cycle_oracle()
    twin = build_cycle_oracle(torch.float64)
    grabbed = {}
    for tag, optim, net in (('G', model.optimizer_G, model.netG), ('D', model.optimizer_D, model.netD),
                            ('E', model.optimizer_E, model.netE)):
        orig = optim.step

        def stepper(orig=orig, tag=tag, net=net):
            grabbed[tag] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
            return orig()
        optim.step = stepper

    A, attr = cycle_inputs(0)
    oracle.set_input(A, attr)
    oracle.optimize_parameters()
    twin.set_input(A.double(), attr.double())
    twin.optimize_parameters()
    model.set_input({'A': A, 'B_attr': attr, 'A_paths': ['a'] * 4, 'B_paths': ['b'] * 4})
    model.optimize_parameters()

    got = model.get_current_losses()
    for i, n in enumerate(names):
        ref = gold['it0/losses'][i]
        assert abs(got[n] - ref) <= 2e-4 * max(1.0, abs(ref)), 'loss %s: hip %.7g reference %.7g' % (n, got[n], ref)
    for k in ('fake_x', 'rec_x', 'fake_y', 'rec_y', 'real_y'):
        assert_close(getattr(model, k), torch.from_numpy(gold['it0/' + k]), 2e-4, k + ' vs reference')
    for tag in ('G', 'D', 'E'):
        for k, g64 in twin.grads[tag].items():
            if g64 is None:
                continue
            hip, ref32 = grabbed[tag][k], oracle.grads[tag][k]
            scale = float(g64.norm())
            e_hip = float((hip.double().cpu() - g64).norm())
            e_ref = float((ref32.double() - g64).norm())
            assert e_hip <= 2 * e_ref + 3e-2 * scale + 1e-6, \
                'grad%s %s: |hip-g64| %.3e, |ref32-g64| %.3e, |g64| %.3e' % (tag, k, e_hip, e_ref, scale)
            st = gold['it0/grad%s/stat/%s' % (tag, k)]
            assert abs(float(hip.double().norm()) - st[2]) <= 3e-2 * st[2] + 1e-4, 'grad%s %s norm vs reference' % (tag, k)   # (+1e-4: tensors whose true gradient is 0 carry only noise)
    # parameters after the three Adam steps.  The first Adam step moves every parameter by lr * sign(gradient): where
    # the true gradient is 0 (biases in front of an affine-less InstanceNorm) the sign is noise, so each entry may
    # differ from the reference by up to 2 lr -- bound the |.|-sum accordingly.
    for tag, net in (('G', model.netG), ('D', model.netD), ('E', model.netE)):
        for k, v in net.state_dict().items():
            ref = gold['it0/after%s/%s' % (tag, k)]
            slack = 2e-3 * (ref[1] + 1e-3) + 2.02 * opt.lr * v.numel()
            assert abs(float(v.double().abs().sum()) - ref[1]) <= slack, 'after-step %s %s' % (tag, k)

    # second iteration keeps running (now from slightly different weights: Adam turns gradient noise into +-lr moves)
    A, attr = cycle_inputs(1)
    model.set_input({'A': A, 'B_attr': attr, 'A_paths': ['a'] * 4, 'B_paths': ['b'] * 4})
    model.optimize_parameters()
    got = model.get_current_losses()
    for i, n in enumerate(names):
        ref = gold['it1/losses'][i]
        assert abs(got[n] - ref) <= 5e-2 * max(1.0, abs(ref)), 'it1 loss %s: hip %.6g reference %.6g' % (n, got[n], ref)
    vis = model.get_current_visuals()
    assert list(vis.keys())[:3] == ['real_x', 'fake_x', 'rec_x'] and 'attr_2' in vis
