"""BASELINE.json configs 4 and 5 at their spatial size (256x256) with the FULL networks, against the oracle's CPU step.

  config 4  wsgan_emb, `--noisy true --bayesian true --bnn_dropout 0.2 --noisy_var_type ae`, bnn_T 10: MC-dropout Elo
            encoder (10 stochastic passes per rating: 2 x 10 in forward, 10 more in backward_G), heteroscedastic
            z_rec term, resampled ratings for D (reference models/wsgan_emb_model.py:214-259, 418-428;
            util/util.py:153-171).  The Dropout2d keep-masks and the resample draws of the oracle run are replayed
            on the GPU (a GPU RNG stream cannot match a CPU one).
  config 5  wsgan_cycle: unconditional D, ResNet-18 encoder (max pooling, head [64, 1]) TRAINED together with G,
            D updated first (reference models/wsgan_cycle_model.py:166-256).

Full size = 9-block ResnetGenerator ngf 64, 3-layer PatchGAN ndf 64, ResNet-18 at 224, AlexNet at 224, images 256x256;
the batch is 2 so that the oracle's CPU step finishes in well under a minute.  Compared: every loss, the generated
images, ratings, the BatchNorm / InstanceNorm running statistics after the step, and every G / D (/ E) gradient tensor.

Tolerances: losses 2e-4 (heteroscedastic z_rec: 2e-3, it divides by an MC variance); images / ratings 2e-4 of the largest
magnitude; running statistics 1e-3; gradients by relative L2 against the fp32 oracle, 2e-2 (no float64 twin at this size;
ReLU / max-pool decisions flip between implementations, see test_gpu_nets.py -- the sharp gradient checks are the small
fixtures of test_gpu_step.py / test_gpu_cycle.py; here an indexing error at 64x64 / 128x128 planes would show as O(1)).
"""
import os
import sys

import pytest
import torch

from oracle import networks_ref as N
from oracle import step_ref as S
from util_cmp import assert_close

pytestmark = pytest.mark.gpu

SIZE = 256
BATCH = 2


def _parse(argv):
    from pcgan_amd.options.train_options import TrainOptions
    from pcgan_amd.models import create_model
    old, sys.argv = sys.argv, argv
    so, sys.stdout = sys.stdout, open(os.devnull, 'w')
    try:
        opt = TrainOptions().parse()
        model = create_model(opt)
        model.setup(opt)
    finally:
        sys.stdout.close()
        sys.argv, sys.stdout = old, so
    return model, opt


def _common(tmp, name):
    return ['train.py', '--dataroot', 'synthetic', '--name', name, '--checkpoints_dir', tmp, '--gpu_ids', '0',
            '--which_model_netG', 'resnet_9blocks', '--which_model_netD', 'n_layers', '--n_layers_D', '3',
            '--fineSize', str(SIZE), '--loadSize', str(SIZE), '--batchSize', str(BATCH), '--display_id', '-1',
            '--pretrained_model_path_IP', os.path.join(tmp, 'IP.pth')]


def _rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _grab(model, tags):
    grabbed = {}
    for tag in tags:
        optim, net = getattr(model, 'optimizer_' + tag), getattr(model, 'net' + tag)
        orig = optim.step

        def stepper(orig=orig, tag=tag, net=net):
            grabbed[tag] = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
            return orig()
        optim.step = stepper
    return grabbed


def _check_grads(tag, hip, ref, tol=3e-2):
    """This is the LOOSE band: the oracle differentiates on its OWN ReLU / max-pool decisions, so the distance is that of a handful of
    flipped decisions (every generator tensor sits at the same 1.6e-2 .. 2.2e-2), not of the kernels; the sharp 5e-4 checks with the
    decisions replayed are tests/test_gpu_nets.py / test_gpu_step.py.  The band follows the distance between the HIP forward pass and
    the CPU fp32 forward pass: with the generator stem on the f16 matrix pipe (csrc/thin_conv.hip) its output is CLOSER to float64
    (1.7e-7 against 2.5e-7 for the fp32-MFMA kernel and for the CPU) but further from the CPU's own rounding (3.0e-7 against 1.4e-7,
    scripts/acc_thin.py), so a few more decisions flip: 1.58e-2 -> 2.1e-2 on this fixture (PCGAN_THIN_MASK isolates it to the stem)."""
    for k, og in ref.items():
        if og is None:
            continue
        scale = float(og.abs().max())
        sibling = ref.get(k[:-4] + 'weight') if k.endswith('.bias') else None
        if sibling is not None and scale <= 1e-4 * float(sibling.abs().max()):
            # bias in front of an affine-less InstanceNorm / a BatchNorm: true gradient 0, noise on both sides
            assert float(hip[k].abs().max()) <= 1e-3 * float(sibling.abs().max()) + 1e-6, 'grad%s %s should be ~0' % (tag, k)
            continue
        hg = hip[k]
        if tag == 'G' and k == 'model.1.weight':       # the rating channel's filter slice: true gradient 0
            hg, og = hg[:, :-1], og[:, :-1]
        if scale < 1e-7:
            assert float(hg.abs().max()) < 1e-5, 'grad%s %s should be ~0' % (tag, k)
            continue
        e = _rel_l2(hg, og)
        # a bias gradient is ONE signed sum over N*H*W = 131072 pixels per channel (heavy cancellation, 3 values for the
        # generator head): its fp32 value depends on the summation order on either side; the weights of the same layer are
        # the sharp check
        t = 1e-1 if k.endswith('.bias') else tol
        if os.environ.get('PCGAN_TEST_VERBOSE'):
            print('grad%s %-40s rel L2 %.3e' % (tag, k, e))
        assert e <= t, 'grad%s %s: relative L2 against the oracle %.3e > %.1e (max |g| %.3e)' % (tag, k, e, t, scale)


def _check_buffers(tag, hip_net, ref_net, tol=1e-3):
    rsd = ref_net.state_dict()
    for k, v in hip_net.state_dict().items():
        if k.endswith('num_batches_tracked'):
            assert int(v) == int(rsd[k]), '%s %s' % (tag, k)
        elif 'running' in k:
            assert_close(v, rsd[k], tol, '%s %s after the step' % (tag, k), atol=1e-5)


@pytest.fixture
def production_route(monkeypatch):
    """configs 4 / 5 run at batch 8 / 16 (32768 / 65536 output pixels per residual convolution: the matrix-pipe window / split
    kernels at image width 64); the batch of 2 used here would stay under the host's routing threshold -- lower it so the same
    kernels are under test.  Both tests REQUEST this fixture by name and assert afterwards (`_assert_production_route`) that
    the residual convolutions really went that way and that their operand maxima came from the norm kernels."""
    from pcgan_amd.hip import ops
    monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    assert ops._plan(ops._L.PASS_FWD, BATCH, 256, 64, 64, 256, 3, 3, 1, 1, 1, ops.F32).route == 'hsplit'
    assert ops._plan(ops._L.PASS_BWD_DATA, BATCH, 256, 64, 64, 256, 3, 3, 1, 1, 1, ops.F32).route == 'hsplit'
    assert ops._plan(ops._L.PASS_BWD_WEIGHT, BATCH, 256, 64, 64, 256, 3, 3, 1, 1, 1, ops.F32).route == 'hsplit'
    return dict(ops.ROUTE_STATS), dict(ops.AMAX_STATS)


def _assert_production_route(before, g_passes):
    """the 18 residual convolutions of every generator pass ran forward, data gradient AND weight gradient on the window /
    split kernels of the fp16 route (`hsplit`: bsplit_halo_kernel<..., 64>, hsplit_wgrad_kernel), with the operand maxima handed
    over by the instance-norm kernels"""
    from pcgan_amd.hip import ops
    routes0, amax0 = before
    for pass_ in ('fwd', 'dgrad', 'wgrad'):
        n = ops.ROUTE_STATS.get((pass_, 'hsplit'), 0) - routes0.get((pass_, 'hsplit'), 0)
        assert n >= 18 * g_passes, 'residual convolutions on the hsplit route, %s: %d < %d (%s)' % (pass_, n, 18 * g_passes, dict(ops.ROUTE_STATS))
    attached = ops.AMAX_STATS['attached'] - amax0['attached']
    assert attached >= 18 * g_passes * 4, ops.AMAX_STATS


def test_config4_bayesian_noisy_256(tmp_path, dev, production_route):
    from pcgan_amd.hip import nn as hnn
    from pcgan_amd.models import networks
    from pcgan_amd.util import util as hutil
    import bench
    tmp = str(tmp_path)
    torch.manual_seed(11)
    e = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7, noisy=True, bnn_dropout=0.2)
    torch.save(e.state_dict(), os.path.join(tmp, 'E.pth'))
    torch.save(networks.define_IP('alexnet', 3).state_dict(), os.path.join(tmp, 'IP.pth'))
    model, opt = _parse(_common(tmp, 'c4') + ['--model', 'wsgan_emb', '--noisy', 'true', '--bayesian', 'true',
                                             '--bnn_dropout', '0.2', '--noisy_var_type', 'ae',
                                             '--pretrained_model_path_E', os.path.join(tmp, 'E.pth')])
    assert opt.bnn_T == 10 and opt.fineSize_E == 224
    G = N.ResnetGeneratorRef(3, 3, 1, 64, 'instance', 9)
    D = N.NLayerDiscriminatorRef(3, 1, 64, 3, 'batch', True)
    E = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18', 0.2), 'avg', (32, 1), 1, 0.7, True, 0.2)
    IP = N.AlexNetFeatureRef(3, 'None')
    for ref, net in ((G, model.netG), (D, model.netD), (E, model.netE), (IP, model.netIP)):
        ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    oracle = S.WSGANEmbStepRef(G, D, E, IP, noisy=True, bayesian=True, noisy_var_type='ae', bnn_T=10)
    b = bench.synthetic_batch(BATCH, SIZE, 4)
    # oracle on the CPU; every keep-mask (3 x 10 encoder passes x 18 dropout sites) and resample draw is recorded
    N.Dropout2dRec.record = []
    oracle.draws = []
    torch.manual_seed(99)
    oracle.set_input(b['A'], b['B'], [int(v) for v in b['label']])
    oracle.optimize_parameters()
    masks, N.Dropout2dRec.record = N.Dropout2dRec.record, None
    assert len(masks) > 0 and len(oracle.draws) == 2
    grabbed = _grab(model, 'GD')
    hnn.Dropout2d.mask_source = iter(masks)
    hutil.inject_noise(iter(oracle.draws))
    try:
        model.set_input(b)
        model.optimize_parameters()
        torch.cuda.synchronize()
        assert next(hnn.Dropout2d.mask_source, None) is None, 'the HIP step consumed fewer dropout masks than the oracle drew'
    finally:
        hnn.Dropout2d.mask_source = None
        hutil.inject_noise(None)
    got, want = model.get_current_losses(), oracle.losses()
    for k, v in want.items():
        tol = 2e-3 if k == 'z_rec' else 2e-4
        assert abs(got[k] - v) <= tol * max(1.0, abs(v)), 'config 4 loss %s: hip %.7g oracle %.7g' % (k, got[k], v)
    for k in ('fake_B', 'rec_A', 'y_A', 'y_B', 'embedding_A', 'embedding_B', 'resample_A', 'resample_B'):
        assert_close(getattr(model, k), getattr(oracle, k).detach(), 2e-4, 'config 4 ' + k)
    assert tuple(model.fake_B.shape) == (BATCH, 3, SIZE, SIZE)
    _check_grads('G', grabbed['G'], oracle.grads_G)
    _check_grads('D', grabbed['D'], oracle.grads_D)
    for tag, hn, on in (('G', model.netG, G), ('D', model.netD, D), ('E', model.netE, E)):
        _check_buffers(tag, hn, on)
    _assert_production_route(production_route, 2)


def test_config5_cycle_256(tmp_path, dev, production_route):
    from pcgan_amd.models import networks
    tmp = str(tmp_path)
    torch.manual_seed(12)
    e = networks.define_E('resnet18', 3, 'normal', 'max', [64, 1], 1, 0.2)
    torch.save(e.base.model.state_dict(), os.path.join(tmp, 'base.pth'))
    torch.save(networks.define_IP('alexnet', 3).state_dict(), os.path.join(tmp, 'IP.pth'))
    model, opt = _parse(_common(tmp, 'c5') + ['--model', 'wsgan_cycle', '--attr_bins', '[10, 30, 50]', '--attr_mean', '35.0',
                                             '--attr_std', '20.0', '--pretrained_model_path_E', os.path.join(tmp, 'base.pth')])
    assert opt.pooling_E == 'max' and list(opt.cnn_dim_E) == [64, 1]
    G = N.ResnetGeneratorRef(3, 3, 1, 64, 'instance', 9)
    D = N.NLayerDiscriminatorRef(3, 0, 64, 3, 'batch', True)
    E = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'max', (64, 1), 1, opt.cnn_relu_slope_E, False)
    IP = N.AlexNetFeatureRef(3, 'None')
    for ref, net in ((G, model.netG), (D, model.netD), (E, model.netE), (IP, model.netIP)):
        ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    oracle = S.WSGANCycleStepRef(G, D, E, IP, attr_mean=[35.0], attr_std=[20.0])
    g = torch.Generator().manual_seed(5)
    A = torch.rand(BATCH, 3, SIZE, SIZE, generator=g) * 2 - 1
    attr = torch.rand(BATCH, 1, 1, 1, generator=g) * 60
    oracle.set_input(A, attr)
    oracle.optimize_parameters()
    grabbed = _grab(model, 'GDE')
    model.set_input({'A': A, 'B_attr': attr, 'A_paths': [''] * BATCH, 'B_paths': [''] * BATCH})
    model.optimize_parameters()
    torch.cuda.synchronize()
    got, want = model.get_current_losses(), oracle.losses()
    for k, v in want.items():
        assert abs(got[k] - v) <= 2e-4 * max(1.0, abs(v)), 'config 5 loss %s: hip %.7g oracle %.7g' % (k, got[k], v)
    for k in ('fake_x', 'rec_x', 'fake_y', 'rec_y', 'real_y'):
        assert_close(getattr(model, k), getattr(oracle, k).detach(), 2e-4, 'config 5 ' + k)
    assert tuple(model.fake_x.shape) == (BATCH, 3, SIZE, SIZE)
    for tag in 'GDE':
        _check_grads(tag, grabbed[tag], oracle.grads[tag])
    for tag, hn, on in (('G', model.netG, G), ('D', model.netD, D), ('E', model.netE, E)):
        _check_buffers(tag, hn, on)
    _assert_production_route(production_route, 2)
