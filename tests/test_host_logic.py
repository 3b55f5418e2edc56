"""CPU: host-side mirror of the reference interface -- option parser, model/dataset registries,
factories (construction only: forward needs the GPU), integer helpers against the reference's
golden vectors, schedulers, checkpoint key layout, fused-optimizer plumbing, loud failure off-GPU."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import networks_ref as N

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _parse(argv):
    from pcgan_amd.options.train_options import TrainOptions
    old, sys.argv = sys.argv, ['train.py'] + argv
    try:
        return TrainOptions().parse()
    finally:
        sys.argv = old


def test_wsgan_emb_option_defaults_match_reference(tmp_path):
    opt = _parse(['--dataroot', 'synthetic', '--model', 'wsgan_emb', '--gpu_ids', '-1', '--checkpoints_dir', str(tmp_path)])
    # reference models/wsgan_emb_model.py:67-78 and options/train_options.py:7-26
    assert opt.which_model_netG == 'unet_128' and opt.which_model_netD == 'n_layers' and opt.n_layers_D == 4
    assert opt.batchSize == 10 and opt.fineSize == 128 and opt.loadSize == 128 and opt.no_lsgan is True
    assert opt.norm_G == 'instance' and opt.norm_D == 'batch' and opt.pool_size == 0
    assert opt.lr == 0.0002 and opt.beta1 == 0.5 and opt.niter == 50 and opt.niter_decay == 50
    assert opt.lambda_IP == 1.0 and opt.lambda_z == 1.0 and opt.lambda_A == 0.5 and opt.lr_E == 0.0
    assert opt.relabel_D == [0, 1, 0] and opt.cnn_dim_E == [32, 1] and opt.cnn_relu_slope_E == 0.7
    assert opt.dataset_mode == 'wsgan_emb' and opt.gpu_ids == [] and opt.isTrain
    assert os.path.exists(os.path.join(str(tmp_path), opt.name, 'opt_train.txt'))
    assert os.path.exists(os.path.join(str(tmp_path), opt.name, 'cmd_train.txt'))


def test_model_registry():
    from pcgan_amd import models
    from pcgan_amd.models.wsgan_emb_model import WSGANEmbModel
    assert models.find_model_using_name('wsgan_emb') is WSGANEmbModel
    with pytest.raises(NotImplementedError):
        models.find_model_using_name('pix2pix')


def test_integer_helpers_bit_exact_vs_reference_golden():
    from pcgan_amd.util import util
    g = np.load(os.path.join(GOLD, 'ints.npz'))
    bins = [float(b) for b in g['bins']]
    assert [util.get_attr_label(float(a), bins) for a in g['attrs']] == [int(v) for v in g['labels']]
    assert (util.get_attr_label(3.0, [5]) is None) == bool(g['short_is_none'])
    for s, r in zip(g['strs'], g['parsed']):
        assert repr(util.str2list(str(s))) == str(r)
    for b, v in zip(g['bools'], g['bool_vals']):
        assert util.str2bool(str(b)) == bool(v)
    with pytest.raises(Exception):
        util.str2bool('maybe')


def test_lr_schedule_matches_reference_golden(tmp_path):
    from pcgan_amd.models import networks
    g = np.load(os.path.join(GOLD, 'lr_schedule.npz'))
    opt = _parse(['--dataroot', 'synthetic', '--model', 'wsgan_emb', '--gpu_ids', '-1', '--checkpoints_dir', str(tmp_path)])
    p = torch.nn.Parameter(torch.zeros(1))
    optim = torch.optim.Adam([p], lr=opt.lr)
    sched = networks.get_scheduler(optim, opt)
    for ref in g['lr']:
        assert abs(optim.param_groups[0]['lr'] - ref) < 1e-12
        optim.step()
        sched.step()


def test_state_dict_key_layouts_match_reference():
    from pcgan_amd.models import networks
    G = networks.define_G(3, 3, 1, 8, 'resnet_9blocks', norm='instance', init_type='normal')
    D = networks.define_D(3, 1, 8, 'n_layers', 3, 'batch', True, 'normal')
    E = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
    En = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7, noisy=True, bnn_dropout=0.2)
    IP = networks.define_IP('alexnet', 3)
    assert list(G.state_dict()) == list(N.ResnetGeneratorRef(3, 3, 1, 8, 'instance', 9).state_dict())
    assert list(D.state_dict()) == list(N.NLayerDiscriminatorRef(3, 1, 8, 3, 'batch', True).state_dict())
    assert list(E.state_dict()) == list(N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18')).state_dict())
    assert list(IP.state_dict()) == list(N.AlexNetFeatureRef().state_dict())
    # counts measured on the reference
    assert (len(G.state_dict()), len(D.state_dict()), len(E.state_dict()), len(En.state_dict())) == (117, 22, 129, 138)
    assert sum(p.numel() for p in networks.define_D(3, 1, 64, 'n_layers', 3, 'batch', True).parameters()) == 2766657
    assert G.state_dict()['model.19.weight'].shape == (32, 16, 3, 3)     # ConvTranspose: (Cin, Cout, 3, 3)


def test_full_size_parameter_counts():
    from pcgan_amd.models import networks
    G = networks.define_G(3, 3, 1, 64, 'resnet_9blocks', norm='instance', init_type='normal')
    E = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
    assert sum(p.numel() for p in G.parameters()) == 11381315
    assert sum(p.numel() for p in E.parameters()) == 11324353


def test_factories_error_behaviour():
    from pcgan_amd.models import networks
    for fn, args in ((networks.define_G, (3, 3, 1, 8, 'no_such_net')), (networks.define_D, (3, 1, 8, 'no_such_net')),
                     (networks.define_E, ('no_such_net',)), (networks.define_IP, ('no_such_net', 3)),
                     (networks.get_norm_layer, ('no_such_norm',))):
        with pytest.raises(NotImplementedError):
            fn(*args)
    with pytest.raises(NotImplementedError):
        networks.define_G(3, 3, 1, 8, 'unet_128')          # known to the reference, outside the hot path
    with pytest.raises(NotImplementedError):
        networks.init_weights(torch.nn.Conv2d(1, 1, 1), 'no_such_init')
    # resnet_<n>blocks for any n
    assert len([m for m in networks.define_G(3, 3, 1, 8, 'resnet_2blocks', norm='instance').model
                if isinstance(m, networks.ResnetBlock)]) == 2


def test_illegal_flag_combinations_raise(tmp_path):
    from pcgan_amd.models import create_model
    base = ['--dataroot', 'synthetic', '--model', 'wsgan_emb', '--gpu_ids', '-1', '--checkpoints_dir', str(tmp_path),
            '--which_model_netG', 'resnet_2blocks', '--ngf', '8']
    with pytest.raises(RuntimeError, match='Aleatoric'):
        create_model(_parse(base + ['--noisy_var_type', 'a']))
    with pytest.raises(RuntimeError, match='Epistemic'):
        create_model(_parse(base + ['--noisy', 'true', '--noisy_var_type', 'e']))


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: off-GPU the HIP modules fail loudly instead of computing something else."""
    from pcgan_amd.models import networks
    D = networks.define_D(3, 1, 8, 'n_layers', 3, 'batch', True, 'normal')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        D(torch.zeros(1, 3, 32, 32), torch.zeros(1, 1, 1, 1))
    from pcgan_amd.util.util import upsample2d
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        upsample2d(torch.zeros(1, 3, 8, 8), 16)
    x = torch.zeros(1, 3, 8, 8)
    assert upsample2d(x, 8) is x and upsample2d(x, 0) is x          # identity cases never touch the device


def test_product_never_imports_oracle_or_reference():
    root = os.path.join(os.path.dirname(GOLD), '..', 'pc-gan_amd')
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(d, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, f
                assert '/root/reference' not in text, f


def test_fused_adam_flat_buffer_plumbing():
    from pcgan_amd.hip.optim import FusedAdam
    from pcgan_amd.models import networks
    D = networks.define_D(3, 1, 8, 'n_layers', 3, 'batch', True, 'normal')
    before = {k: v.clone() for k, v in D.state_dict().items()}
    opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    for k, v in D.state_dict().items():
        assert torch.equal(v, before[k])                             # re-homing keeps the values
    lo, hi = opt.flat.data_ptr(), opt.flat.data_ptr() + opt.flat.numel() * 4
    for p in D.parameters():
        assert lo <= p.data_ptr() < hi and p.grad is not None and p.grad.shape == p.shape
    assert opt.numel == sum(p.numel() for p in D.parameters())
    # autograd accumulates INTO the flat gradient buffer
    w = D.model[0].weight
    (w * 2).sum().backward()
    assert float(opt.gflat.sum()) == 2.0 * w.numel()
    opt.zero_grad()
    assert float(opt.gflat.abs().sum()) == 0.0 and w.grad.data_ptr() >= opt.gflat.data_ptr()
    # load_state_dict copies in place: parameters stay views of the flat buffer
    D.load_state_dict(before)
    assert lo <= w.data_ptr() < hi
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        opt.step()


def test_gan_loss_target_vectors():
    from pcgan_amd.models import networks
    crit = networks.GANLoss(use_lsgan=False)
    pred = torch.zeros(3, 1, 2, 2)
    assert crit.get_target_vector(pred, True).tolist() == [1.0]
    assert crit.get_target_vector(pred, False).tolist() == [0.0]
    assert crit.get_target_vector(pred, [0, 1, 0]).tolist() == [0.0, 1.0, 0.0]
    assert crit.get_target_tensor(pred, [0, 1, 0]).shape == pred.shape
    assert crit.get_target_vector(pred, torch.tensor([1.0, 0.0, 1.0])).tolist() == [1.0, 0.0, 1.0]


def test_synthetic_dataset_dict_layout(tmp_path):
    from pcgan_amd.data import CreateDataLoader
    opt = _parse(['--dataroot', 'synthetic', '--model', 'wsgan_emb', '--gpu_ids', '-1', '--checkpoints_dir', str(tmp_path),
                  '--fineSize', '16', '--batchSize', '4', '--nThreads', '0'])
    batch = next(iter(CreateDataLoader(opt).load_data()))
    assert set(batch) == {'A', 'B', 'label', 'A_paths', 'B_paths'}
    assert batch['A'].shape == (4, 3, 16, 16) and batch['label'].dtype == torch.int64
    assert float(batch['A'].min()) >= -1 and float(batch['A'].max()) <= 1
    assert set(batch['label'].tolist()) <= {0, 2}


def test_declared_route_table_for_the_benchmark_layers():
    """hip/ops.py: ROUTE_TABLE (round 4: the nested conditions of _plan became an ordered table per pass).  The routes of config 2's
    layers at batch 32 are pinned here -- forward / data gradient / weight gradient -- so that a re-ordering of the table or a changed
    `*_supported` predicate in the library shows up on the CPU (the predicates are host code: no GPU needed)."""
    from pcgan_amd.hip import ops, lib as L
    ops._sentinel = lambda: None          # (no device word on the CPU)
    expect = {
        # name: (N, C, H, W, K, R, S, stride, pad, pad_mode) -> (fwd, dgrad, wgrad)
        'G.res': ((32, 256, 32, 32, 256, 3, 3, 1, 1, 1), ('hsplit', 'hsplit', 'hsplit')),
        'G.stem': ((32, 4, 128, 128, 64, 7, 7, 1, 3, 1), ('thin', 'packed', 'hsplit')),
        'G.down1': ((32, 64, 128, 128, 128, 3, 3, 2, 1, 0), ('hgemm', 'hgemm', 'hsplit')),
        'G.down2': ((32, 128, 64, 64, 256, 3, 3, 2, 1, 0), ('hgemm', 'hgemm', 'hsplit')),
        'G.head': ((32, 64, 128, 128, 3, 7, 7, 1, 3, 1), ('packed', 'thin', 'generic')),
        'D.c0': ((32, 4, 128, 128, 64, 4, 4, 2, 1, 0), ('thin', 'packed', 'generic')),
        'D.c3': ((32, 256, 16, 16, 512, 4, 4, 1, 1, 0), ('hgemm', 'hgemm', 'hsplit')),
        'D.c4': ((32, 512, 15, 15, 1, 4, 4, 1, 1, 0), ('packed', 'packed', 'generic')),
        'E.conv1': ((32, 3, 224, 224, 64, 7, 7, 2, 3, 0), ('thin', 'packed', 'generic')),
        'E.l1': ((32, 64, 56, 56, 64, 3, 3, 1, 1, 0), ('hgemm', 'hgemm', 'generic')),
        'E.l4': ((32, 512, 7, 7, 512, 3, 3, 1, 1, 0), ('hgemm', 'hgemm', 'generic')),
        'IP.c1': ((32, 3, 224, 224, 64, 11, 11, 4, 2, 0), ('packed', 'packed', 'generic')),
    }
    assert [r[0] for r in ops.ROUTE_TABLE[L.PASS_FWD]] == ['hsplit', 'hgemm', 'bsplit', 'thin', 'packed']
    assert [r[0] for r in ops.ROUTE_TABLE[L.PASS_BWD_DATA]] == ['hsplit', 'bsplit', 'hgemm', 'thin', 'packed']
    assert [r[0] for r in ops.ROUTE_TABLE[L.PASS_BWD_WEIGHT]] == ['hsplit', 'bsplit', 'generic']
    got = {}
    for name, (shape, want) in expect.items():
        got[name] = tuple(ops._plan(ps, *shape, L.F32).route for ps in (L.PASS_FWD, L.PASS_BWD_DATA, L.PASS_BWD_WEIGHT))
    wrong = {k: (got[k], expect[k][1]) for k in expect if got[k] != expect[k][1]}
    assert not wrong, wrong
    # bf16 tensors: the one-product forms
    assert tuple(ops._plan(ps, 32, 256, 32, 32, 256, 3, 3, 1, 1, 1, L.BF16).route for ps in (L.PASS_FWD, L.PASS_BWD_DATA, L.PASS_BWD_WEIGHT)) == \
        ('bsplit', 'bsplit', 'hsplit')
