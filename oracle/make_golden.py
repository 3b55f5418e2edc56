"""ORACLE fixture generator (test infrastructure).  Run ONCE in the build container:

    python oracle/make_golden.py            # writes tests/golden/*.npz

It IMPORTS the reference (read-only, /root/reference) on CPU, feeds it seeded inputs and the
deterministic weights of oracle/weights.py, and records what the reference computes.  Only
the resulting data (inputs/expected outputs) is committed; the reference source never leaves
the container and nothing in tests/, smoke() or bench.py reads /root/reference at run time.

`torchvision` is not installed here; the reference's model file imports
`torchvision.transforms` at module top without using it on this path, so two empty stub
modules are registered (an ordinary missing-module workaround).
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import weights as W  # noqa: E402


def import_reference():
    for name in ('torchvision', 'torchvision.transforms'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.path.insert(0, REF)
    import models.networks as ref_networks           # noqa: E402
    return ref_networks


def t2n(t):
    return t.detach().cpu().numpy()


def grads_summary(named_grads, prefix, out, full=True, stride=97):
    """full tensors for small nets; per-tensor (sum, abs-sum, l2) + strided sample otherwise."""
    for k, g in named_grads:
        if g is None:
            continue
        a = t2n(g).astype(np.float64)
        out['%s/stat/%s' % (prefix, k)] = np.array([a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())])
        if full:
            out['%s/full/%s' % (prefix, k)] = t2n(g)
        else:
            out['%s/samp/%s' % (prefix, k)] = t2n(g).reshape(-1)[::stride].copy()


def run_net(net, inputs, seed_dy, out, prefix, full_grads=True, fp64=False):
    """forward + backward of one net on seeded inputs; records outputs, input grads, param grads,
    buffers after the call."""
    if fp64:
        net = net.double()
        inputs = [i.double() for i in inputs]
    xs = [i.clone().requires_grad_(i.is_floating_point()) for i in inputs]
    y = net(*xs)
    ys = list(y) if isinstance(y, (tuple, list)) else [y]
    dys = [W.seeded_normal(tuple(o.shape), seed_dy + j).to(o.dtype) for j, o in enumerate(ys)]
    torch.autograd.backward(ys, dys)
    for j, o in enumerate(ys):
        out['%s/out%d' % (prefix, j)] = t2n(o)
    for j, x in enumerate(xs):
        if x.grad is not None:
            out['%s/din%d' % (prefix, j)] = t2n(x.grad)
    grads_summary([(k, p.grad) for k, p in net.named_parameters()], prefix + '/dparam', out, full_grads)
    for k, b in net.named_buffers():
        if 'running' in k:
            a = t2n(b).astype(np.float64)
            out['%s/buf/%s' % (prefix, k)] = np.array([a.sum(), np.abs(a).sum()])


def golden_nets(rn, outdir):
    """Per-network vectors: G (2 and 9 blocks), D, E (plain / noisy), IP."""
    out = {}
    torch.manual_seed(0)
    nl_in = rn.get_norm_layer('instance')
    nl_bn = rn.get_norm_layer('batch')
    for nb in (2, 9):
        g = rn.ResnetGenerator(3, 3, 1, 8, norm_layer=nl_in, n_blocks=nb)
        g.load_state_dict(W.fill_state_dict(g.state_dict(), 10 + nb))
        x = W.seeded_tensor((2, 3, 16, 16), 100 + nb)
        z = W.seeded_normal((2, 1, 1, 1), 200 + nb)
        run_net(g, [x, z], 300 + nb, out, 'G%d' % nb)
        # z given once for the whole batch (1,nz,1,1) is broadcast (probe_quirks)
        g.zero_grad()
        with torch.no_grad():
            out['G%d/out_zbroadcast' % nb] = t2n(g(x, z[:1]))
    d = rn.NLayerDiscriminator(3, 1, 8, n_layers=3, norm_layer=nl_bn, use_sigmoid=True)
    d.load_state_dict(W.fill_state_dict(d.state_dict(), 20))
    run_net(d, [W.seeded_tensor((3, 3, 32, 32), 101), W.seeded_normal((3, 1, 1, 1), 201)], 301, out, 'D')
    for noisy in (False, True):
        e = rn.SiameseFeature(rn.ResNetFeature(3, 'resnet18'), pooling='avg', cnn_dim=[32, 1], cnn_pad=1,
                              cnn_relu_slope=0.7, noisy=noisy, drop_layer=rn.get_dropout_layer(0.))
        e.load_state_dict(W.fill_state_dict(e.state_dict(), 30))
        run_net(e, [W.seeded_tensor((3, 3, 64, 64), 102)], 302, out, 'E_noisy%d' % int(noisy), full_grads=False)
    ip = rn.AlexNetFeature(input_nc=3, pooling='None')
    ip.load_state_dict(W.fill_state_dict(ip.state_dict(), 40))
    run_net(ip, [W.seeded_tensor((2, 3, 64, 64), 103)], 303, out, 'IP', full_grads=False)
    np.savez_compressed(os.path.join(outdir, 'nets.npz'), **out)
    print('nets.npz: %d arrays' % len(out))


STEP_VARIANTS = {
    'default': [],
    'noisy_a': ['--noisy', 'true', '--noisy_var_type', 'a'],
    'bayesian_e': ['--bayesian', 'true', '--bnn_dropout', '0.2', '--noisy_var_type', 'e', '--bnn_T', '3'],
    'bayesian_noisy_ae': ['--bayesian', 'true', '--noisy', 'true', '--bnn_dropout', '0.2', '--noisy_var_type', 'ae',
                          '--bnn_T', '3'],
    'use_real_A': ['--use_real_A'],
    'lambda_A_GAN': ['--lambda_A_GAN', '0.3', '--lambda_L1', '0.7'],
    'detach_fake_B': ['--detach_fake_B'],
    'no_ip_no_z': ['--lambda_IP', '0', '--lambda_z', '0'],
    # one label for the whole batch, drawn with np.random.choice; per-label input keys (models/wsgan_emb_model.py:199-207)
    'no_mixed_label_D': ['--no_mixed_label_D'],
    # InstanceNorm discriminator: its convolutions gain biases, the norm entries become buffers only (models/networks.py:740-743)
    'norm_D_instance': ['--norm_D', 'instance'],
}


def step_batch(name, it):
    """the seeded input dict of iteration `it` of a step variant (shared with the tests)"""
    A = W.seeded_tensor((4, 3, 32, 32), 500 + it)
    B = W.seeded_tensor((4, 3, 32, 32), 600 + it)
    if name == 'no_mixed_label_D':
        # what WSGANEmbDataset yields under --no_mixed_label_D: one pair set per label that has lines (0 and 2 here)
        A2 = W.seeded_tensor((4, 3, 32, 32), 520 + it)
        B2 = W.seeded_tensor((4, 3, 32, 32), 620 + it)
        return {'0_A': A, '0_B': B, '0_A_paths': ['a0'] * 4, '0_B_paths': ['b0'] * 4,
                '2_A': A2, '2_B': B2, '2_A_paths': ['a2'] * 4, '2_B_paths': ['b2'] * 4}
    label = torch.tensor([0, 2, 2, 0] if it == 0 else [2, 0, 1, 0], dtype=torch.int64)
    return {'A': A, 'B': B, 'label': label, 'A_paths': ['a'] * 4, 'B_paths': ['b'] * 4}


NP_SEED = 4242      # numpy's global generator is seeded with NP_SEED + it before set_input (np.random.choice of the label)


def golden_steps(rn, outdir, only=None):
    """Full optimize_parameters() x2 through the reference's own option parser and model class."""
    from options.train_options import TrainOptions
    from models import create_model
    tmp = tempfile.mkdtemp(prefix='pcgan_golden_')
    for name, extra in STEP_VARIANTS.items():
        if only and name not in only:
            continue
        noisy = 'true' in [a for i, a in enumerate(extra) if i > 0 and extra[i - 1] == '--noisy']
        drop = 0.2 if '--bnn_dropout' in extra else 0.0
        # fabricated "pretrained" checkpoints with deterministic weights
        e = rn.SiameseFeature(rn.ResNetFeature(3, 'resnet18', dropout=drop), pooling='avg', cnn_dim=[32, 1], cnn_pad=1,
                              cnn_relu_slope=0.7, noisy=noisy, drop_layer=rn.get_dropout_layer(drop))
        e_path = os.path.join(tmp, 'E_%s.pth' % name)
        torch.save(W.fill_state_dict(e.state_dict(), 30), e_path)
        ip = rn.AlexNetFeature(input_nc=3, pooling='None')
        ip_path = os.path.join(tmp, 'IP.pth')
        torch.save(W.fill_state_dict(ip.state_dict(), 40), ip_path)
        sys.argv = ['train.py', '--dataroot', tmp, '--model', 'wsgan_emb', '--name', 'g_' + name,
                    '--checkpoints_dir', tmp, '--gpu_ids', '-1', '--which_model_netG', 'resnet_9blocks',
                    '--which_model_netD', 'n_layers', '--n_layers_D', '3', '--ngf', '8', '--ndf', '8',
                    '--fineSize', '32', '--loadSize', '32', '--fineSize_E', '64', '--fineSize_IP', '64',
                    '--batchSize', '4', '--pretrained_model_path_E', e_path, '--pretrained_model_path_IP', ip_path,
                    '--display_id', '-1', '--embedding_bins', '[-1.0, 0.0, 1.5]', '--embedding_mean', '0.1',
                    '--embedding_std', '0.8'] + extra
        opt = TrainOptions().parse()
        model = create_model(opt)
        model.setup(opt)
        model.netG.load_state_dict(W.damp_generator_head(W.fill_state_dict(model.netG.state_dict(), 19)))
        model.netD.load_state_dict(W.fill_state_dict(model.netD.state_dict(), 20))
        out = {}
        for it in range(2):
            torch.manual_seed(1234 + it)
            np.random.seed(NP_SEED + it)
            # capture grads at the moment of each optimizer step
            grabbed, origs = {}, {}
            for tag, optim, net in (('G', model.optimizer_G, model.netG), ('D', model.optimizer_D, model.netD)):
                orig = origs[tag] = optim.step

                def stepper(orig=orig, tag=tag, net=net):
                    grabbed[tag] = [(k, None if p.grad is None else p.grad.detach().clone())
                                    for k, p in net.named_parameters()]
                    return orig()
                optim.step = stepper
            model.set_input(step_batch(name, it))
            model.optimize_parameters()
            model.optimizer_G.step, model.optimizer_D.step = origs['G'], origs['D']
            p = 'it%d' % it
            out[p + '/label_AB'] = np.array([int(v) for v in model.label_AB], dtype=np.int64)
            losses = model.get_current_losses()
            out[p + '/losses'] = np.array([losses[k] for k in model.loss_names], dtype=np.float64)
            for k in ('fake_B', 'rec_A', 'embedding_A', 'embedding_B', 'y_A', 'y_B'):
                out['%s/%s' % (p, k)] = t2n(getattr(model, k))
            if hasattr(model, 'resample_B') and opt.noisy_var_type:
                out[p + '/resample_A'] = t2n(model.resample_A)
                out[p + '/resample_B'] = t2n(model.resample_B)
            full = (name == 'default')
            grads_summary(grabbed['G'], p + '/gradG', out, full)
            grads_summary(grabbed['D'], p + '/gradD', out, full)
            for tag, net in (('G', model.netG), ('D', model.netD)):
                for k, v in net.state_dict().items():
                    a = t2n(v).astype(np.float64)
                    out['%s/after%s/%s' % (p, tag, k)] = np.array([a.sum(), np.abs(a).sum()])
            for k, v in model.netE.state_dict().items():
                if 'running' in k or 'num_batches' in k:
                    a = t2n(v).astype(np.float64)
                    out['%s/afterE/%s' % (p, k)] = np.array([a.sum(), np.abs(a).sum()])
        out['loss_names'] = np.array(model.loss_names)
        np.savez_compressed(os.path.join(outdir, 'step_%s.npz' % name), **out)
        print('step_%s.npz: %d arrays, losses it0 %s' % (name, len(out), out['it0/losses']))
        if name == 'default':
            # LR schedule through the reference's scheduler (models/networks.py:57-62)
            lrs = []
            for _ in range(opt.niter + opt.niter_decay + 1):
                lrs.append(model.optimizers[0].param_groups[0]['lr'])
                for s in model.schedulers:
                    s.step()
            np.savez(os.path.join(outdir, 'lr_schedule.npz'), lr=np.array(lrs), niter=opt.niter,
                     niter_decay=opt.niter_decay, epoch_count=opt.epoch_count, base_lr=opt.lr)


def golden_visuals(rn, outdir):
    """get_current_visuals() of wsgan_emb (models/wsgan_emb_model.py:486-497): after set_input + forward() the model runs G on
    real_A[0:1] once per fixed rating bin IN TRAIN MODE, which moves the InstanceNorm running statistics; recorded: the
    attr_<i> images and G's buffers before / after the call."""
    from options.train_options import TrainOptions
    from models import create_model
    tmp = tempfile.mkdtemp(prefix='pcgan_golden_vis_')
    e = rn.SiameseFeature(rn.ResNetFeature(3, 'resnet18', dropout=0.0), pooling='avg', cnn_dim=[32, 1], cnn_pad=1,
                          cnn_relu_slope=0.7, noisy=False, drop_layer=rn.get_dropout_layer(0.0))
    e_path = os.path.join(tmp, 'E.pth')
    torch.save(W.fill_state_dict(e.state_dict(), 30), e_path)
    ip = rn.AlexNetFeature(input_nc=3, pooling='None')
    ip_path = os.path.join(tmp, 'IP.pth')
    torch.save(W.fill_state_dict(ip.state_dict(), 40), ip_path)
    sys.argv = ['train.py', '--dataroot', tmp, '--model', 'wsgan_emb', '--name', 'g_vis', '--checkpoints_dir', tmp, '--gpu_ids', '-1',
                '--which_model_netG', 'resnet_9blocks', '--which_model_netD', 'n_layers', '--n_layers_D', '3', '--ngf', '8',
                '--ndf', '8', '--fineSize', '32', '--loadSize', '32', '--fineSize_E', '64', '--fineSize_IP', '64', '--batchSize', '4',
                '--pretrained_model_path_E', e_path, '--pretrained_model_path_IP', ip_path, '--display_id', '-1',
                '--embedding_bins', '[-1.0, 0.0, 1.5]', '--embedding_mean', '0.1', '--embedding_std', '0.8']
    opt = TrainOptions().parse()
    model = create_model(opt)
    model.setup(opt)
    model.netG.load_state_dict(W.damp_generator_head(W.fill_state_dict(model.netG.state_dict(), 19)))
    model.netD.load_state_dict(W.fill_state_dict(model.netD.state_dict(), 20))
    # forward() only, no optimizer step in front: Adam moves the parameters whose true gradient is 0 (biases, the rating
    # channel's filter slice) by +-lr according to the SIGN of fp32 noise, and exactly those parameters shift the plane
    # means that the running statistics record -- after a step two correct implementations differ there by construction
    torch.manual_seed(1234)
    model.set_input(step_batch('default', 0))
    model.forward()
    out = {}
    for k, v in model.netG.state_dict().items():
        if 'running' in k:
            out['before/' + k] = t2n(v).copy()
    vis = model.get_current_visuals()
    out['names'] = np.array(list(vis.keys()))
    for k, v in vis.items():
        out['vis/' + k] = t2n(v)
    for k, v in model.netG.state_dict().items():
        if 'running' in k:
            out['after/' + k] = t2n(v).copy()
    out['requires_grad_after'] = np.array([p.requires_grad for p in model.netG.parameters()])
    np.savez_compressed(os.path.join(outdir, 'visuals.npz'), **out)
    print('visuals.npz: %d arrays, visuals %s' % (len(out), list(vis.keys())))


SAMPLER_VARIANTS = ('default', 'noisy_a', 'bayesian_e')


def golden_samplers(rn, outdir):
    """sample_from_prior() / sample_from_label(l) of wsgan_emb (models/wsgan_emb_model.py:279-298), the reference's own
    methods on the tiny config: the image generated from the encoder's rating of real_B, and one image per fixed rating bin.
    Both run G (and E) in TRAIN mode as the reference does (train.py never calls eval()), so the InstanceNorm / BatchNorm
    running statistics move; recorded: the images, embedding_B and the G / E buffer checksums afterwards.  The noisy / bayesian
    branches read `self.real_B_E`, an attribute only forward() sets (:281-288), so forward() runs first in every variant."""
    from options.train_options import TrainOptions
    from models import create_model
    tmp = tempfile.mkdtemp(prefix='pcgan_golden_samp_')
    out = {}
    for name in SAMPLER_VARIANTS:
        extra = STEP_VARIANTS[name]
        noisy = '--noisy' in extra and extra[extra.index('--noisy') + 1] == 'true'
        drop = 0.2 if '--bnn_dropout' in extra else 0.0
        e = rn.SiameseFeature(rn.ResNetFeature(3, 'resnet18', dropout=drop), pooling='avg', cnn_dim=[32, 1], cnn_pad=1,
                              cnn_relu_slope=0.7, noisy=noisy, drop_layer=rn.get_dropout_layer(drop))
        e_path = os.path.join(tmp, 'E_%s.pth' % name)
        torch.save(W.fill_state_dict(e.state_dict(), 30), e_path)
        ip = rn.AlexNetFeature(input_nc=3, pooling='None')
        ip_path = os.path.join(tmp, 'IP.pth')
        torch.save(W.fill_state_dict(ip.state_dict(), 40), ip_path)
        sys.argv = ['train.py', '--dataroot', tmp, '--model', 'wsgan_emb', '--name', 'g_samp_' + name, '--checkpoints_dir', tmp,
                    '--gpu_ids', '-1', '--which_model_netG', 'resnet_9blocks', '--which_model_netD', 'n_layers', '--n_layers_D', '3',
                    '--ngf', '8', '--ndf', '8', '--fineSize', '32', '--loadSize', '32', '--fineSize_E', '64', '--fineSize_IP', '64',
                    '--batchSize', '4', '--pretrained_model_path_E', e_path, '--pretrained_model_path_IP', ip_path,
                    '--display_id', '-1', '--embedding_bins', '[-1.0, 0.0, 1.5]', '--embedding_mean', '0.1',
                    '--embedding_std', '0.8'] + extra
        opt = TrainOptions().parse()
        model = create_model(opt)
        model.setup(opt)
        model.netG.load_state_dict(W.damp_generator_head(W.fill_state_dict(model.netG.state_dict(), 19)))
        model.netD.load_state_dict(W.fill_state_dict(model.netD.state_dict(), 20))
        torch.manual_seed(1234)
        model.set_input(step_batch(name, 0))
        model.forward()
        torch.manual_seed(777)                   # the MC-dropout passes of the bayesian sampler draw from here
        with torch.no_grad():
            out[name + '/prior'] = t2n(model.sample_from_prior())
            out[name + '/embedding_B'] = t2n(model.embedding_B)
            for label in range(3):
                out['%s/label%d' % (name, label)] = t2n(model.sample_from_label(label))
        for tag, net in (('G', model.netG), ('E', model.netE)):
            for k, v in net.state_dict().items():
                if 'running' in k or 'num_batches' in k:
                    a = t2n(v).astype(np.float64)
                    out['%s/after%s/%s' % (name, tag, k)] = np.array([a.sum(), np.abs(a).sum()])
    np.savez_compressed(os.path.join(outdir, 'samplers.npz'), **out)
    print('samplers.npz: %d arrays' % len(out))


def golden_cycle_step(rn, outdir):
    """wsgan_cycle: optimize_parameters() x2 through the reference's own parser and model class:
    unconditional D, trained ResNet-18 encoder (max pooling, cnn_dim [64, 1]), resnet generator."""
    from options.train_options import TrainOptions
    from models import create_model
    tmp = tempfile.mkdtemp(prefix='pcgan_golden_cycle_')
    base = rn.ResNetFeature(3, 'resnet18')
    base_path = os.path.join(tmp, 'resnet18_base.pth')
    torch.save(W.fill_state_dict(base.model.state_dict(), 31), base_path)        # plain ResNet state_dict (load_base)
    ip = rn.AlexNetFeature(input_nc=3, pooling='None')
    ip_path = os.path.join(tmp, 'IP.pth')
    torch.save(W.fill_state_dict(ip.state_dict(), 40), ip_path)
    sys.argv = ['train.py', '--dataroot', tmp, '--model', 'wsgan_cycle', '--name', 'g_cycle',
                '--checkpoints_dir', tmp, '--gpu_ids', '-1', '--which_model_netG', 'resnet_9blocks',
                '--which_model_netD', 'n_layers', '--n_layers_D', '3', '--ngf', '8', '--ndf', '8',
                '--fineSize', '32', '--loadSize', '32', '--fineSize_E', '64', '--fineSize_IP', '64',
                '--batchSize', '4', '--pretrained_model_path_E', base_path, '--pretrained_model_path_IP', ip_path,
                '--display_id', '-1', '--attr_bins', '[10, 30, 50]', '--attr_mean', '35.0', '--attr_std', '20.0']
    opt = TrainOptions().parse()
    model = create_model(opt)
    model.setup(opt)
    model.netG.load_state_dict(W.damp_generator_head(W.fill_state_dict(model.netG.state_dict(), 19)))
    model.netD.load_state_dict(W.fill_state_dict(model.netD.state_dict(), 20))
    esd = model.netE.state_dict()                       # the head (cnn.*) is not covered by load_base: make it deterministic
    filled = W.fill_state_dict(esd, 33)
    for k in esd:
        if k.startswith('cnn'):
            esd[k] = filled[k]
    model.netE.load_state_dict(esd)
    out = {}
    for it in range(2):
        A = W.seeded_tensor((4, 3, 32, 32), 700 + it)
        attr = (W.seeded_tensor((4, 1, 1, 1), 800 + it) + 1.0) * 30.0          # attributes in [0, 60)
        torch.manual_seed(4321 + it)
        grabbed, origs = {}, {}
        for tag, optim, net in (('G', model.optimizer_G, model.netG), ('D', model.optimizer_D, model.netD),
                                ('E', model.optimizer_E, model.netE)):
            orig = origs[tag] = optim.step

            def stepper(orig=orig, tag=tag, net=net):
                grabbed[tag] = [(k, None if p.grad is None else p.grad.detach().clone()) for k, p in net.named_parameters()]
                return orig()
            optim.step = stepper
        model.set_input({'A': A, 'B_attr': attr, 'A_paths': ['a'] * 4, 'B_paths': ['b'] * 4})
        model.optimize_parameters()
        model.optimizer_G.step, model.optimizer_D.step, model.optimizer_E.step = origs['G'], origs['D'], origs['E']
        p = 'it%d' % it
        losses = model.get_current_losses()
        out[p + '/losses'] = np.array([losses[k] for k in model.loss_names], dtype=np.float64)
        for k in ('fake_x', 'rec_x', 'fake_y', 'rec_y', 'real_y'):
            out['%s/%s' % (p, k)] = t2n(getattr(model, k))
        grads_summary(grabbed['G'], p + '/gradG', out, False)
        grads_summary(grabbed['D'], p + '/gradD', out, False)
        grads_summary(grabbed['E'], p + '/gradE', out, False)
        for tag, net in (('G', model.netG), ('D', model.netD), ('E', model.netE)):
            for k, v in net.state_dict().items():
                a = t2n(v).astype(np.float64)
                out['%s/after%s/%s' % (p, tag, k)] = np.array([a.sum(), np.abs(a).sum()])
    out['loss_names'] = np.array(model.loss_names)
    np.savez_compressed(os.path.join(outdir, 'cycle_step.npz'), **out)
    print('cycle_step.npz: %d arrays, losses it0 %s' % (len(out), out['it0/losses']))


def golden_siamese(rn, outdir):
    """Elo-encoder trainer: three Adam steps of the reference's SiameseNetwork (ResNet-18 trunk, head
    [32, 1], average pooling) on seeded pairs.  siamese.py itself cannot be imported (torchvision / cv2 / tqdm at module
    top) and BinaryNLLLoss.__init__ calls .cuda(): the loss expression (networks.py:479-481) is applied here in place."""
    base = rn.ResNetFeature(3, 'resnet18')
    net = rn.SiameseNetwork(base, pooling='avg', cnn_dim=[32, 1], cnn_pad=1, cnn_relu_slope=0.7, fc_dim=[],
                            drop_layer=rn.get_dropout_layer(0.0))
    net.load_state_dict(W.fill_state_dict(net.state_dict(), 61))
    params = list(net.base.parameters()) + list(net.cnn.parameters())
    optimizer = torch.optim.Adam(params, lr=2e-4)
    lut = torch.tensor([0.0, 0.5, 1.0])
    out = {}
    for it in range(3):
        img0 = W.seeded_tensor((4, 3, 64, 64), 900 + it)
        img1 = W.seeded_tensor((4, 3, 64, 64), 950 + it)
        label = torch.tensor([[0, 2, 1, 2], [2, 2, 0, 1], [1, 0, 0, 2]][it])
        optimizer.zero_grad()
        f1, f2, score = net(img0, img1)
        prob = torch.sigmoid(score)
        target = lut[label].reshape(prob.size(0), 1, 1, 1).expand(prob.size(0), 1, prob.size(2), prob.size(3))
        loss = -(target * torch.log(prob + 1e-20) + (1 - target) * torch.log(1 - prob + 1e-20)).mean()
        loss.backward()
        p = 'it%d' % it
        out[p + '/loss'] = np.array(float(loss))
        out[p + '/f1'], out[p + '/f2'], out[p + '/prob'] = t2n(f1), t2n(f2), t2n(prob)
        grads_summary([(k, q.grad) for k, q in net.named_parameters()], p + '/grad', out, False)
        optimizer.step()
        for k, v in net.state_dict().items():
            a = t2n(v).astype(np.float64)
            out['%s/after/%s' % (p, k)] = np.array([a.sum(), np.abs(a).sum()])
    np.savez_compressed(os.path.join(outdir, 'siamese_step.npz'), **out)
    print('siamese_step.npz: %d arrays, losses %s' % (len(out), [float(out['it%d/loss' % i]) for i in range(3)]))


# Elo-trainer variants (reference siamese.py:577-669): name -> (noisy, rsample, lb_or_mc, bnn_dropout, T_train, M)
SIAMESE_VARIANTS = {
    'bayesian': (False, True, 'lb', 0.2, 2, 1),
    'noisy_std': (True, False, 'lb', 0.0, 1, 1),
    'noisy_mc': (True, True, 'mc', 0.0, 1, 3),
    'noisy_lb': (True, True, 'lb', 0.0, 1, 2),
    'bayesian_noisy_lb': (True, True, 'lb', 0.2, 2, 2),
    'bayesian_noisy_std': (True, False, 'lb', 0.2, 2, 1),
}
SIAMESE_LR_SIGMA = 1e-4        # (the reference's default 2e-7 would leave cnn_logvar unmoved in two steps)


def golden_siamese_variants(rn, outdir):
    """The reparameterised / MC-dropout variants of the Elo trainer's iteration (reference siamese.py:577-669: --noisy with
    --rsample mc / lb or the score / score_std form, --bayesian with T_train passes): two iterations each of the reference's
    SiameseNetwork with its own `reparameterize` (util/util.py:130-133), the branch bodies applied as in golden_siamese (siamese.py
    cannot be imported); a second Adam on `cnn_logvar` (siamese.py:552-553, 667-668).  torch.manual_seed(1000 + it) before every
    iteration: the oracle draws the same dropout masks and eps under the same seed."""
    import util.util as ref_util
    lut = torch.tensor([0.0, 0.5, 1.0])

    def criterion(prob, label):
        target = lut[label].reshape(prob.size(0), 1, 1, 1).expand(prob.size(0), 1, prob.size(2), prob.size(3))
        return -(target * torch.log(prob + 1e-20) + (1 - target) * torch.log(1 - prob + 1e-20)).mean()

    out = {}
    for name, (noisy, rsample, lb_or_mc, p_drop, T, M) in SIAMESE_VARIANTS.items():
        base = rn.ResNetFeature(3, 'resnet18', dropout=p_drop)
        net = rn.SiameseNetwork(base, pooling='avg', cnn_dim=[32, 1], cnn_pad=1, cnn_relu_slope=0.7, fc_dim=[], noisy=noisy,
                                drop_layer=rn.get_dropout_layer(p_drop), rsample=rsample)
        net.load_state_dict(W.fill_state_dict(net.state_dict(), 61))
        params = list(net.base.parameters()) + list(net.cnn.parameters())
        optimizer = torch.optim.Adam(params, lr=2e-4)
        optimizer_sigma = torch.optim.Adam(net.cnn_logvar.parameters(), lr=SIAMESE_LR_SIGMA) if noisy else None
        bayesian = p_drop > 0
        for it in range(2):
            img0 = W.seeded_tensor((4, 3, 64, 64), 900 + it)
            img1 = W.seeded_tensor((4, 3, 64, 64), 950 + it)
            label = torch.tensor([[0, 2, 1, 2], [2, 2, 0, 1]][it])
            torch.manual_seed(1000 + it)
            optimizer.zero_grad()
            if noisy:
                optimizer_sigma.zero_grad()
            TT = T if bayesian else 1
            loss = 0.0
            for t in range(TT):
                if noisy and rsample:
                    y1, y2, logvar1, logvar2 = net(img0, img1)
                    if lb_or_mc == 'mc':
                        prob_ = 0.0
                        for m in range(M):
                            score = ref_util.reparameterize(y1, logvar1) - ref_util.reparameterize(y2, logvar2)
                            prob_ = prob_ + 1. / M * torch.sigmoid(score)
                        loss = loss + 1. / TT * criterion(prob_, label)
                    else:
                        for m in range(M):
                            score = ref_util.reparameterize(y1, logvar1) - ref_util.reparameterize(y2, logvar2)
                            prob_ = torch.sigmoid(score)
                            loss = loss + 1. / (TT * M) * criterion(prob_, label)
                elif noisy:
                    y1, y2, score, score_std = net(img0, img1)
                    prob_ = torch.sigmoid(score / (score_std + 1e-20))
                    loss = loss + 1. / TT * criterion(prob_, label)
                else:
                    y1, y2, score = net(img0, img1)
                    prob_ = torch.sigmoid(score)
                    loss = loss + 1. / TT * criterion(prob_, label)
            loss.backward()
            q = '%s/it%d' % (name, it)
            out[q + '/loss'] = np.array(float(loss))
            out[q + '/f1'], out[q + '/prob'] = t2n(y1), t2n(prob_)        # of the last pass
            grads_summary([(k, v.grad) for k, v in net.named_parameters()], q + '/grad', out, False)
            optimizer.step()
            if noisy:
                optimizer_sigma.step()
            for k, v in net.state_dict().items():
                a = t2n(v).astype(np.float64)
                out['%s/after/%s' % (q, k)] = np.array([a.sum(), np.abs(a).sum()])
        print('siamese variant %-20s losses %s' % (name, [float(out['%s/it%d/loss' % (name, i)]) for i in range(2)]))
    np.savez_compressed(os.path.join(outdir, 'siamese_variants.npz'), **out)


def golden_ints(outdir):
    """Integer-exact helpers."""
    from util import util as ref_util
    bins = [1, 21, 41, 61, 81, float('inf')]
    attrs = [-5, 1, 20.999, 21, 80, 81, 1e9, float('nan'), 40.9999, 61]
    labels = [ref_util.get_attr_label(a, bins) for a in attrs]
    short = ref_util.get_attr_label(3.0, [5])
    strs = ['[]', '[1, 2, 3]', ' [0.5, -1.25] ', '[[1, 2], [3, 4]]']
    parsed = [repr(ref_util.str2list(s)) for s in strs]
    bools = ['yes', 'True', 't', 'Y', '1', 'no', 'FALSE', 'f', 'n', '0']
    np.savez(os.path.join(outdir, 'ints.npz'), bins=np.array(bins), attrs=np.array(attrs),
             labels=np.array(labels, dtype=np.int64), short_is_none=np.array(short is None),
             strs=np.array(strs), parsed=np.array(parsed), bools=np.array(bools),
             bool_vals=np.array([ref_util.str2bool(b) for b in bools]))
    print('ints.npz labels', labels)


def golden_fid(outdir):
    """Frechet distance (compute_fid_score.py:126-205) on seeded Gaussians' statistics, incl. a rank-deficient pair that
    takes the eps branch.  The reference module imports `scipy.misc.imread` (removed from SciPy) and
    `torchvision.models` at its top without touching them in these two functions: ordinary missing-module stubs."""
    import scipy
    if 'scipy.misc' not in sys.modules:
        misc = types.ModuleType('scipy.misc')
        misc.imread = None
        sys.modules['scipy.misc'] = misc
        scipy.misc = misc
    if 'torchvision.models' not in sys.modules:
        sys.modules['torchvision.models'] = types.ModuleType('torchvision.models')
        sys.modules['torchvision'].models = sys.modules['torchvision.models']
    import compute_fid_score as ref_fid
    rng = np.random.default_rng(2024)
    out = {}
    cases = [(8, 40), (16, 200), (64, 500), (32, 20)]          # (dims, samples); the last one is rank deficient
    for i, (d, n) in enumerate(cases):
        a = rng.normal(size=(n, d)) @ rng.normal(size=(d, d)) + rng.normal(size=d)
        b = rng.normal(size=(n, d)) * rng.uniform(0.5, 2.0, size=d) + rng.normal(size=d) * 0.3
        m1, s1 = a.mean(axis=0), np.cov(a, rowvar=False)
        m2, s2 = b.mean(axis=0), np.cov(b, rowvar=False)
        out['act1_%d' % i], out['act2_%d' % i] = a, b
        out['fid_%d' % i] = np.float64(ref_fid.calculate_frechet_distance(m1, s1, m2, s2))
        out['fid_self_%d' % i] = np.float64(ref_fid.calculate_frechet_distance(m1, s1, m1, s1))
    np.savez(os.path.join(outdir, 'fid.npz'), **out)
    print('fid.npz', [float(out['fid_%d' % i]) for i in range(len(cases))], [float(out['fid_self_%d' % i]) for i in range(len(cases))])


LR_E_ARGS = ['--lr_E', '0.0001']


def golden_step_lr_E(rn, outdir):
    """`--lr_E > 0`: the reference's OWN update_G_and_E / backward_GE / backward_G_alone (models/wsgan_emb_model.py:331-369,
    439-449, 463-476), two iterations.  That branch steps G and E and then back-propagates through the retained graph, which
    torch >= 1.5 refuses when a stock optimizer wrote the parameters; here `torch.optim.Adam` is replaced -- while the
    reference builds its optimizers, and only there -- by oracle.step_ref.AdamThroughData (same update, written through p.data as
    the optimizers of the reference's era did).  The optimizer's semantics are DEFINED by that class; everything else in the run
    is the reference's code."""
    from options.train_options import TrainOptions
    from models import create_model
    from oracle import step_ref as S
    tmp = tempfile.mkdtemp(prefix='pcgan_golden_lrE_')
    e = rn.SiameseFeature(rn.ResNetFeature(3, 'resnet18', dropout=0.0), pooling='avg', cnn_dim=[32, 1], cnn_pad=1,
                          cnn_relu_slope=0.7, noisy=False, drop_layer=rn.get_dropout_layer(0.0))
    e_path = os.path.join(tmp, 'E.pth')
    torch.save(W.fill_state_dict(e.state_dict(), 30), e_path)
    ip = rn.AlexNetFeature(input_nc=3, pooling='None')
    ip_path = os.path.join(tmp, 'IP.pth')
    torch.save(W.fill_state_dict(ip.state_dict(), 40), ip_path)
    sys.argv = ['train.py', '--dataroot', tmp, '--model', 'wsgan_emb', '--name', 'g_lrE', '--checkpoints_dir', tmp, '--gpu_ids', '-1',
                '--which_model_netG', 'resnet_9blocks', '--which_model_netD', 'n_layers', '--n_layers_D', '3', '--ngf', '8',
                '--ndf', '8', '--fineSize', '32', '--loadSize', '32', '--fineSize_E', '64', '--fineSize_IP', '64', '--batchSize', '4',
                '--pretrained_model_path_E', e_path, '--pretrained_model_path_IP', ip_path, '--display_id', '-1',
                '--embedding_bins', '[-1.0, 0.0, 1.5]', '--embedding_mean', '0.1', '--embedding_std', '0.8'] + LR_E_ARGS
    opt = TrainOptions().parse()
    stock = torch.optim.Adam
    torch.optim.Adam = S.AdamThroughData
    try:
        model = create_model(opt)
        model.setup(opt)
    finally:
        torch.optim.Adam = stock
    assert isinstance(model.optimizer_E, S.AdamThroughData) and isinstance(model.optimizer_G, S.AdamThroughData)
    model.netG.load_state_dict(W.damp_generator_head(W.fill_state_dict(model.netG.state_dict(), 19)))
    model.netD.load_state_dict(W.fill_state_dict(model.netD.state_dict(), 20))
    out = {}
    for it in range(2):
        torch.manual_seed(1234 + it)
        np.random.seed(NP_SEED + it)
        grabbed, origs = [], {}
        for tag, optim, net in (('G', model.optimizer_G, model.netG), ('E', model.optimizer_E, model.netE), ('D', model.optimizer_D, model.netD)):
            orig = origs[tag] = optim.step

            def stepper(orig=orig, tag=tag, net=net):
                grabbed.append((tag, [(k, None if p.grad is None else p.grad.detach().clone()) for k, p in net.named_parameters()]))
                return orig()
            optim.step = stepper
        model.set_input(step_batch('default', it))
        model.optimize_parameters()
        model.optimizer_G.step, model.optimizer_E.step, model.optimizer_D.step = origs['G'], origs['E'], origs['D']
        assert [t for t, _ in grabbed] == ['G', 'E', 'G', 'D'], [t for t, _ in grabbed]
        p = 'it%d' % it
        losses = model.get_current_losses()
        out[p + '/losses'] = np.array([losses[k] for k in model.loss_names], dtype=np.float64)
        for k in ('fake_B', 'rec_A', 'embedding_A', 'embedding_B', 'y_A', 'y_B'):
            out['%s/%s' % (p, k)] = t2n(getattr(model, k))
        for (tag, grads), name in zip(grabbed, ('gradG', 'gradE', 'gradG_alone', 'gradD')):
            grads_summary(grads, '%s/%s' % (p, name), out, full=(name == 'gradG_alone'))
        for tag, net in (('G', model.netG), ('D', model.netD), ('E', model.netE)):
            for k, v in net.state_dict().items():
                a = t2n(v).astype(np.float64)
                out['%s/after%s/%s' % (p, tag, k)] = np.array([a.sum(), np.abs(a).sum()])
    out['loss_names'] = np.array(model.loss_names)
    np.savez_compressed(os.path.join(outdir, 'step_lr_E.npz'), **out)
    print('step_lr_E.npz: %d arrays, losses it0 %s it1 %s' % (len(out), out['it0/losses'], out['it1/losses']))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    ap.add_argument('--only', default='')
    ap.add_argument('--variants', default='', help='comma-separated step variants (with --only steps); default: all')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    rn = import_reference()
    torch.set_num_threads(4)
    if a.only in ('', 'nets'):
        golden_nets(rn, a.out)
    if a.only in ('', 'ints'):
        golden_ints(a.out)
    if a.only in ('', 'steps'):
        golden_steps(rn, a.out, [v for v in a.variants.split(',') if v] or None)
    if a.only in ('', 'visuals'):
        golden_visuals(rn, a.out)
    if a.only in ('', 'samplers'):
        golden_samplers(rn, a.out)
    if a.only in ('', 'cycle'):
        golden_cycle_step(rn, a.out)
    if a.only in ('', 'siamese'):
        golden_siamese(rn, a.out)
    if a.only in ('', 'siamese_variants'):
        golden_siamese_variants(rn, a.out)
    if a.only in ('', 'fid'):
        golden_fid(a.out)
    if a.only in ('', 'lr_E'):
        golden_step_lr_E(rn, a.out)
