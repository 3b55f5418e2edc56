"""Network factories of the PC-GAN hot path on the HIP layer set.

Same plugin surface as the reference's models/networks.py for the part of the zoo that the
wsgan_emb / wsgan_cycle configurations reach:
define_G / define_D / define_E / define_IP, get_norm_layer, init_weights, init_net,
get_scheduler, GANLoss, and the module classes with the reference's nn.Sequential index
layout, so state_dict keys are identical (`model.10.conv_block.1.weight`, `model.2.running_mean`,
...).  Architectures outside that scope raise NotImplementedError naming themselves.
"""
import functools

import numpy as np
import torch
import torch.nn as tnn
from torch.nn import init
from torch.optim import lr_scheduler

from ..hip import functional as HF
from ..hip import nn as hnn
from ..hip.nn import IdentityMapping, run_sequential  # noqa: F401  (IdentityMapping is part of the surface)

MAGIC_EPS = 1e-20


# ------------------------------------------------------------------------- helpers
def get_norm_layer(norm_type='instance'):
    """reference models/networks.py:22-34"""
    if norm_type == 'batch':
        return functools.partial(hnn.BatchNorm2d, affine=True)
    if norm_type == 'instance':
        return functools.partial(hnn.InstanceNorm2d, affine=False, track_running_stats=True)
    if norm_type == 'none':
        return IdentityMapping
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def _is_instance_norm(norm_layer):
    f = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    return f is hnn.InstanceNorm2d or f is tnn.InstanceNorm2d


def get_dropout_layer(dropout=0.):
    """reference models/networks.py:37-42"""
    return functools.partial(hnn.Dropout2d, p=dropout) if dropout > 0 else IdentityMapping


def get_scheduler(optimizer, opt):
    """reference models/networks.py:57-69"""
    if opt.lr_policy == 'lambda':
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + 1 + opt.epoch_count - opt.niter) / float(opt.niter_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if opt.lr_policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == 'plateau':
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.2, threshold=0.01, patience=5)
    # the reference RETURNS (does not raise) this object; keep the observable behaviour
    return NotImplementedError('learning rate policy [%s] is not implemented', opt.lr_policy)


_WEIGHT_FILLERS = {
    # --init_type -> in-place filler of a Conv / Linear weight (the draws consume torch's global generator exactly as the
    # reference's torch.nn.init calls do, so seeded initialisations coincide)
    'normal': lambda w, gain: init.normal_(w, 0.0, gain),
    'xavier': lambda w, gain: init.xavier_normal_(w, gain=gain),
    'kaiming': lambda w, gain: init.kaiming_normal_(w, a=0, mode='fan_in'),
    'orthogonal': lambda w, gain: init.orthogonal_(w, gain=gain),
}


def init_weights(net, init_type='normal', gain=0.02):
    """Initialisation rule of the reference (models/networks.py:72-93), by module class NAME as there: anything whose
    class name contains 'Conv' or 'Linear' and owns a weight gets the `init_type` filler and a zero bias; anything named
    *BatchNorm2d* gets weight ~ N(1, gain), bias 0; an unknown init_type raises when the first such layer is met."""
    fill = _WEIGHT_FILLERS.get(init_type)

    def visit(module):
        name = type(module).__name__
        if ('Conv' in name or 'Linear' in name) and hasattr(module, 'weight'):
            if fill is None:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            fill(module.weight.data, gain)
            if getattr(module, 'bias', None) is not None:
                module.bias.data.zero_()
        elif 'BatchNorm2d' in name:
            init.normal_(module.weight.data, 1.0, gain)
            module.bias.data.zero_()

    print('initialize network with %s' % init_type)
    net.apply(visit)
    from ..hip import ops            # .data writes do not bump tensor versions: drop packed-weight copies
    ops.invalidate_packed_weights()


def init_net(net, init_type='normal', gpu_ids=[]):
    """reference models/networks.py:96-102.  One process drives one GPU here: the net moves to
    gpu_ids[0]; data parallelism is the per-rank replica + RCCL gradient all-reduce of
    pcgan_amd.hip.parallel, not nn.DataParallel."""
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(torch.device('cuda', gpu_ids[0]))
    init_weights(net, init_type)
    return net


# ------------------------------------------------------------------------- factories
def define_G(input_nc, output_nc, nz, ngf, which_model_netG='unet_128', norm='batch', nl='relu',
             dropout=0, init_type='xavier', gpu_ids=[], upsample='bilinear', size=512, embed_size=256, n_layers_G=7):
    """reference models/networks.py:106-143.  `resnet_<n>blocks` for any n (the reference names
    only 6 and 9; BASELINE config 1 needs 2)."""
    norm_layer = get_norm_layer(norm_type=norm)
    if which_model_netG.startswith('resnet_') and which_model_netG.endswith('blocks'):
        try:
            n_blocks = int(which_model_netG[len('resnet_'):-len('blocks')])
        except ValueError:
            raise NotImplementedError('Generator model name [%s] is not recognized' % which_model_netG)
        netG = ResnetGenerator(input_nc, output_nc, nz, ngf, norm_layer=norm_layer, dropout=dropout, n_blocks=n_blocks)
    elif which_model_netG in ('unet_128', 'unet_256', 'unet_128_input', 'unet_128_all', 'unet_256_input',
                              'unet_256_all', 'gan_stability', 'mnist_fc', 'unet', 'unet_all'):
        raise NotImplementedError('Generator [%s] is outside the MI355X hot path (SURVEY.md section 8); '
                                  'use resnet_<n>blocks' % which_model_netG)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % which_model_netG)
    return init_net(netG, init_type, gpu_ids)


def define_D(input_nc, nz, ndf, which_model_netD, n_layers_D=3, norm='batch', use_sigmoid=False, init_type='normal',
             num_Ds=1, gpu_ids=[], use_projection=True, size=512, embed_size=256, num_classes=1):
    """reference models/networks.py:147-175"""
    norm_layer = get_norm_layer(norm_type=norm)
    if which_model_netD == 'basic':
        netD = NLayerDiscriminator(input_nc, nz, ndf, n_layers=3, norm_layer=norm_layer, use_sigmoid=use_sigmoid)
    elif which_model_netD == 'n_layers':
        netD = NLayerDiscriminator(input_nc, nz, ndf, n_layers_D, norm_layer=norm_layer, use_sigmoid=use_sigmoid)
    elif which_model_netD in ('n_layers_multi', 'n_layers_proj', 'pixel', 'pyramid', 'gan_stability',
                              'gan_stability_class', 'mnist_fc'):
        raise NotImplementedError('Discriminator [%s] is outside the MI355X hot path (SURVEY.md section 8); '
                                  'use basic / n_layers' % which_model_netD)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % which_model_netD)
    return init_net(netD, init_type, gpu_ids)


def define_IP(which_model_netIP, input_nc, gpu_ids=[]):
    """reference models/networks.py:179-193 (weights are not initialised here: they are loaded)."""
    if which_model_netIP == 'alexnet':
        netIP = AlexNetFeature(input_nc=input_nc, pooling='None')
    elif 'vgg' in which_model_netIP:
        raise NotImplementedError('Identity-preserving net [%s] is outside the MI355X hot path' % which_model_netIP)
    else:
        raise NotImplementedError('Identity-preserving model name [%s] is not recognized' % which_model_netIP)
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        netIP.to(torch.device('cuda', gpu_ids[0]))
    return netIP


def define_E(which_model_netE, input_nc=3, init_type='kaiming', pooling='max', cnn_dim=[], cnn_pad=1,
             cnn_relu_slope=0.2, gpu_ids=[], fine_size_E=224, noisy=False, bnn_dropout=0.):
    """reference models/networks.py:232-253"""
    drop_layer = get_dropout_layer(dropout=bnn_dropout)
    if which_model_netE == 'alexnet':
        base = AlexNetFeature(input_nc=input_nc, pooling='None')
    elif 'resnet' in which_model_netE:
        base = ResNetFeature(input_nc=input_nc, which_model=which_model_netE, dropout=bnn_dropout)
    elif which_model_netE in ('DTN', 'mnist_fc'):
        raise NotImplementedError('Encoder base [%s] is outside the MI355X hot path' % which_model_netE)
    else:
        raise NotImplementedError('Model [%s] is not implemented.' % which_model_netE)
    netE = SiameseFeature(base, pooling=pooling, cnn_dim=cnn_dim, cnn_pad=cnn_pad, cnn_relu_slope=cnn_relu_slope,
                          noisy=noisy, drop_layer=drop_layer)
    return init_net(netE, init_type, gpu_ids)


# ------------------------------------------------------------------------- losses
class GANLoss(tnn.Module):
    """reference models/networks.py:386-420: BCE (or MSE when use_lsgan) against a per-sample
    target (bool, list of 0/1 per sample, or -- extension -- a float device tensor [N])."""

    def __init__(self, use_lsgan=True, tensor=torch.FloatTensor):
        super().__init__()
        self.use_lsgan = use_lsgan
        self.Tensor = tensor
        self._const = {}

    def get_target_vector(self, input, target_label):
        if isinstance(target_label, torch.Tensor):
            return target_label.to(device=input.device, dtype=torch.float32).reshape(-1)
        if not isinstance(target_label, (list, tuple)):
            target_label = [target_label]
        vals = tuple(float(1 if t is True else (0 if t is False else t)) for t in target_label)
        key = (vals, input.device)
        if key not in self._const:   # tiny host->device copies, cached so that the step stays async
            self._const[key] = torch.tensor(np.array(vals, dtype=np.float32), device=input.device)
        return self._const[key]

    def get_target_tensor(self, input, target_label):
        t = self.get_target_vector(input, target_label)
        return t.reshape(-1, 1, 1, 1).expand_as(input)

    def __call__(self, inputs, target_label):
        if not isinstance(inputs, list):
            inputs = [inputs]
        loss = 0.0
        for input in inputs:
            if input.dim() < 4:
                input = input.view(input.size(0), -1, 1, 1)
            t = self.get_target_vector(input, target_label)
            if t.numel() == 1 and input.size(0) != 1:
                t = t.expand(input.size(0)).contiguous()
            if self.use_lsgan:
                loss = loss + HF.mse_loss(input, t.reshape(-1, 1, 1, 1).expand_as(input).contiguous())
            else:
                loss = loss + HF.bce_loss(input, t)
        return loss


class L1Loss(tnn.Module):
    """nn.L1Loss() of the reference model (models/wsgan_emb_model.py:141,149) on the HIP reduction."""

    def forward(self, a, b):
        return HF.l1_loss(a, b.detach())


class MSELoss(tnn.Module):
    """nn.MSELoss() (models/wsgan_emb_model.py:143,148)."""

    def forward(self, a, b):
        return HF.mse_loss(a, b.detach())


# ------------------------------------------------------------------------- generator
class ResnetGenerator(tnn.Module):
    """reference models/networks.py:565-612.  Sequential index layout (9 blocks):
    [0]RefPad3 [1]Conv7 [2]IN [3]ReLU [4]Conv3s2 [5]IN [6]ReLU [7]Conv3s2 [8]IN [9]ReLU
    [10..18]ResnetBlock [19]ConvT [20]IN [21]ReLU [22]ConvT [23]IN [24]ReLU [25]RefPad3 [26]Conv7 [27]Tanh."""

    def __init__(self, input_nc, output_nc, nz=0, ngf=64, norm_layer=hnn.BatchNorm2d, dropout=0, n_blocks=6,
                 padding_type='reflect'):
        assert n_blocks >= 0
        super().__init__()
        if dropout > 0:
            raise NotImplementedError('pcgan_amd: generator nn.Dropout (--dropout > 0) is outside the hot path')
        input_nc = input_nc + nz
        self.input_nc, self.output_nc, self.ngf, self.nz = input_nc, output_nc, ngf, nz
        use_bias = _is_instance_norm(norm_layer)
        model = [tnn.ReflectionPad2d(3), hnn.Conv2d(input_nc, ngf, kernel_size=7, padding=0, bias=use_bias),
                 norm_layer(ngf), tnn.ReLU(True)]
        for i in range(2):
            mult = 2 ** i
            model += [hnn.Conv2d(ngf * mult, ngf * mult * 2, kernel_size=3, stride=2, padding=1, bias=use_bias),
                      norm_layer(ngf * mult * 2), tnn.ReLU(True)]
        for _ in range(n_blocks):
            model += [ResnetBlock(ngf * 4, padding_type=padding_type, norm_layer=norm_layer, dropout=dropout,
                                  use_bias=use_bias)]
        for i in range(2):
            mult = 2 ** (2 - i)
            model += [hnn.ConvTranspose2d(ngf * mult, ngf * mult // 2, kernel_size=3, stride=2, padding=1,
                                          output_padding=1, bias=use_bias),
                      norm_layer(ngf * mult // 2), tnn.ReLU(True)]
        model += [tnn.ReflectionPad2d(3), hnn.Conv2d(ngf, output_nc, kernel_size=7, padding=0), tnn.Tanh()]
        self.model = tnn.Sequential(*model)

    def forward(self, input, z=None):
        x = HF.concat_z(input, z) if z is not None else input
        return run_sequential(self.model, x)


class ResnetBlock(tnn.Module):
    """reference models/networks.py:616-652: x + conv_block(x); the skip add is fused into the
    second InstanceNorm's apply pass."""

    def __init__(self, dim, padding_type, norm_layer, dropout, use_bias):
        super().__init__()
        if padding_type not in ('reflect', 'zero'):
            raise NotImplementedError('padding [%s] is not implemented' % padding_type)
        blk = []
        for half in range(2):
            p = 0
            if padding_type == 'reflect':
                blk += [tnn.ReflectionPad2d(1)]
            else:
                p = 1
            blk += [hnn.Conv2d(dim, dim, kernel_size=3, padding=p, bias=use_bias), norm_layer(dim)]
            if half == 0:
                blk += [tnn.ReLU(True)]
        self.conv_block = tnn.Sequential(*blk)
        # the composite call (hip/functional.py: _ResBlockFn) covers the reference's default block -- reflection padding,
        # InstanceNorm: [RefPad, Conv, IN, ReLU, RefPad, Conv, IN]; any other layout runs layer by layer
        self._composite_layout = (padding_type == 'reflect' and len(blk) == 7 and isinstance(blk[2], hnn.InstanceNorm2d)
                                  and isinstance(blk[6], hnn.InstanceNorm2d))

    def forward(self, x):
        if self._composite_layout:
            cb = self.conv_block
            pl = HF.resblock_composite_ok(x, cb[1], cb[5], cb[2], cb[6])
            if pl is not None:
                return HF.resblock(x, cb[1], cb[5], cb[2], cb[6], pl)
        return run_sequential(self.conv_block, x, residual=x)


# ------------------------------------------------------------------------- discriminator
class NLayerDiscriminator(tnn.Module):
    """reference models/networks.py:737-783: conditional PatchGAN; k4 convs (s2 x n_layers, s1, s1)."""

    def __init__(self, input_nc, nz, ndf=64, n_layers=3, norm_layer=hnn.BatchNorm2d, use_sigmoid=False):
        super().__init__()
        use_bias = _is_instance_norm(norm_layer)
        seq = [hnn.Conv2d(input_nc + nz, ndf, kernel_size=4, stride=2, padding=1), tnn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [hnn.Conv2d(ndf * prev, ndf * mult, kernel_size=4, stride=2, padding=1, bias=use_bias),
                    norm_layer(ndf * mult), tnn.LeakyReLU(0.2, True)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [hnn.Conv2d(ndf * prev, ndf * mult, kernel_size=4, stride=1, padding=1, bias=use_bias),
                norm_layer(ndf * mult), tnn.LeakyReLU(0.2, True)]
        seq += [hnn.Conv2d(ndf * mult, 1, kernel_size=4, stride=1, padding=1)]
        if use_sigmoid:
            seq += [tnn.Sigmoid()]
        self.model = tnn.Sequential(*seq)

    def forward(self, input, z=None):
        x = HF.concat_z(input, z) if z is not None else input
        return run_sequential(self.model, x)


# ------------------------------------------------------------------------- encoders
class SiameseFeature(tnn.Module):
    """reference models/networks.py:1008-1083: base trunk -> small conv head -> global pooling
    (-> twin `cnn_logvar` head when noisy)."""

    def __init__(self, base=None, pooling='avg', cnn_dim=[], cnn_pad=1, cnn_relu_slope=0.2, noisy=False,
                 drop_layer=None):
        super().__init__()
        self.pooling = pooling
        self.base = base
        self._noisy = noisy
        drop_layer = drop_layer or IdentityMapping

        def head():
            blk, prev = [], base.feature_dim
            for nf in cnn_dim[:-1]:
                blk += [hnn.Conv2d(prev, nf, kernel_size=3, stride=1, padding=cnn_pad, bias=True),
                        hnn.BatchNorm2d(nf), drop_layer(), tnn.LeakyReLU(cnn_relu_slope)]
                prev = nf
            blk += [hnn.Conv2d(prev, cnn_dim[-1], kernel_size=3, stride=1, padding=cnn_pad, bias=True)]
            return tnn.Sequential(*blk)

        if cnn_dim:
            self.cnn = head()
            self.feature_dim = cnn_dim[-1]
        else:
            self.cnn = None
            self.feature_dim = base.feature_dim
        if noisy:
            assert cnn_dim
            self.cnn_logvar = head()

    def _pool(self, t):
        if self.pooling in ('avg', 'max'):
            assert t.size(2) == t.size(3), 'global pooling expects square feature maps'
            return HF.global_pool(t, self.pooling == 'max')
        return t

    def forward(self, x):
        h = self.base.forward(x)
        out = run_sequential(self.cnn, h) if self.cnn is not None else h
        # ratings leave the encoder as fp32 whatever the storage type of its activations (the bf16 path: the rating
        # arithmetic of the step -- normalisation, resampling, bin look-ups, the z_rec loss -- stays fp32)
        out = HF.cast(self._pool(out), torch.float32)
        if self._noisy:
            return out, HF.cast(self._pool(run_sequential(self.cnn_logvar, h)), torch.float32)
        return out

    def load_pretrained(self, state_dict):
        if isinstance(state_dict, str):
            state_dict = torch.load(state_dict, map_location='cpu')
        for key in list(state_dict.keys()):
            if key.startswith('cxn') or key.startswith('fc'):
                state_dict.pop(key)
        self.load_state_dict(state_dict, strict=True)

    def load_base(self, state_dict):
        self.base.load_pretrained(state_dict)


class SiameseNetwork(SiameseFeature):
    """The comparison network the Elo-rating encoder is trained as (reference models/networks.py:872-992): the SAME trunk
    + conv head as SiameseFeature applied to two images, score = rating(x1) - rating(x2).  Its state_dict (`base.*`,
    `cnn.*`) is what wsgan_emb consumes as --pretrained_model_path_E.  Configurations of the reference outside the
    Elo recipe (fully connected comparison head `fc_dim`, `use_cxn`) raise."""

    def __init__(self, base=None, pooling='avg', cnn_dim=[], cnn_pad=1, cnn_relu_slope=0.5, fc_dim=[], fc_relu_slope=0.2,
                 fc_residual=True, dropout=0.5, use_cxn=False, noisy=False, drop_layer=None, rsample=False):
        if fc_dim or use_cxn:
            raise NotImplementedError('pcgan_amd: SiameseNetwork with fc_dim / use_cxn is outside the MI355X hot path')
        super().__init__(base, pooling=pooling, cnn_dim=cnn_dim, cnn_pad=cnn_pad, cnn_relu_slope=cnn_relu_slope,
                         noisy=noisy, drop_layer=drop_layer)
        self._rsample = rsample
        self.fc = self.cxn = None

    def forward_once(self, x):
        out = SiameseFeature.forward(self, x)
        return out if self._noisy else (out, None)

    def forward(self, input1, input2):
        feature1, logvar1 = self.forward_once(input1)
        feature2, logvar2 = self.forward_once(input2)
        if not self._noisy:
            return feature1, feature2, feature1 - feature2
        if self._rsample:
            return feature1, feature2, logvar1, logvar2
        std = torch.sqrt(torch.exp(logvar1) + torch.exp(logvar2))      # sqrt(std1^2 + std2^2)
        return feature1, feature2, feature1 - feature2, std

    def load_pretrained(self, state_dict):
        """only the trunk comes from the pretrained (ImageNet) file: the conv head stays as initialised (:994-997)"""
        self.base.load_pretrained(state_dict)

    def get_finetune_parameters(self):
        return list(self.cnn.parameters()) if self.cnn is not None else []


class BinaryNLLLoss(tnn.Module):
    """Binary cross entropy with draws (reference models/networks.py:473-482): label 0 / 1 / 2 = "first lower" / draw /
    "first higher" -> target 0 / 0.5 / 1; -(t log(p + 1e-20) + (1 - t) log(1 - p + 1e-20)), mean.  (N values: plain
    tensor arithmetic; the LUT follows the probabilities' device instead of the reference's unconditional .cuda().)"""

    def __init__(self):
        super().__init__()
        self.register_buffer('LUT', torch.tensor([0.0, 0.5, 1.0]), persistent=False)

    def forward(self, prob, label):
        lut = self.LUT.to(prob.device)
        target = lut[label].reshape(prob.size(0), 1, 1, 1).expand(prob.size(0), 1, prob.size(2), prob.size(3))
        loss = -(target * torch.log(prob + 1e-20) + (1 - target) * torch.log(1 - prob + 1e-20))
        return loss.mean()


class ResNetFeature(tnn.Module):
    """reference models/networks.py:1310-1359"""

    def __init__(self, input_nc=3, which_model='resnet18', dropout=0.):
        super().__init__()
        from . import resnet
        table = {'resnet18': (resnet.resnet18, 512), 'resnet34': (resnet.resnet34, 512),
                 'resnet50': (resnet.resnet50, 2048)}
        if which_model not in table:
            raise NotImplementedError('pcgan_amd: encoder trunk [%s] is outside the hot path' % which_model)
        ctor, dim = table[which_model]
        model = ctor(False, dropout=dropout)
        del model.fc
        self.model = model
        self.feature_dim = dim

    def forward(self, x):
        return self.model.features(x)

    def load_pretrained(self, state_dict):
        if isinstance(state_dict, str):
            state_dict = torch.load(state_dict, map_location='cpu')
        self.model.load_state_dict(state_dict, strict=False)


class AlexNetFeature(tnn.Module):
    """reference models/networks.py:1218-1255"""

    def __init__(self, input_nc=3, pooling='max'):
        super().__init__()
        self.pooling = pooling
        self.features = tnn.Sequential(
            hnn.Conv2d(input_nc, 64, kernel_size=11, stride=4, padding=2), tnn.ReLU(inplace=True),
            hnn.MaxPool2d(kernel_size=3, stride=2),
            hnn.Conv2d(64, 192, kernel_size=5, padding=2), tnn.ReLU(inplace=True),
            hnn.MaxPool2d(kernel_size=3, stride=2),
            hnn.Conv2d(192, 384, kernel_size=3, padding=1), tnn.ReLU(inplace=True),
            hnn.Conv2d(384, 256, kernel_size=3, padding=1), tnn.ReLU(inplace=True),
            hnn.Conv2d(256, 256, kernel_size=3, padding=1), tnn.ReLU(inplace=True),
            hnn.MaxPool2d(kernel_size=3, stride=2))
        self.feature_dim = 256

    def forward(self, x):
        x = run_sequential(self.features, x)
        if self.pooling in ('avg', 'max'):
            x = HF.global_pool(x, self.pooling == 'max')
        return x

    def load_pretrained(self, state_dict):
        if isinstance(state_dict, str):
            state_dict = torch.load(state_dict, map_location='cpu')
        for key in list(state_dict.keys()):
            if key.startswith('classifier'):
                state_dict.pop(key)
        self.load_state_dict(state_dict, strict=True)


class Normalize(tnn.Module):
    """reference models/networks.py:2421-2439.  With non-empty mean/std the reference's
    `identity_mapping = mean or std` is truthy, so the module is an IDENTITY; only
    that reachable behaviour is reproduced."""

    def __init__(self, mean=[], std=[]):
        super().__init__()
        self.nc = len(mean)
        self.identity_mapping = bool(mean or std)
        if not self.identity_mapping:
            raise NotImplementedError('pcgan_amd: Normalize with empty mean and std crashes in the reference too')

    def forward(self, input):
        return input
