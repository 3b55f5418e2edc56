"""ORACLE (test infrastructure, not product code).

CPU restatement of the reference's network graphs on the hot path, built from stock
torch.nn modules (the reference's own third-party dependency), with the reference's
nn.Sequential index layout so one state_dict fits the reference, this oracle and the HIP
product alike.  Pinned against golden vectors captured by importing the reference in the
build container (oracle/make_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).

Each class cites the reference lines it restates (relative to the reference root).
"""
import functools

import torch
import torch.nn as nn
import torch.nn.functional as F


def norm_layer_of(kind):
    """models/networks.py:22-34"""
    if kind == 'batch':
        return functools.partial(nn.BatchNorm2d, affine=True)
    if kind == 'instance':
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=True)
    raise NotImplementedError(kind)


class Identity(nn.Module):
    """models/networks.py:2407-2412"""

    def __init__(self, *a):
        super().__init__()

    def forward(self, x):
        return x


class DecisionTape:
    """Test instrument for GRADIENT parity.  ReLU / LeakyReLU sign masks and max-pool arg-max choices make a network's
    gradient a discontinuous function of its input: two correct fp32 implementations (and the float64 twin) flip a handful
    of these decisions where a pre-activation is within rounding of zero, which moves whole gradient tensors by 1e-3..1e-2
    although every operator agrees to 1e-6.  With `replay` set to the decisions ANOTHER forward pass took (the HIP run's,
    recorded in call order) the modules below apply those instead of their own, so the twin differentiates the same smooth
    branch of the function and gradients can be held to operator-level tolerances.  With `replay` None (always, outside
    the tests that set it) the modules ARE the stock torch modules of the reference."""
    replay = None

    @staticmethod
    def take(like):
        t = next(DecisionTape.replay)
        assert tuple(t.shape) == tuple(like.shape), 'decision tape out of step: %s vs %s' % (tuple(t.shape), tuple(like.shape))
        return t

    @staticmethod
    def bind(net, queue):
        """every forward call of `net` replays from `queue` (one iterator per network: a training step calls its nets in
        an interleaved order, but each net sees its own calls in the same order on both sides); queue None unbinds"""
        if hasattr(net, '_tape_forward'):
            net.forward = net._tape_forward
            del net._tape_forward
        if queue is None:
            return
        orig = net._tape_forward = net.forward

        def forward(*a, **kw):
            prev, DecisionTape.replay = DecisionTape.replay, queue
            try:
                return orig(*a, **kw)
            finally:
                DecisionTape.replay = prev
        net.forward = forward


class TapedReLU(nn.ReLU):
    def forward(self, x):
        if DecisionTape.replay is None:
            return super().forward(x)
        return x * DecisionTape.take(x).to(x.dtype)


class TapedLeakyReLU(nn.LeakyReLU):
    def forward(self, x):
        if DecisionTape.replay is None:
            return super().forward(x)
        return torch.where(DecisionTape.take(x).bool(), x, x * self.negative_slope)


def _gather_planes(x, idx):
    """x[n][c] at the per-plane flat indices idx[n][c][...] (the arg-max another pass chose)"""
    return x.flatten(2).gather(2, idx.flatten(2).long()).view(idx.shape)


class TapedMaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        if DecisionTape.replay is None:
            return super().forward(x)
        return _gather_planes(x, next(DecisionTape.replay))


def global_max_pool(t):
    """F.max_pool2d(t, t.size(2)) (models/networks.py:1056-1059), replayable like the modules above"""
    if DecisionTape.replay is None:
        return F.max_pool2d(t, t.size(2))
    return _gather_planes(t, next(DecisionTape.replay).view(t.size(0), t.size(1), 1, 1))


class Dropout2dRec(nn.Module):
    """nn.Dropout2d (models/resnet.py:38-41, models/networks.py:37-42) with the Bernoulli keep-mask
    drawn explicitly -- one (N,C,1,1) bernoulli_(1-p) draw, which is what F.dropout2d consumes from
    the CPU generator (verified against the reference's bayesian golden vectors) -- and recorded in
    `Dropout2dRec.record` so the GPU product can be fed the same masks."""
    record = None      # set to a list to record masks
    inject = None      # set to an iterator of masks to replay instead of drawing

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p <= 0:
            return x
        if Dropout2dRec.inject is not None:
            mask = next(Dropout2dRec.inject).to(x.dtype).view(x.size(0), x.size(1), 1, 1)
        else:
            mask = torch.empty(x.size(0), x.size(1), 1, 1, dtype=x.dtype).bernoulli_(1 - self.p)
        if Dropout2dRec.record is not None:
            Dropout2dRec.record.append(mask.detach().reshape(-1).clone())
        return x * (mask / (1 - self.p))     # torch: noise.div_(1-p); input * noise


def _cat_z(x, z):
    """models/networks.py:610-611"""
    if z is None:
        return x
    zi = z.view(z.size(0), z.size(1), 1, 1).expand(x.size(0), z.size(1), x.size(2), x.size(3))
    return torch.cat((x, zi), 1)


class ResBlockRef(nn.Module):
    """models/networks.py:616-652 (reflect padding, no dropout)"""

    def __init__(self, dim, norm, bias):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, padding=0, bias=bias), norm(dim), TapedReLU(True),
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, padding=0, bias=bias), norm(dim))

    def forward(self, x):
        return x + self.conv_block(x)


class ResnetGeneratorRef(nn.Module):
    """models/networks.py:565-612"""

    def __init__(self, input_nc, output_nc, nz=1, ngf=64, norm='instance', n_blocks=9):
        super().__init__()
        nl = norm_layer_of(norm)
        bias = norm == 'instance'
        m = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc + nz, ngf, 7, padding=0, bias=bias), nl(ngf), TapedReLU(True)]
        for i in range(2):
            c = ngf * 2 ** i
            m += [nn.Conv2d(c, 2 * c, 3, stride=2, padding=1, bias=bias), nl(2 * c), TapedReLU(True)]
        m += [ResBlockRef(ngf * 4, nl, bias) for _ in range(n_blocks)]
        for i in range(2):
            c = ngf * 2 ** (2 - i)
            m += [nn.ConvTranspose2d(c, c // 2, 3, stride=2, padding=1, output_padding=1, bias=bias), nl(c // 2),
                  TapedReLU(True)]
        m += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7, padding=0), nn.Tanh()]
        self.model = nn.Sequential(*m)

    def forward(self, x, z=None):
        return self.model(_cat_z(x, z))


class NLayerDiscriminatorRef(nn.Module):
    """models/networks.py:737-783"""

    def __init__(self, input_nc, nz, ndf=64, n_layers=3, norm='batch', use_sigmoid=True):
        super().__init__()
        nl = norm_layer_of(norm)
        bias = norm == 'instance'
        s = [nn.Conv2d(input_nc + nz, ndf, 4, stride=2, padding=1), TapedLeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            s += [nn.Conv2d(ndf * prev, ndf * mult, 4, stride=2, padding=1, bias=bias), nl(ndf * mult),
                  TapedLeakyReLU(0.2, True)]
        prev, mult = mult, min(2 ** n_layers, 8)
        s += [nn.Conv2d(ndf * prev, ndf * mult, 4, stride=1, padding=1, bias=bias), nl(ndf * mult),
              TapedLeakyReLU(0.2, True), nn.Conv2d(ndf * mult, 1, 4, stride=1, padding=1)]
        if use_sigmoid:
            s += [nn.Sigmoid()]
        self.model = nn.Sequential(*s)

    def forward(self, x, z=None):
        return self.model(_cat_z(x, z))


class BasicBlockRef(nn.Module):
    """models/resnet.py:31-73 -- note Dropout2d BETWEEN conv and BN."""

    def __init__(self, cin, planes, stride=1, downsample=None, dropout=0.):
        super().__init__()
        drop = (lambda: Dropout2dRec(dropout)) if dropout > 0 else Identity
        self.conv1 = nn.Conv2d(cin, planes, 3, stride=stride, padding=1, bias=False)
        self.drop1 = drop()
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = TapedReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.drop2 = drop()
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.drop1(self.conv1(x))))
        out = self.bn2(self.drop2(self.conv2(out)))
        return self.relu(out + idt)


class ResNet18TrunkRef(nn.Module):
    """models/resnet.py:125-190 with layers [2,2,2,2] and `fc` removed (models/networks.py:1337)."""

    def __init__(self, dropout=0., layers=(2, 2, 2, 2)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = TapedReLU(inplace=True)
        self.maxpool = TapedMaxPool2d(3, stride=2, padding=1)
        cin = 64
        for li, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2)), 1):
            blocks = []
            for b in range(n):
                s = stride if b == 0 else 1
                down = None
                if s != 1 or cin != planes:
                    down = nn.Sequential(nn.Conv2d(cin, planes, 1, stride=s, bias=False), nn.BatchNorm2d(planes))
                blocks.append(BasicBlockRef(cin, planes, s, down, dropout))
                cin = planes
            setattr(self, 'layer%d' % li, nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))


class ResNetFeatureRef(nn.Module):
    """models/networks.py:1310-1354"""

    def __init__(self, which='resnet18', dropout=0.):
        super().__init__()
        layers = {'resnet18': (2, 2, 2, 2), 'resnet34': (3, 4, 6, 3)}[which]
        self.model = ResNet18TrunkRef(dropout, layers)
        self.feature_dim = 512

    def forward(self, x):
        m = self.model
        x = m.maxpool(m.relu(m.bn1(m.conv1(x))))
        return m.layer4(m.layer3(m.layer2(m.layer1(x))))


class AlexNetFeatureRef(nn.Module):
    """models/networks.py:1218-1246"""

    def __init__(self, input_nc=3, pooling='None'):
        super().__init__()
        self.pooling = pooling
        self.features = nn.Sequential(
            nn.Conv2d(input_nc, 64, 11, stride=4, padding=2), TapedReLU(inplace=True), TapedMaxPool2d(3, 2),
            nn.Conv2d(64, 192, 5, padding=2), TapedReLU(inplace=True), TapedMaxPool2d(3, 2),
            nn.Conv2d(192, 384, 3, padding=1), TapedReLU(inplace=True),
            nn.Conv2d(384, 256, 3, padding=1), TapedReLU(inplace=True),
            nn.Conv2d(256, 256, 3, padding=1), TapedReLU(inplace=True), TapedMaxPool2d(3, 2))
        self.feature_dim = 256

    def forward(self, x):
        x = self.features(x)
        if self.pooling == 'avg':
            x = F.avg_pool2d(x, x.size(2))
        elif self.pooling == 'max':
            x = global_max_pool(x)
        return x


class SiameseFeatureRef(nn.Module):
    """models/networks.py:1008-1068"""

    def __init__(self, base, pooling='avg', cnn_dim=(32, 1), cnn_pad=1, slope=0.7, noisy=False, dropout=0.):
        super().__init__()
        self.base, self.pooling, self._noisy = base, pooling, noisy
        drop = (lambda: Dropout2dRec(dropout)) if dropout > 0 else Identity

        def head():
            blk, prev = [], base.feature_dim
            for nf in cnn_dim[:-1]:
                blk += [nn.Conv2d(prev, nf, 3, padding=cnn_pad), nn.BatchNorm2d(nf), drop(), TapedLeakyReLU(slope)]
                prev = nf
            return nn.Sequential(*blk, nn.Conv2d(prev, cnn_dim[-1], 3, padding=cnn_pad))

        self.cnn = head() if cnn_dim else None
        if noisy:
            self.cnn_logvar = head()

    def _pool(self, t):
        if self.pooling == 'avg':
            return F.avg_pool2d(t, t.size(2))
        if self.pooling == 'max':
            return global_max_pool(t)
        return t

    def forward(self, x):
        h = self.base(x)
        out = self._pool(self.cnn(h) if self.cnn is not None else h)
        if self._noisy:
            return out, self._pool(self.cnn_logvar(h))
        return out
