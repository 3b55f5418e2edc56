"""nn.Module layer set whose forward/backward run on the HIP kernels.

The classes subclass the stock torch.nn containers only to inherit parameter/buffer
registration, default initialisation and state_dict key names (so reference checkpoints
load unchanged: SURVEY.md appendix B); every forward() dispatches to
pcgan_amd.hip.functional.  `run_sequential` executes an nn.Sequential laid out exactly like
the reference's (same indices => same state_dict keys) with the obvious fusions:
  ReflectionPad2d(p) + Conv2d           -> one conv launch with reflected gather indices
  Conv2d + {ReLU, LeakyReLU, Tanh, Sigmoid} (no norm between) -> activation in the epilogue
  {InstanceNorm2d, BatchNorm2d} + [Dropout2d] + activation    -> one normalise+activate pass
"""
import torch
import torch.nn as tnn

import os

from . import functional as F
from .lib import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID


BN_TICKETS = os.environ.get('PCGAN_BN_TICKETS', '1') != '0'      # one-launch statistics for the large BatchNorm tensors (A/B switch)


class Conv2d(tnn.Conv2d):
    """nn.Conv2d on the implicit-GEMM MFMA kernel (square stride / padding, groups=1)."""

    def forward(self, x, pad_mode=0, extra_pad=0, act=ACT_NONE, slope=0.0):
        assert self.groups == 1 and self.dilation == (1, 1) and self.padding_mode == 'zeros'
        assert self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1]
        pad = self.padding[0]
        if pad_mode == 1:
            assert pad == 0, 'reflection padding replaces the conv padding'
            pad = extra_pad
        return F.conv2d(x, self.weight, self.bias, self.stride[0], pad, pad_mode, act, slope)


class ConvTranspose2d(tnn.ConvTranspose2d):
    def forward(self, x):
        assert self.groups == 1 and self.dilation == (1, 1)
        assert self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1]
        return F.conv_transpose2d(x, self.weight, self.bias, self.stride[0], self.padding[0],
                                  self.output_padding[0])


class InstanceNorm2d(tnn.InstanceNorm2d):
    """InstanceNorm2d(affine=False, track_running_stats=True) as the reference builds it
    (models/networks.py:26).  Running statistics are updated but (in train mode) unused."""

    def forward(self, x, act=ACT_NONE, slope=0.0, residual=None):
        if self.affine:
            raise NotImplementedError('pcgan_amd: InstanceNorm2d(affine=True) is outside the hot path')
        use_input_stats = self.training or not self.track_running_stats
        return F.instance_norm_act(x, self.running_mean, self.running_var, self.momentum, self.eps, act, slope,
                                   residual, use_input_stats)


class BatchNorm2d(tnn.BatchNorm2d):
    """BatchNorm2d(affine=True); train mode uses batch statistics (the reference never calls
    .eval() while training)."""

    def _tickets(self, device):
        """arrival counters of the one-launch statistics kernels (forward / backward), zeroed once (csrc/norm.hip: last arriver)"""
        t = self.__dict__.get('_pcgan_tickets')
        if t is None or t[0].device != device:
            buf = torch.zeros(2 * self.num_features, dtype=torch.int32, device=device)
            t = self.__dict__['_pcgan_tickets'] = (buf[:self.num_features], buf[self.num_features:])
        return t

    def forward(self, x, act=ACT_NONE, slope=0.0, residual=None):
        training = self.training or self.running_mean is None
        tickets = self._tickets(x.device) if (training and x.is_cuda and self.running_mean is not None and BN_TICKETS) else None
        return F.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum,
                                self.eps, act, slope, residual, training,
                                self.num_batches_tracked if training else None, tickets)


class Dropout2d(tnn.Dropout2d):
    """nn.Dropout2d; `Dropout2d.mask_source` (an iterator of N*C keep-flag tensors, consumed in call
    order) lets a parity test replay the masks the oracle drew."""
    mask_source = None
    flag_arena = None      # hip/graphs.py: static flag storage while a forward is captured into a hipGraph
    flag_demand = None     # hip/graphs.py: [count] -- the warm-up calls add the flags every site draws (sizes the arena)

    def forward(self, x):
        mask = None
        if self.training and self.p > 0:
            if Dropout2d.flag_demand is not None:
                Dropout2d.flag_demand[0] += x.shape[0] * x.shape[1]
            if Dropout2d.mask_source is not None:
                mask = next(Dropout2d.mask_source).to(device=x.device, dtype=torch.float32)
            elif Dropout2d.flag_arena is not None:
                mask = Dropout2d.flag_arena.take(x.shape[0] * x.shape[1], float(self.p))
        return F.dropout2d(x, self.p, self.training, mask)


class MaxPool2d(tnn.MaxPool2d):
    def forward(self, x):
        k = self.kernel_size if isinstance(self.kernel_size, int) else self.kernel_size[0]
        s = self.stride if isinstance(self.stride, int) else self.stride[0]
        p = self.padding if isinstance(self.padding, int) else self.padding[0]
        return F.max_pool2d(x, k, s, p)


class IdentityMapping(tnn.Module):
    """models/networks.py:2407-2412."""

    def __init__(self, *args):
        super().__init__()

    def forward(self, x):
        return x


_ACT_OF = {
    tnn.ReLU: lambda m: (ACT_RELU, 0.0),
    tnn.LeakyReLU: lambda m: (ACT_LRELU, m.negative_slope),
    tnn.Tanh: lambda m: (ACT_TANH, 0.0),
    tnn.Sigmoid: lambda m: (ACT_SIGMOID, 0.0),
}


def _act_of(m):
    for cls, fn in _ACT_OF.items():
        if isinstance(m, cls):
            return fn(m)
    return None


def _is_passthrough(m):
    return isinstance(m, IdentityMapping) or isinstance(m, tnn.Identity) or \
        (isinstance(m, (tnn.Dropout, tnn.Dropout2d)) and (m.p <= 0.0 or not m.training))


def run_sequential(seq, x, residual=None):
    """Execute an nn.Sequential of the layer types above with fusion.  `residual`, if given, is
    added by the LAST normalisation layer of the sequence (ResnetBlock skip connection)."""
    mods = list(seq)
    last_norm = -1
    if residual is not None:
        for j, m in enumerate(mods):
            if isinstance(m, (InstanceNorm2d, BatchNorm2d)):
                last_norm = j
        assert last_norm >= 0, 'residual fusion needs a normalisation layer'
    i, n = 0, len(mods)
    while i < n:
        m = mods[i]
        if _is_passthrough(m):
            i += 1
        elif isinstance(m, tnn.ReflectionPad2d):
            conv = mods[i + 1]
            assert isinstance(conv, Conv2d), 'ReflectionPad2d must be followed by a Conv2d'
            p = m.padding[0]
            act = _act_of(mods[i + 2]) if i + 2 < n else None
            if act is not None:
                x = conv(x, 1, p, act[0], act[1])
                i += 3
            else:
                x = conv(x, 1, p)
                i += 2
        elif isinstance(m, Conv2d):
            act = _act_of(mods[i + 1]) if i + 1 < n else None
            if act is not None:
                x = m(x, 0, 0, act[0], act[1])
                i += 2
            else:
                x = m(x)
                i += 1
        elif isinstance(m, (InstanceNorm2d, BatchNorm2d)):
            j = i + 1
            drop = None
            while j < n and (_is_passthrough(mods[j]) or isinstance(mods[j], Dropout2d)):
                if isinstance(mods[j], Dropout2d) and not _is_passthrough(mods[j]):
                    drop = mods[j]
                j += 1
            act = _act_of(mods[j]) if j < n else None
            res = residual if i == last_norm else None
            if drop is not None:
                # norm -> dropout -> activation (Elo head, models/networks.py:1020-1024): the
                # dropout sits between, so the activation cannot be fused into the norm pass
                x = m(x, ACT_NONE, 0.0, res)
                x = drop(x)
                if act is not None:
                    x = F.activation(x, act[0], act[1])
                    j += 1
            elif act is not None:
                x = m(x, act[0], act[1], res)
                j += 1
            else:
                x = m(x, ACT_NONE, 0.0, res)
            i = j
        elif _act_of(m) is not None:
            a = _act_of(m)
            x = F.activation(x, a[0], a[1])
            i += 1
        elif getattr(m, '_composite_layout', False):
            # a run of ResnetBlocks (the generator's nine): one library call and one autograd node for the chain when every block
            # qualifies for the composite path (hip/functional.py: restrunk), else block by block
            j = i
            while j < n and getattr(mods[j], '_composite_layout', False):
                j += 1
            y = F.restrunk(x, mods[i:j]) if j - i >= 2 else None
            if y is None:
                for k in range(i, j):
                    x = mods[k](x)
            else:
                x = y
            i = j
        else:
            x = m(x)   # ConvTranspose2d, MaxPool2d, Dropout2d, nested blocks ...
            i += 1
    return x
