"""hipGraph replay of a frozen net's no-grad forward (pc-gan_amd/hip/graphs.py): the replay must be the eager launch sequence --
same outputs bit for bit, same in-place BatchNorm running statistics and batch counters -- and Dropout2d inside it must see fresh,
independent keep flags per replay (one refill of the static flag buffer in front of each)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _encoder(dev, p):
    from pcgan_amd.models import networks
    torch.manual_seed(3)
    return networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7, noisy=p > 0, bnn_dropout=p).to(dev)


def test_replay_is_the_eager_forward(dev):
    from pcgan_amd.hip import graphs
    e1 = _encoder(dev, 0.0)
    e2 = copy.deepcopy(e1)
    g = graphs.GraphedNoGrad(e2)
    before = dict(graphs.STATS)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for it in range(6):
            x = (torch.rand(4, 3, 64, 64, generator=gen) * 2 - 1).to(dev)
            y1, y2 = e1(x), g(x)
            assert torch.equal(y1, y2), 'call %d' % it
    assert graphs.STATS['captured'] == before['captured'] + 1 and graphs.STATS['replayed'] == before['replayed'] + 4
    s1, s2 = e1.state_dict(), e2.state_dict()
    for k in s1:      # train-mode BatchNorm: running statistics and batch counters moved identically
        assert torch.equal(s1[k], s2[k]), k
    assert int(s2['base.model.bn1.num_batches_tracked']) == 6
    # while autograd is recording the call is the eager forward
    with torch.enable_grad():
        y = g(torch.rand(4, 3, 64, 64).to(dev).requires_grad_(True))
        assert y.requires_grad


def test_dropout_flags_inside_a_replay(dev, monkeypatch):
    from pcgan_amd.hip import graphs, nn as hnn
    p = 0.3
    e1 = _encoder(dev, p)
    e2 = copy.deepcopy(e1)
    g = graphs.GraphedNoGrad(e2, warm=0)
    x = (torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(2)) * 2 - 1).to(dev)
    filled = []
    orig = graphs.FlagArena.refill

    def refill(self):
        orig(self)
        filled.append(self.buf[:self.used].clone())
    monkeypatch.setattr(graphs.FlagArena, 'refill', refill)
    with torch.no_grad():
        outs = [g(x) for _ in range(3)]
        # the eager net fed the flags of each replay (in Dropout2d call order) reproduces it bit for bit
        for flags, (y, lv) in zip(filled, outs):
            # slice the flat flag buffer per site in call order: record the sizes with one eager pass
            sizes = []
            hook = hnn.Dropout2d.forward

            def rec(self_, t):
                sizes.append(t.shape[0] * t.shape[1])
                return hook(self_, t)
            monkeypatch.setattr(hnn.Dropout2d, 'forward', rec)
            e3 = copy.deepcopy(e1)
            e3(x)
            monkeypatch.setattr(hnn.Dropout2d, 'forward', hook)
            assert sum(sizes) == flags.numel()
            parts, o = [], 0
            for n in sizes:
                parts.append(flags[o:o + n])
                o += n
            hnn.Dropout2d.mask_source = iter(parts)
            try:
                y1, lv1 = e1(x)
            finally:
                hnn.Dropout2d.mask_source = None
            assert torch.equal(y1, y) and torch.equal(lv1, lv)
    assert len(filled) == 3 and not torch.equal(filled[0], filled[1]), 'every replay draws its own flags'
    keep = torch.cat(filled).mean().item()
    assert abs(keep - (1 - p)) < 0.03, keep
