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


def test_capture_follows_mode_and_weights_and_survives_a_failed_capture(dev, monkeypatch):
    """a capture is keyed on the train / eval pattern and on the parameter versions: eval() after a capture must not
    replay the train-mode launches (batch statistics, running-statistics updates), load_state_dict must not replay stale packed
    weights; the flag arena is sized from the warm-up calls (a large MC-dropout batch fits); a failing capture leaves the call eager."""
    from pcgan_amd.hip import graphs
    e1 = _encoder(dev, 0.0)
    e2 = copy.deepcopy(e1)
    g = graphs.GraphedNoGrad(e2)
    x = (torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(5)) * 2 - 1).to(dev)
    with torch.no_grad():
        for _ in range(4):
            assert torch.equal(e1(x), g(x))
        e1.eval(), e2.eval()
        n0 = int(e2.state_dict()['base.model.bn1.num_batches_tracked'])
        for _ in range(4):
            assert torch.equal(e1(x), g(x)), 'eval mode after a train-mode capture'
        assert int(e2.state_dict()['base.model.bn1.num_batches_tracked']) == n0, 'an eval pass must not count batches'
        e1.train(), e2.train()
        sd = {k: (v * 1.5 if k.endswith('conv1.weight') else v) for k, v in e1.state_dict().items()}
        e1.load_state_dict(sd), e2.load_state_dict(sd)
        for _ in range(4):
            assert torch.equal(e1(x), g(x)), 'new weights after a capture'
    # arena sized from the warm-up: more flags than the old fixed 1 << 17 floats
    big = _encoder(dev, 0.25)
    gb = graphs.GraphedNoGrad(big)
    xb = torch.rand(40, 3, 64, 64).to(dev)
    before = dict(graphs.STATS)
    with torch.no_grad():
        for _ in range(4):
            yb, _lv = gb(xb)
    assert graphs.STATS['captured'] == before['captured'] + 1 and graphs.STATS.get('capture_failed', 0) == before.get('capture_failed', 0)
    assert max(ent.get('flags', 0) for ent in gb.state.values()) > (1 << 17)
    # a capture that raises: the key turns eager-only, the call still answers, nothing is retried
    def boom(self, n, p):
        raise RuntimeError('arena exhausted (test)')
    monkeypatch.setattr(graphs.FlagArena, 'take', boom)
    small = _encoder(dev, 0.25)
    gs = graphs.GraphedNoGrad(small)
    with torch.no_grad(), pytest.warns(UserWarning):
        for _ in range(4):
            ys, _lv = gs(x)
            assert torch.isfinite(ys).all()
    assert graphs.STATS.get('capture_failed', 0) == before.get('capture_failed', 0) + 1


def test_flag_pool_follows_the_seed(dev):
    """pre-drawn Dropout2d keep flags are dropped on torch.manual_seed(): a seeded MC-dropout run is reproducible from its seed"""
    from pcgan_amd.hip import functional as F
    torch.manual_seed(11)
    a = F._keep_flags(1000, 0.2, dev).clone()
    F._keep_flags(500, 0.2, dev)
    torch.manual_seed(11)
    b = F._keep_flags(1000, 0.2, dev).clone()
    assert torch.equal(a, b)
