import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from oracle import networks_ref as N, weights as W
from pcgan_amd.models import networks
dev = torch.device('cuda:0')
def rl2(a, b): return float((a.double().cpu() - b.double()).norm() / b.double().norm())
for (bs, smooth, freeze) in [(3, False, False), (4, False, False), (4, True, False), (4, True, True)]:
    ref = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
    ref.load_state_dict(W.fill_state_dict(ref.state_dict(), 30))
    hip = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
    hip.load_state_dict(ref.state_dict()); hip.to(dev)
    x = W.seeded_tensor((bs, 3, 32, 32), 7)
    if smooth:
        x = torch.tanh(F.avg_pool2d(F.pad(x, (2, 2, 2, 2), mode='reflect'), 5, 1) * 3)
    if freeze:
        for p in hip.parameters(): p.requires_grad = False
    target = W.seeded_normal((bs, 1, 1, 1), 9)
    outs = {}
    hooks = {}
    for name, net, dt, d in (('hip', hip, torch.float32, dev), ('c32', ref, torch.float32, 'cpu'), ('c64', None, torch.float64, 'cpu')):
        if net is None:
            net = ref.double()
        xi = x.detach().clone().to(dt).to(d).requires_grad_(True)
        up = F.interpolate(xi, size=(64, 64), mode='bilinear', align_corners=True) if name != 'hip' else None
        if name == 'hip':
            from pcgan_amd.hip import functional as HF
            up = HF.upsample2d(xi, 64)
        y = net(up)
        loss = F.mse_loss(y, target.to(dt).to(d)) if name != 'hip' else HF.mse_loss(y, target.to(d))
        loss.backward()
        outs[name] = (y.detach().cpu(), xi.grad.detach().cpu())
    print('bs%d smooth%d freeze%d: y hip %.2e c32 %.2e | din hip %.3e c32 %.3e' % (
        bs, smooth, freeze, rl2(outs['hip'][0], outs['c64'][0]), rl2(outs['c32'][0], outs['c64'][0]),
        rl2(outs['hip'][1], outs['c64'][1]), rl2(outs['c32'][1], outs['c64'][1])))
