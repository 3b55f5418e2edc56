import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from oracle import networks_ref as N, weights as W
from pcgan_amd.models import networks
from pcgan_amd.hip import functional as HF
from pcgan_amd.hip.nn import run_sequential
from pcgan_amd.hip.lib import ACT_RELU
def rl2(a, b): return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-300))
dev = torch.device('cuda:0')
ref32 = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
ref32.load_state_dict(W.fill_state_dict(ref32.state_dict(), 30))
ref64 = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
ref64.load_state_dict(ref32.state_dict()); ref64.double()
hip = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
hip.load_state_dict(ref32.state_dict()); hip.to(dev)
x0 = W.seeded_tensor((3, 3, 64, 64), 102)
dy = W.seeded_normal((3, 1, 1, 1), 302)
def staged(name):
    outs = []
    if name == 'hip':
        E = hip; x = x0.clone().to(dev).requires_grad_(True); m = E.base.model
        outs.append(x); h = m.conv1(x); outs.append(h); h = m.bn1(h, ACT_RELU); outs.append(h); h = m.maxpool(h); outs.append(h)
        for st in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in st:
                h = blk(h); outs.append(h)
        h = run_sequential(E.cnn, h); outs.append(h); y = HF.global_pool(h, False); outs.append(y)
        d = dy.to(dev)
    else:
        E = ref64 if name == 'c64' else ref32; dt = torch.float64 if name == 'c64' else torch.float32
        x = x0.clone().to(dt).requires_grad_(True); m = E.base.model
        outs.append(x); h = m.conv1(x); outs.append(h); h = m.relu(m.bn1(h)); outs.append(h); h = m.maxpool(h); outs.append(h)
        for st in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in st:
                h = blk(h); outs.append(h)
        h = E.cnn(h); outs.append(h); y = F.avg_pool2d(h, h.size(2)); outs.append(y)
        d = dy.to(dt)
    for o in outs[1:]: o.retain_grad()
    y.backward(d)
    return [(o.detach().cpu(), o.grad.detach().cpu()) for o in outs]
H, C32, C64 = staged('hip'), staged('c32'), staged('c64')
names = ['input', 'conv1', 'bn1relu', 'maxpool'] + ['blk%d' % i for i in range(8)] + ['cnn', 'pool']
for n, h, c, d in zip(names, H, C32, C64):
    print('%-8s %-18s fwd hip %.2e c32 %.2e | grad hip %.3e c32 %.3e (|g| %.2e)' % (n, tuple(h[0].shape), rl2(h[0], d[0]), rl2(c[0], d[0]), rl2(h[1], d[1]), rl2(c[1], d[1]), d[1].norm()))
