import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn as nn
from oracle import networks_ref as N, weights as W
from pcgan_amd.models import resnet as R
from pcgan_amd.hip import nn as hnn
from pcgan_amd.hip.lib import ACT_RELU
def rl2(a, b): return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-300))
dev = torch.device('cuda:0')
ds_ref = nn.Sequential(nn.Conv2d(128, 256, 1, stride=2, bias=False), nn.BatchNorm2d(256))
ref = N.BasicBlockRef(128, 256, 2, ds_ref).double()
ref.load_state_dict({k: v.double() for k, v in W.fill_state_dict(ref.state_dict(), 5).items()})
ds_hip = nn.Sequential(hnn.Conv2d(128, 256, 1, stride=2, bias=False), hnn.BatchNorm2d(256))
hip = R.BasicBlock(128, 256, 2, ds_hip)
hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}); hip.to(dev)
x0 = torch.relu(W.seeded_normal((3, 128, 8, 8), 1))   # post-ReLU input with exact zeros
dy = W.seeded_normal((3, 256, 4, 4), 2)
# oracle staged
def run_ref():
    x = x0.double().requires_grad_(True)
    idt = ref.downsample[1](ref.downsample[0](x)); c1 = ref.conv1(x); b1 = torch.relu(ref.bn1(c1)); c2 = ref.conv2(b1); b2 = ref.bn2(c2)
    out = torch.relu(b2 + idt)
    ts = [x, c1, b1, c2, idt, out]
    for t in ts[1:]: t.retain_grad()
    out.backward(dy.double()); return [(t.detach(), t.grad) for t in ts]
def run_hip():
    x = x0.to(dev).requires_grad_(True)
    dsc = hip.downsample[0](x); idt = hip.downsample[1](dsc); c1 = hip.conv1(x); b1 = hip.bn1(c1, ACT_RELU); c2 = hip.conv2(b1)
    out = hip.bn2(c2, ACT_RELU, 0.0, idt)
    ts = [x, c1, b1, c2, idt, out]
    for t in ts[1:]: t.retain_grad()
    out.backward(dy.to(dev)); return [(t.detach().cpu(), t.grad.cpu()) for t in ts]
A, B = run_hip(), run_ref()
for n, a, b in zip(['x', 'conv1', 'bn1relu', 'conv2', 'identity', 'out'], A, B):
    print('%-9s fwd %.2e  grad %.3e' % (n, rl2(a[0], b[0]), rl2(a[1], b[1])))
