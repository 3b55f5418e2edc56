import sys, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle import networks_ref as N, weights as W
from pcgan_amd.models import networks
from pcgan_amd.hip.lib import ACT_RELU
def rl2(a, b): return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-300))
dev = torch.device('cuda:0')
ref64 = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
ref64.load_state_dict(W.fill_state_dict(ref64.state_dict(), 30))
hip = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
hip.load_state_dict(ref64.state_dict()); hip.to(dev); ref64.double()
cap = {'hip': {}, 'ref': {}}
def hip_fwd(self, x):
    dsc = self.downsample[0](x); idt = self.downsample[1](dsc); c1 = self.conv1(x); b1 = self.bn1(c1, ACT_RELU); c2 = self.conv2(b1)
    out = self.bn2(c2, ACT_RELU, 0.0, idt)
    for n, t in (('x', x), ('c1', c1), ('b1', b1), ('c2', c2), ('idt', idt), ('dsc', dsc), ('out', out)):
        t.retain_grad(); cap['hip'][n] = t
    return out
def ref_fwd(self, x):
    dsc = self.downsample[0](x); idt = self.downsample[1](dsc); c1 = self.conv1(x); b1 = torch.relu(self.bn1(c1)); c2 = self.conv2(b1)
    out = torch.relu(self.bn2(c2) + idt)
    for n, t in (('x', x), ('c1', c1), ('b1', b1), ('c2', c2), ('idt', idt), ('dsc', dsc), ('out', out)):
        t.retain_grad(); cap['ref'][n] = t
    return out
hb = hip.base.model.layer3[0]; rb = ref64.base.model.layer3[0]
hb.forward = types.MethodType(hip_fwd, hb); rb.forward = types.MethodType(ref_fwd, rb)
x0 = W.seeded_tensor((3, 3, 64, 64), 102); dy = W.seeded_normal((3, 1, 1, 1), 302)
yh = hip(x0.to(dev)); yh.backward(dy.to(dev))
yr = ref64(x0.double()); yr.backward(dy.double())
for n in ('out', 'idt', 'dsc', 'c2', 'b1', 'c1', 'x'):
    print('%-4s fwd %.2e grad %.3e  |g|=%.3e' % (n, rl2(cap['hip'][n], cap['ref'][n]), rl2(cap['hip'][n].grad, cap['ref'][n].grad), float(cap['ref'][n].grad.norm())))
# recompute pieces from the HIP-side captured tensors with fp64 torch to localise
import torch.nn.functional as F
g = cap['hip']
gc1 = g['c1'].grad.double().cpu(); gds = g['dsc'].grad.double().cpu()
wx = rb.conv1.weight.detach(); wd = rb.downsample[0].weight.detach()
r1 = torch.nn.grad.conv2d_input(g['x'].shape, wx, gc1, stride=2, padding=1)
r2 = torch.nn.grad.conv2d_input(g['x'].shape, wd, gds, stride=2, padding=0)
print('x.grad vs recomputed-from-hip-intermediates', rl2(g['x'].grad, r1 + r2), ' conv1 part', float(r1.norm()), 'ds part', float(r2.norm()))
oh = cap['hip']['out'].detach().cpu().double(); orf = cap['ref']['out'].detach()
mh, mr = oh > 0, orf > 0
print('mask flips', int((mh != mr).sum()), 'of', mh.numel())
gi_h = cap['hip']['idt'].grad.cpu().double(); gi_r = cap['ref']['idt'].grad
d = (gi_h - gi_r).abs()
print('idt.grad: max abs err %.3e at value %.3e ; n(err>1e-6*max)=%d' % (float(d.max()), float(gi_r.flatten()[d.argmax()]), int((d > 1e-6 * gi_r.abs().max()).sum())))
go_h = cap['hip']['out'].grad.cpu().double(); go_r = cap['ref']['out'].grad
d2 = (go_h - go_r).abs(); print('out.grad max abs err %.3e (max %.3e)' % (float(d2.max()), float(go_r.abs().max())))
idx = d.flatten().topk(5).indices
for i in idx:
    print('  idx', int(i), 'out_hip %.3e out_ref %.3e  dy_hip %.4e dy_ref %.4e  gi_hip %.4e gi_ref %.4e' % (float(oh.flatten()[i]), float(orf.flatten()[i]), float(go_h.flatten()[i]), float(go_r.flatten()[i]), float(gi_h.flatten()[i]), float(gi_r.flatten()[i])))
