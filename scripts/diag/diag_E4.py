import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from oracle import networks_ref as N, weights as W
from pcgan_amd.models import networks
from pcgan_amd.hip.lib import ACT_RELU
def rl2(a, b): return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-300))
dev = torch.device('cuda:0')
ref64 = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
ref64.load_state_dict(W.fill_state_dict(ref64.state_dict(), 30)); 
hip = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
hip.load_state_dict(ref64.state_dict()); hip.to(dev); ref64.double()
x0 = W.seeded_tensor((3, 3, 64, 64), 102)
# capture input of layer3[0] from the fp64 net
m = ref64.base.model
with torch.no_grad():
    h = m.maxpool(m.relu(m.bn1(m.conv1(x0.double())))); h = m.layer2(m.layer1(h))
xin = h.float()
print('input zeros frac', float((xin == 0).float().mean()), 'absmax', float(xin.abs().max()))
for which in ('layer3', 'layer2'):
    if which == 'layer2':
        with torch.no_grad():
            h = m.maxpool(m.relu(m.bn1(m.conv1(x0.double())))); h = m.layer1(h)
        xin = h.float()
    rb = getattr(m, which)[0]; hb = getattr(hip.base.model, which)[0]
    dy = W.seeded_normal(tuple(rb(xin.double()).shape), 77)
    def run_ref():
        x = xin.double().requires_grad_(True)
        idt = rb.downsample[1](rb.downsample[0](x)); c1 = rb.conv1(x); b1 = torch.relu(rb.bn1(c1)); c2 = rb.conv2(b1); b2 = rb.bn2(c2)
        out = torch.relu(b2 + idt); ts = [x, c1, b1, c2, idt, out]
        for t in ts[1:]: t.retain_grad()
        out.backward(dy.double()); return [(t.detach(), t.grad) for t in ts]
    def run_hip():
        x = xin.to(dev).requires_grad_(True)
        dsc = hb.downsample[0](x); idt = hb.downsample[1](dsc); c1 = hb.conv1(x); b1 = hb.bn1(c1, ACT_RELU); c2 = hb.conv2(b1)
        out = hb.bn2(c2, ACT_RELU, 0.0, idt); ts = [x, c1, b1, c2, idt, dsc, out]
        for t in ts[1:]: t.retain_grad()
        out.backward(dy.to(dev)); return [(t.detach().cpu(), t.grad.cpu()) for t in ts]
    A, B = run_hip(), run_ref()
    print(which)
    for n, a, b in zip(['x', 'conv1', 'bn1relu', 'conv2', 'identity'], A, B):
        print('  %-9s fwd %.2e  grad %.3e' % (n, rl2(a[0], b[0]), rl2(a[1], b[1])))
    # split x-grad into its two contributions
    from pcgan_amd.hip import ops
    g_c1 = ops.conv2d_bwd_data(A[1][1].to(dev).contiguous(), hb.conv1.weight.detach(), tuple(xin.shape[2:]), 2, 1, 0).cpu()
    g_ds = ops.conv2d_bwd_data(A[5][1].to(dev).contiguous(), hb.downsample[0].weight.detach(), tuple(xin.shape[2:]), 2, 0, 0).cpu()
    r_c1 = torch.nn.grad.conv2d_input(xin.shape, rb.conv1.weight.detach(), B[1][1], stride=2, padding=1)
    print('  conv1 dgrad alone %.3e' % rl2(g_c1, r_c1))
    print('  sum check %.3e' % rl2(g_c1 + g_ds, B[0][1]))
