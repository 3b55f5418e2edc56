import sys, os, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from test_gpu_step import build_hip_model
from test_oracle_golden import build_oracle_step, step_inputs
from pcgan_amd.hip import functional as HF
from pcgan_amd.hip.nn import run_sequential
from pcgan_amd.hip.lib import ACT_RELU
def rl2(a, b): return float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-300))
model, opt = build_hip_model('default', pathlib.Path(tempfile.mkdtemp()))
twin = build_oracle_step('default'); o32 = build_oracle_step('default')
for net in (twin.netG, twin.netD, twin.netE, twin.netIP): net.double()
A, B, label = step_inputs(0)
model.set_input({'A': A, 'B': B, 'label': torch.tensor(label), 'A_paths': ['a'] * 4, 'B_paths': ['b'] * 4}); model.forward()
x0 = model.fake_B_E.detach().cpu()
tgt = model.y_B.detach().cpu()
def staged(name):
    outs = []
    if name == 'hip':
        E = model.netE; x = x0.cuda().requires_grad_(True); m = E.base.model
        outs.append(x)
        h = m.conv1(x); outs.append(h)
        h = m.bn1(h, ACT_RELU); outs.append(h)
        h = m.maxpool(h); outs.append(h)
        for st in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in st:
                h = blk(h); outs.append(h)
        h = run_sequential(E.cnn, h); outs.append(h)
        y = HF.global_pool(h, False); outs.append(y)
        loss = HF.mse_loss(y, tgt.cuda())
    else:
        E = twin.netE if name == 'c64' else o32.netE
        dt = torch.float64 if name == 'c64' else torch.float32
        x = x0.to(dt).requires_grad_(True); m = E.base.model
        outs.append(x)
        h = m.conv1(x); outs.append(h)
        h = m.relu(m.bn1(h)); outs.append(h)
        h = m.maxpool(h); outs.append(h)
        for st in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in st:
                h = blk(h); outs.append(h)
        h = E.cnn(h); outs.append(h)
        y = F.avg_pool2d(h, h.size(2)); outs.append(y)
        loss = F.mse_loss(y, tgt.to(dt))
    for o in outs[1:]:
        o.retain_grad()
    loss.backward()
    return [(o.detach().cpu(), (o.grad.detach().cpu() if o.grad is not None else torch.zeros_like(o).cpu())) for o in outs]
H, C32, C64 = staged('hip'), staged('c32'), staged('c64')
names = ['input', 'conv1', 'bn1relu', 'maxpool'] + ['blk%d' % i for i in range(8)] + ['cnn', 'pool']
for n, h, c, d in zip(names, H, C32, C64):
    print('%-8s shape %-18s fwd hip %.2e c32 %.2e | grad hip %.3e c32 %.3e  (|g|=%.3e)' % (n, tuple(h[0].shape), rl2(h[0], d[0]), rl2(c[0], d[0]), rl2(h[1], d[1]), rl2(c[1], d[1]), d[1].norm()))
