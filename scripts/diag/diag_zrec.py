import sys, os, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from test_gpu_step import build_hip_model
from test_oracle_golden import build_oracle_step, step_inputs
def rl2(a, b): return float((a.double().cpu() - b.double()).norm() / b.double().norm())
model, opt = build_hip_model('default', pathlib.Path(tempfile.mkdtemp()))
oracle = build_oracle_step('default'); twin = build_oracle_step('default')
for net in (twin.netG, twin.netD, twin.netE, twin.netIP): net.double()
A, B, label = step_inputs(0)
oracle.set_input(A, B, label); oracle.forward()
twin.set_input(A.double(), B.double(), label); twin.forward()
model.set_input({'A': A, 'B': B, 'label': torch.tensor(label), 'A_paths': ['a'] * 4, 'B_paths': ['b'] * 4}); model.forward()
res = {}
for name, m in (('hip', model), ('c32', oracle), ('c64', twin)):
    pred = m.netE(m.fake_B_E)
    if name == 'hip':
        loss = m.criterionRec(pred, m.y_B)
    else:
        loss = F.mse_loss(pred, m.y_B)
    g_fbE, = torch.autograd.grad(loss, m.fake_B_E, retain_graph=True)
    g_fb, = torch.autograd.grad(loss, m.fake_B, retain_graph=True)
    gw = torch.autograd.grad(loss, [p for p in m.netG.parameters()], retain_graph=True, allow_unused=True)
    res[name] = (pred.detach().cpu(), m.y_B.cpu(), float(loss), g_fbE.cpu(), g_fb.cpu(), gw[0].cpu(), gw[-2].cpu())
for i, nm in enumerate(['pred_y', 'y_B', 'loss', 'd fake_B_E', 'd fake_B', 'd G.model.1.weight', 'd G.model.26.weight']):
    if nm == 'loss':
        print(nm, res['hip'][i], res['c32'][i], res['c64'][i]); continue
    print('%-20s hip %.3e c32 %.3e' % (nm, rl2(res['hip'][i], res['c64'][i]), rl2(res['c32'][i], res['c64'][i])))
print('pred', res['hip'][0].flatten(), res['c64'][0].flatten())
print('y_B ', res['hip'][1].flatten(), res['c64'][1].flatten())
