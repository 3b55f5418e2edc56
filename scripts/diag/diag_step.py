"""diagnostic: per-quantity error of the HIP step vs the oracle step (not a test)."""
import sys, os, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_gpu_step import build_hip_model, _grab_grads, _rel_l2
from test_oracle_golden import build_oracle_step, step_inputs
variant = sys.argv[1] if len(sys.argv) > 1 else 'default'
tmp = pathlib.Path(tempfile.mkdtemp())
model, opt = build_hip_model(variant, tmp)
oracle = build_oracle_step(variant)
grabbed = _grab_grads(model)
for it in range(2):
    A, B, label = step_inputs(it)
    torch.manual_seed(1234 + it)
    oracle.set_input(A, B, label); oracle.optimize_parameters()
    model.set_input({'A': A, 'B': B, 'label': torch.tensor(label), 'A_paths': ['a'] * 4, 'B_paths': ['b'] * 4})
    model.optimize_parameters()
    print('--- it', it)
    for k in ('fake_B', 'rec_A', 'y_A', 'y_B'):
        a, b = getattr(model, k).detach().cpu().double(), getattr(oracle, k).detach().double()
        print('%-8s maxabs err %.3e  (max %.3e)' % (k, (a - b).abs().max(), b.abs().max()))
    gl, ol = model.get_current_losses(), oracle.losses()
    for n in gl:
        print('loss %-14s hip %.7f oracle %.7f diff %.2e' % (n, gl[n], ol[n], gl[n] - ol[n]))
    worst = sorted(((_rel_l2(grabbed['G'][k], g), k) for k, g in oracle.grads_G.items() if g is not None), reverse=True)[:6]
    print('worst G grads', worst)
    worst = sorted(((_rel_l2(grabbed['D'][k], g), k) for k, g in oracle.grads_D.items()), reverse=True)[:4]
    print('worst D grads', worst)
    with torch.no_grad():
        for hnet, onet in ((model.netG, oracle.netG), (model.netD, oracle.netD)):
            op = dict(onet.named_parameters())
            for k, hp in hnet.named_parameters():
                hp.copy_(op[k])
