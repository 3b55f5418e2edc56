"""diagnostic: per-layer gradient error of HIP and of the fp32 oracle vs the fp64 twin."""
import sys, os, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_gpu_step import build_hip_model, _grab_grads, _rel_l2
from test_oracle_golden import build_oracle_step, step_inputs
variant = 'default'
tmp = pathlib.Path(tempfile.mkdtemp())
lip, lz, la = os.environ.get('LIP', '1.0'), os.environ.get('LZ', '1.0'), os.environ.get('LA', '0.5')
model, opt = build_hip_model(variant, tmp, ['--lambda_IP', lip, '--lambda_z', lz, '--lambda_A', la])
oracle = build_oracle_step(variant)
twin = build_oracle_step(variant)
for o in (oracle, twin):
    o.opt.lambda_IP, o.opt.lambda_z, o.opt.lambda_A = float(lip), float(lz), float(la)
for net in (twin.netG, twin.netD, twin.netE, twin.netIP):
    net.double()
grabbed = _grab_grads(model)
A, B, label = step_inputs(0)
oracle.set_input(A, B, label); oracle.optimize_parameters()
twin.set_input(A.double(), B.double(), label); twin.optimize_parameters()
model.set_input({'A': A, 'B': B, 'label': torch.tensor(label), 'A_paths': ['a'] * 4, 'B_paths': ['b'] * 4})
model.optimize_parameters()
print('CONFIG lip', lip, 'lz', lz, 'la', la)
for k, g64 in twin.grads_G.items():
    if k in ('model.1.weight', 'model.14.conv_block.1.weight', 'model.26.weight'):
        hg, og = grabbed['G'][k].cpu(), oracle.grads_G[k]
        if k == 'model.1.weight':
            print('   z-slice: hip max %.3e  oracle max %.3e  g64 max %.3e' % (hg[:, 3].abs().max(), og[:, 3].abs().max(), g64[:, 3].abs().max()))
            for c in range(3):
                print('   ch%d: hip %.3e oracle %.3e' % (c, _rel_l2(hg[:, c], g64[:, c]), _rel_l2(og[:, c], g64[:, c])))
        print('%-28s |g64| %.3e  hip-vs-64 %.3e  cpu32-vs-64 %.3e  hip-vs-cpu32 %.3e' % (k, g64.norm(), _rel_l2(hg, g64), _rel_l2(og, g64), _rel_l2(hg, og)))
