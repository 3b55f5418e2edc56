"""which torch ops still launch device copies / elementwise kernels inside one optimize_parameters()"""
import os, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device('cuda:0')
model, opt = bench.build_model(0, 32, 128, tempfile.mkdtemp())
b = bench.synthetic_batch(32, 128, 0)
b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
for _ in range(2):
    model.set_input(b); model.optimize_parameters()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    model.set_input(b); model.optimize_parameters()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::add_', 'aten::add', 'aten::mul', 'aten::contiguous', 'aten::clone', 'aten::fill_', 'aten::zero_',
                  'aten::sub', 'aten::div', 'aten::index_select', 'aten::sum', 'aten::expand'):
        st = [s for s in (e.stack or []) if 'pc-gan_amd' in s or 'pcgan_amd' in s]
        cnt[(e.name, str(e.input_shapes)[:60], st[0][-70:] if st else '?')] += 1
for k, v in cnt.most_common(40):
    print(v, k)
