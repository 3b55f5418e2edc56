"""host-side issue time vs GPU time of one optimize_parameters(): is the step launch-bound?"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device('cuda:0')
tmp = tempfile.mkdtemp()
model, opt = bench.build_model(0, 32, 128, tmp)
batches = [bench.synthetic_batch(32, 128, 0, it) for it in range(2)]
batches = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in batches]
for i in range(3):
    model.set_input(batches[i % 2]); model.optimize_parameters()
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for i in range(10):
    h0 = time.perf_counter()
    model.set_input(batches[i % 2]); model.optimize_parameters()
    host.append(time.perf_counter() - h0)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('host issue per step: %s ms' % ' '.join('%.1f' % (h * 1e3) for h in host))
print('issue total %.1f ms, wall incl. sync %.1f ms  (per step %.1f / %.1f)' % (t_issue * 1e3, t_all * 1e3, t_issue * 100, t_all * 100))
