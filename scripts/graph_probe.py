"""probe: can a whole optimize_parameters() be captured in a HIP graph (torch.cuda.graph) and replayed?"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pcgan_amd.hip import ops
dev = torch.device('cuda:0')
tmp = tempfile.mkdtemp()
small = os.environ.get('GP_SMALL', '1') == '1'
if small:
    model, opt = bench.build_model(0, 4, 32, tmp, ngf=8, ndf=8, fine_e=64, n_blocks=2)
    b = bench.synthetic_batch(4, 32, 0)
else:
    model, opt = bench.build_model(0, 32, 128, tmp, dtype=os.environ.get('GP_DTYPE', 'fp32'))
    b = bench.synthetic_batch(32, 128, 0)
b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
# eager warm-up on a side stream (allocator, packed weights, lazily created streams)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        model.set_input(static); model.optimize_parameters()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print('eager losses', {k: round(v, 5) for k, v in model.get_current_losses().items()}, flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        model.set_input(static)
        model.optimize_parameters()
    print('captured', flush=True)
except Exception as e:
    print('CAPTURE FAILED:', type(e).__name__, str(e)[:2000], flush=True)
    raise
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print('replayed; losses', {k: round(float(getattr(model, 'loss_' + k)), 5) for k in model.loss_names}, flush=True)
n = 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    g.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('graph replay: %.2f ms/step (host issue %.2f ms)' % ((t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    model.set_input(static); model.optimize_parameters()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('eager: %.2f ms/step (host issue %.2f ms)' % ((t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3), flush=True)
