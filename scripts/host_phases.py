"""where the host spends a steady-state step (wall-clock per call, GPU running behind): set_input / forward / update_G / update_D.  A phase
that takes about as long as the GPU needs for a step contains a host <-> device synchronisation (or the launch queue's back-pressure)."""
import sys, time, tempfile, contextlib, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda', 0)
with contextlib.redirect_stdout(sys.stderr):
    model, opt = bench.build_model(0, bench.PER_GPU_BATCH, bench.SIZE, tempfile.mkdtemp(prefix='pcgan_hp_'))
batches = [bench.synthetic_batch(bench.PER_GPU_BATCH, bench.SIZE, 0, it) for it in range(2)]
batches = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in batches]      # uploaded by set_input, as in bench.py
for i in range(6):
    model.set_input(batches[i % 2]); model.optimize_parameters()
torch.cuda.synchronize()
acc = {}
def timed(name, f):
    t = time.perf_counter(); f(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
n = 20
t0 = time.perf_counter()
for i in range(n):
    timed('set_input', lambda: model.set_input(batches[i % 2]))
    timed('forward', model.forward)
    timed('update_G', model.update_G)
    timed('update_D', model.update_D)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('step %.2f ms, host done issuing after %.2f ms per step' % (t_all / n * 1e3, t_issue / n * 1e3))
for k, v in acc.items():
    print('  %-10s %.3f ms per step' % (k, v / n * 1e3))
