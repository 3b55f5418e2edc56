"""host time per phase of ONE step issued from an idle GPU (no back-pressure of the launch queue): set_input / forward / update_G /
update_D, median over 15 steps; and the same with each discriminator pass counted (calls of netD.forward)."""
import sys, time, tempfile, contextlib, statistics, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
with contextlib.redirect_stdout(sys.stderr):
    model, opt = bench.build_model(0, bench.PER_GPU_BATCH, bench.SIZE, tempfile.mkdtemp(prefix='pcgan_hp_'))
batches = [bench.synthetic_batch(bench.PER_GPU_BATCH, bench.SIZE, 0, it) for it in range(2)]
batches = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in batches]
for i in range(6):
    model.set_input(batches[i % 2]); model.optimize_parameters()
torch.cuda.synchronize()
acc = {}
dfwd = []
orig = model.netD.forward
def dtimed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); dfwd.append(time.perf_counter() - t); return r
model.netD.forward = dtimed
def timed(name, f):
    t = time.perf_counter(); f(); acc.setdefault(name, []).append(time.perf_counter() - t)
for i in range(15):
    torch.cuda.synchronize()
    timed('set_input', lambda: model.set_input(batches[i % 2]))
    timed('forward', model.forward)
    timed('update_G', model.update_G)
    timed('update_D', model.update_D)
torch.cuda.synchronize()
tot = 0.0
for k, v in acc.items():
    m = statistics.median(v) * 1e3
    tot += m
    print('  %-10s %.3f ms' % (k, m))
print('  total      %.3f ms; one discriminator forward pass (host) %.3f ms x %d per step' % (tot, statistics.median(dfwd) * 1e3, len(dfwd) // 15))
