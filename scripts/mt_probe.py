import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import torch, bench
tmp = tempfile.mkdtemp()
model, opt = bench.build_model(0, 32, 128, tmp)
b = bench.synthetic_batch(32, 128, 0)
b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        model.set_input(b); model.optimize_parameters()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3
run(5)
for rep in range(3):
    for flag in (True, False):
        torch.autograd.set_multithreading_enabled(flag)
        run(2)
        w, h = run(12)
        print('autograd multithreading %-5s  %.2f ms/step (host %.2f)' % (flag, w, h), flush=True)
