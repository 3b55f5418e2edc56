"""Soak run of the benchmark step: N steps, device memory / host RSS / losses sampled every 100 -- caches keyed by tensors or shapes must
not grow with the step count.  Usage: python scripts/soak.py [steps]"""
import contextlib, os, resource, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pcgan_amd.hip import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
with contextlib.redirect_stdout(sys.stderr):
    model, opt = bench.build_model(0, bench.PER_GPU_BATCH, bench.SIZE, tempfile.mkdtemp(prefix='pcgan_soak_'))
batches = [bench.synthetic_batch(bench.PER_GPU_BATCH, bench.SIZE, 0, it) for it in range(4)]
batches = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in batches]
marks = []
for i in range(n):
    model.set_input(batches[i % 4])
    model.optimize_parameters()
    if (i + 1) % 100 == 0:
        torch.cuda.synchronize()
        ops.check_nonfinite()
        L = model.get_current_losses()
        assert all(v == v and abs(v) < 1e6 for v in L.values()), L
        marks.append((torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10))
        print('step %4d: device allocated %d MiB, reserved %d MiB, host max RSS %d MiB, G_GAN %.4f D_fake %.4f' % (
            i + 1, marks[-1][0], marks[-1][1], marks[-1][2], L['G_GAN'], L['D_fake']), flush=True)
assert marks[-1][0] <= marks[1][0] + 64 and marks[-1][1] <= marks[1][1] + 1024, 'device memory grows with the step count: %r' % marks
assert marks[-1][2] <= marks[1][2] + 256, 'host memory grows with the step count: %r' % marks
print('soak ok: %d steps' % n)
