import sys, os, tempfile, traceback, collections
sys.path.insert(0, os.getcwd())
import torch, bench
from pcgan_amd.hip import ops
tmp = tempfile.mkdtemp()
model, opt = bench.build_model(0, 32, 128, tmp)
b = bench.synthetic_batch(32, 128, 0)
b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
where = collections.Counter()
orig = ops.amax_of
def spy(x):
    ent = x.__dict__.get('_pcgan_amax')
    if ent is None or ent[0] != x._version:
        fn = x.grad_fn.name() if x.grad_fn is not None else 'nograd'
        st = traceback.extract_stack(limit=5)
        where[(tuple(x.shape), fn, st[-2].name)] += 1
    return orig(x)
ops.amax_of = spy
model.set_input(b); model.optimize_parameters()
where.clear()
model.set_input(b); model.optimize_parameters()
torch.cuda.synchronize()
print(ops.AMAX_STATS)
for k, v in sorted(where.items(), key=lambda kv: -kv[1] * kv[0][0][0] * kv[0][0][1] * kv[0][0][2] * kv[0][0][3]): print(v, k)
