"""Host work per step: time to ISSUE one optimize_parameters() starting from an idle GPU (synchronize, then time the call without
waiting for the device), with the composite residual-block calls (one per block / one per run of blocks) on and off.  Inside a long run the HIP queue back-pressures the host
(about one step ahead of the GPU), so the in-region issue time of bench.py tracks the GPU time, not the host's work."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from pcgan_amd.hip import ops
tmp = tempfile.mkdtemp()
model, opt = bench.build_model(0, 32, 128, tmp)
bs = [bench.synthetic_batch(32, 128, 0, i) for i in range(2)]
bs = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in bs]      # uploaded by set_input, as in bench.py
def step(i):
    model.set_input(bs[i % 2]); model.optimize_parameters()
for comp, trunk in ((True, True), (True, False), (False, False), (True, True), (True, False)):
    ops.COMPOSITE, ops.TRUNK = comp, trunk
    c0 = dict(ops.COMPOSITE_STATS)
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    issue, total = [], []
    for i in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    issue.sort(); total.sort()
    print('composite=%d trunk=%d  host issue from idle: median %.2f ms (min %.2f)   step from idle: median %.2f ms   (trunk calls %d, block calls %d)' % (
        comp, trunk, issue[5], issue[0], total[5], ops.COMPOSITE_STATS.get('trunk_fwd', 0) - c0.get('trunk_fwd', 0), ops.COMPOSITE_STATS['fwd'] - c0['fwd']))
