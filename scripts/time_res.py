import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import ops
dev = torch.device('cuda:0')
x = torch.rand(32, 256, 32, 32, device=dev) * 2 - 1
w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
b = torch.zeros(256, device=dev)
def t(fn, it=20):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / it)
    return best
print(os.environ.get('PCGAN_STAGGER'), 'fwd ms %.4f' % t(lambda: ops.conv2d_fwd(x, w, b, 1, 1, 1)))
