"""times the dominant convolution (256->256 3x3 reflect @32x32, bs 32): fwd / dgrad / wgrad, HIP events, packed weights."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import ops
dev = torch.device('cuda:0')
x = torch.rand(32, 256, 32, 32, device=dev) * 2 - 1
w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
b = torch.zeros(256, device=dev)
dy = torch.randn(32, 256, 32, 32, device=dev)
cache = {}
def t(fn, it=20):
    for _ in range(60): fn()   # ~20 ms: lets the clocks settle
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / it)
    return best
tag = ' '.join('%s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('PCGAN_'))
print(tag, 'fwd0 %.4f dgrad0 %.4f (zero pad) ' % (t(lambda: ops.conv2d_fwd(x, w, b, 1, 1, 0, pack_cache=cache)),
      t(lambda: ops.conv2d_bwd_data(dy, w, (32, 32), 1, 1, 0, pack_cache=cache))))
print(tag, 'fwd %.4f  dgrad %.4f  wgrad %.4f ms' % (
    t(lambda: ops.conv2d_fwd(x, w, b, 1, 1, 1, pack_cache=cache)),
    t(lambda: ops.conv2d_bwd_data(dy, w, (32, 32), 1, 1, 1, pack_cache=cache)),
    t(lambda: ops.conv2d_bwd_weight(x, dy, (256, 256, 3, 3), 1, 1, 1))))
