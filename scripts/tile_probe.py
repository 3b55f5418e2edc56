"""forward / data-gradient time of a few hgemm layers under the current PCGAN_TILE setting"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import ops
dev = torch.device('cuda:0')
LAYERS = [('G.down1', 32, 64, 128, 128, 3, 2, 1), ('G.down2', 32, 128, 64, 256, 3, 2, 1), ('D.c1', 32, 64, 64, 128, 4, 2, 1), ('D.c2', 32, 128, 32, 256, 4, 2, 1),
          ('D.c3', 32, 256, 16, 512, 4, 1, 1), ('E.l1', 32, 64, 56, 64, 3, 1, 1), ('E.l2', 32, 128, 28, 128, 3, 1, 1), ('E.l3', 32, 256, 14, 256, 3, 1, 1),
          ('E.l4', 32, 512, 7, 512, 3, 1, 1), ('IP.c2', 32, 64, 27, 192, 5, 1, 2)]
def timeit(fn):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20)
    return best
out = []
for name, N, C, H, K, k, stride, pad in LAYERS:
    x = torch.rand(N, C, H, H, device=dev) * 2 - 1
    w = torch.randn(K, C, k, k, device=dev) * 0.05
    P = (H + 2 * pad - k) // stride + 1
    dy = torch.randn(N, K, P, P, device=dev)
    ops._attach_amax(x, ops.amax_of(x)); ops._attach_amax(dy, ops.amax_of(dy))
    cf, cb = {}, {}
    tf = timeit(lambda: ops.conv2d_fwd(x, w, None, stride, pad, 0, pack_cache=cf))
    tb = timeit(lambda: ops.conv2d_bwd_data(dy, w, (H, H), stride, pad, 0, pack_cache=cb))
    out.append('%s %.3f %.3f' % (name, tf, tb))
print(os.environ.get('PCGAN_TILE', 'default'), ' | '.join(out))
