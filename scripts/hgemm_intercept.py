"""fixed cost of an hgemm_kernel launch: a zero-padded 3x3 convolution (K = 256, 32x32, bs 32) at C = 16 .. 256 input channels -- time is linear in
the number of K stages (9 C / 16); the intercept is prologue + epilogue"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import ops
dev = torch.device('cuda:0')
res = []
for C in (16, 32, 64, 128, 256):
    x = torch.randn(32, C, 32, 32, device=dev).relu_()
    w = torch.randn(256, C, 3, 3, device=dev) * 0.02
    cf = {}
    fn = lambda: ops.conv2d_fwd(x, w, None, 1, 1, 0, pack_cache=cf)
    for _ in range(10): fn()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 50)
    res.append((C, best))
    print('C %4d  stages %4d  %.4f ms' % (C, 9 * C // 16, best))
(c0, t0), (c1, t1) = res[1], res[-1]
b = (t1 - t0) / (9 * (c1 - c0) / 16)
print('per stage %.3f us, intercept %.1f us (includes the absmax launch of x: ~4 us)' % (b * 1e3, (t0 - b * 9 * c0 / 16) * 1e3))
