"""fixed (prologue + epilogue + launch) vs per-K cost of the chunked conv kernel: time C = 16..256 -> 256, 3x3 reflect @32x32 bs32"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import ops
dev = torch.device('cuda:0')
def t(fn, it=20):
    for _ in range(60): fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / it)
    return best
for bias in (True, False):
    for C in (16, 32, 64, 128, 256):
        x = torch.rand(32, C, 32, 32, device=dev) * 2 - 1
        w = torch.randn(256, C, 3, 3, device=dev) * 0.02
        b = torch.zeros(256, device=dev) if bias else None
        cache = {}
        ms = t(lambda: ops.conv2d_fwd(x, w, b, 1, 1, 1, pack_cache=cache))
        print('bias=%d C=%3d  %.4f ms  (pure MFMA at 155 TF: %.4f)' % (bias, C, ms, 2 * 32 * 1024 * 256 * C * 9 / 155e12 * 1e3))
