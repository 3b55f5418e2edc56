"""Co-residency probe: the residual forward window kernel (160 VGPRs, 8 waves) on one stream and the residual weight gradient on another
-- with 8-wave workgroups (144 VGPRs: 2 x 160 + 2 x 144 > 512 per SIMD, the two never share a CU) or 4-wave workgroups
(PCGAN_WGRAD_BM=128, 152 VGPRs: 2 x 160 + 152 fits) -- timed alone, back to back on one stream, and concurrently.
usage: [PCGAN_WGRAD_BM=128] python scripts/corun_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import ops

dev = torch.device('cuda:0')
N, C, H = 32, 256, 32
x = torch.rand(N, C, H, H, device=dev) * 2 - 1
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
dy = torch.randn(N, C, H, H, device=dev)
ops._attach_amax(x, ops.amax_of(x))
ops._attach_amax(dy, ops.amax_of(dy))
cf = {}
dw = torch.zeros(C, C, 3, 3, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
IT = 40


def fwd():
    ops.conv2d_fwd(x, w, None, 1, 1, 1, pack_cache=cf)


def wg():
    ops.conv2d_bwd_weight(x, dy, (C, C, 3, 3), 1, 1, 1, accumulate_into=dw)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / IT


def alone(f):
    def run():
        for _ in range(IT):
            f()
    return run


def serial():
    for _ in range(IT):
        fwd()
        wg()


def concurrent():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        for _ in range(IT):
            fwd()
    with torch.cuda.stream(s2):
        for _ in range(IT):
            wg()
    cur.wait_stream(s1)
    cur.wait_stream(s2)


tf, tw, ts, tc = timed(alone(fwd)), timed(alone(wg)), timed(serial), timed(concurrent)
print('PCGAN_WGRAD_BM=%s  forward alone %.4f ms  weight gradient alone %.4f  back to back %.4f  concurrent %.4f  (sum of alone %.4f)'
      % (os.environ.get('PCGAN_WGRAD_BM', '-'), tf, tw, ts, tc, tf + tw))
