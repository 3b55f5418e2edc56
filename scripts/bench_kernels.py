"""Kernel dashboard: times every distinct convolution of the config-2 step (fwd / dgrad / wgrad) with
HIP events and prints ms, TFLOP/s and the weighted contribution to one optimize_parameters().
Usage: python scripts/bench_kernels.py [filter-substring]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import ops

dev = torch.device('cuda:0')
DT = torch.bfloat16 if os.environ.get('BK_DTYPE') == 'bf16' else torch.float32
N = 32
# name, C, H, K, R, stride, pad, pad_mode, transposed, (n_fwd, n_dgrad, n_wgrad) per step, batch
L = [
    ('G.stem7x7 4->64@128', 4, 128, 64, 7, 1, 3, 1, False, (2, 1, 2)),
    ('G.down1 64->128@128', 64, 128, 128, 3, 2, 1, 0, False, (2, 2, 2)),
    ('G.down2 128->256@64', 128, 64, 256, 3, 2, 1, 0, False, (2, 2, 2)),
    ('G.res 256->256@32', 256, 32, 256, 3, 1, 1, 1, False, (36, 36, 36)),
    ('G.up1 T256->128@32', 128, 64, 256, 3, 2, 1, 0, True, (2, 2, 2)),
    ('G.up2 T128->64@64', 64, 128, 128, 3, 2, 1, 0, True, (2, 2, 2)),
    ('G.head7x7 64->3@128', 64, 128, 3, 7, 1, 3, 1, False, (2, 2, 2)),
    ('D.c0 4->64@128', 4, 128, 64, 4, 2, 1, 0, False, (4, 1, 3)),
    ('D.c1 64->128@64', 64, 64, 128, 4, 2, 1, 0, False, (4, 4, 3)),
    ('D.c2 128->256@32', 128, 32, 256, 4, 2, 1, 0, False, (4, 4, 3)),
    ('D.c3 256->512@16', 256, 16, 512, 4, 1, 1, 0, False, (4, 4, 3)),
    ('D.c4 512->1@15', 512, 15, 1, 4, 1, 1, 0, False, (4, 4, 3)),
    ('E.conv1 3->64@224', 3, 224, 64, 7, 2, 3, 0, False, (3, 1, 0)),
    ('E.l1 64->64@56', 64, 56, 64, 3, 1, 1, 0, False, (12, 4, 0)),
    ('E.l2a 64->128@56s2', 64, 56, 128, 3, 2, 1, 0, False, (3, 1, 0)),
    ('E.l2 128->128@28', 128, 28, 128, 3, 1, 1, 0, False, (9, 3, 0)),
    ('E.l3a 128->256@28s2', 128, 28, 256, 3, 2, 1, 0, False, (3, 1, 0)),
    ('E.l3 256->256@14', 256, 14, 256, 3, 1, 1, 0, False, (9, 3, 0)),
    ('E.l4a 256->512@14s2', 256, 14, 512, 3, 2, 1, 0, False, (3, 1, 0)),
    ('E.l4 512->512@7', 512, 7, 512, 3, 1, 1, 0, False, (9, 3, 0)),
    ('E.cnn0 512->32@7', 512, 7, 32, 3, 1, 1, 0, False, (3, 1, 0)),
    ('IP.c1 3->64@224', 3, 224, 64, 11, 4, 2, 0, False, (2, 1, 0)),
    ('IP.c2 64->192@27', 64, 27, 192, 5, 1, 2, 0, False, (2, 1, 0)),
    ('IP.c3 192->384@13', 192, 13, 384, 3, 1, 1, 0, False, (2, 1, 0)),
    ('IP.c4 384->256@13', 384, 13, 256, 3, 1, 1, 0, False, (2, 1, 0)),
    ('IP.c5 256->256@13', 256, 13, 256, 3, 1, 1, 0, False, (2, 1, 0)),
]


def timeit(fn, iters, reps=3):
    for _ in range(2):
        fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    return best


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    tot = 0.0
    print('%-24s %8s %8s %8s | %7s %7s %7s | %8s' % ('layer', 'fwd ms', 'dgrad', 'wgrad', 'fwd TF', 'dgr TF', 'wgr TF',
                                                     'ms/step'))
    for name, C, H, K, R, stride, pad, pm, tr, cnt in L:
        if flt not in name:
            continue
        P = (H + 2 * pad - R) // stride + 1
        x = (torch.rand(N, C, H, H, device=dev) * 2 - 1).to(DT)
        w = torch.randn(K, C, R, R, device=dev) * 0.05
        b = torch.zeros(K, device=dev)
        dy = torch.randn(N, K, P, P, device=dev).to(DT)
        if ops.HSPLIT and DT == torch.float32:     # operand maxima as their producers hand them over in the step (no absmax pass in the timings)
            ops._attach_amax(x, ops.amax_of(x))
            ops._attach_amax(dy, ops.amax_of(dy))
        flop = 2.0 * N * P * P * K * C * R * R
        iters = 10 if flop > 5e9 else 20
        cf, cb = {}, {}      # packed weights cached as in the step (and the step's routing)
        t_f = timeit(lambda: ops.conv2d_fwd(x, w, b, stride, pad, pm, pack_cache=cf), iters)
        t_d = timeit(lambda: ops.conv2d_bwd_data(dy, w, (H, H), stride, pad, pm, pack_cache=cb), iters)
        t_w = timeit(lambda: ops.conv2d_bwd_weight(x, dy, (K, C, R, R), stride, pad, pm), iters)
        if tr:   # ConvTranspose: its forward is the conv's dgrad and vice versa
            t_f, t_d = t_d, t_f
        step = cnt[0] * t_f + cnt[1] * t_d + cnt[2] * t_w
        tot += step
        print('%-24s %8.3f %8.3f %8.3f | %7.1f %7.1f %7.1f | %8.2f' % (
            name, t_f, t_d, t_w, flop / t_f / 1e9, flop / t_d / 1e9, flop / t_w / 1e9, step))
    print('sum of conv time per step: %.1f ms' % tot)


if __name__ == '__main__':
    main()
