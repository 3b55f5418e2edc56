"""the <= 4-channel convolutions alone (G stem, E conv1, D c0 forward; G head data gradient): thin_conv.hip on the f16 matrix pipe against
igemm2_kernel<.., 4> on fp32 MFMA (PCGAN_THIN=0's route), HIP events, bs 32"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcgan_amd.hip import ops
dev = 'cuda:0'
LAYERS = [('G.stem 4->64 7x7 reflect @128', 32, 4, 128, 64, 7, 1, 3, 1, False),
          ('E.conv1 3->64 7x7 s2 @224', 32, 3, 224, 64, 7, 2, 3, 0, False),
          ('D.c0 4->64 4x4 s2 @128', 32, 4, 128, 64, 4, 2, 1, 0, False),
          ('G.head dgrad 64->3 7x7 reflect @128', 32, 64, 128, 3, 7, 1, 3, 1, True)]
for name, N, C, H, K, k, st, pad, pm, dgrad in LAYERS:
    P = (H + 2 * pad - k) // st + 1
    w = torch.randn(K, C, k, k, device=dev) * 0.05
    src = torch.randn(N, K, P, P, device=dev) if dgrad else torch.randn(N, C, H, H, device=dev)
    flop = 2.0 * N * P * P * K * C * k * k
    res = []
    for thin in (True, False):
        ops.THIN = thin
        cache = {}
        f = (lambda: ops.conv2d_bwd_data(src, w, (H, H), st, pad, pm, pack_cache=cache)) if dgrad else \
            (lambda: ops.conv2d_fwd(src, w, None, st, pad, pm, pack_cache=cache))
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    print('%-40s thin %.4f ms (%.0f TFLOP/s)   fp32 MFMA %.4f ms (%.0f)' % (name, res[0], flop / res[0] / 1e9, res[1], flop / res[1] / 1e9), flush=True)

print('weight gradients (hsplit_wgrad_kernel with <= 4 gathered channels / 49 taps against the fp32-MFMA kernels)')
for name, N, C, H, K, k, st, pad, pm in [('G.stem 4->64 7x7 reflect @128', 32, 4, 128, 64, 7, 1, 3, 1), ('D.c0 4->64 4x4 s2 @128', 32, 4, 128, 64, 4, 2, 1, 0),
                                         ('E.conv1 3->64 7x7 s2 @224', 32, 3, 224, 64, 7, 2, 3, 0)]:
    P = (H + 2 * pad - k) // st + 1
    x = torch.randn(N, C, H, H, device=dev)
    dy = torch.randn(N, K, P, P, device=dev)
    flop = 2.0 * N * P * P * K * C * k * k
    res, outs = [], []
    for thin in (True, False):
        ops.THIN_WGRAD = thin
        f = lambda: ops.conv2d_bwd_weight(x, dy, (K, C, k, k), st, pad, pm)
        for _ in range(3):
            o = f()
        outs.append(o.double())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    print('%-40s matrix pipe %.4f ms (%.0f TFLOP/s)   fp32 MFMA %.4f ms (%.0f)   rel diff %.2e  route %s' % (
        name, res[0], flop / res[0] / 1e9, res[1], flop / res[1] / 1e9, float((outs[0] - outs[1]).norm() / outs[1].norm()),
        [k_ for k_ in ops.ROUTE_STATS if k_[0] == 'wgrad']), flush=True)
