"""Reproducer hunt: the head convolution's weight gradient (64 -> 3, 7x7, bs 32) on one stream while kernels of the Elo encoder's backward run
on another stream: is the result the same as alone?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
N = 32
x = torch.randn(N, 64, 128, 128, generator=g).relu_().to(dev)
dy = torch.randn(N, 3, 128, 128, generator=g).to(dev)
ref = ops.conv2d_bwd_weight(x, dy, (3, 64, 7, 7), 1, 3, 1).clone()
torch.cuda.synchronize()
sB = torch.cuda.Stream()
# partner work: data gradients of encoder-like layers (hgemm route), BN backward, max-pool backward
def mk(C, H, K, k, s, p):
    P = (H + 2 * p - k) // s + 1
    w = (torch.randn(K, C, k, k, generator=g) * 0.05).to(dev)
    d = torch.randn(N, K, P, P, generator=g).to(dev)
    return (w, d, H, s, p, {})
layers = [mk(64, 56, 64, 3, 1, 1), mk(64, 56, 128, 3, 2, 1), mk(128, 28, 128, 3, 1, 1), mk(256, 14, 256, 3, 1, 1), mk(512, 7, 512, 3, 1, 1), mk(3, 224, 64, 7, 2, 3)]
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
xr = torch.randn(32, 256, 32, 32, generator=g).to(dev)
wr = (torch.randn(256, 256, 3, 3, generator=g) * 0.05).to(dev)
cr = {}
big = torch.randn(64 * 1024 * 1024, device=dev)
def partner():
    if which == 'copy':
        for _ in range(6):
            big.clone()
        return
    if which == 'halo':
        for _ in range(4):
            ops.conv2d_fwd(xr, wr, None, 1, 1, 1, pack_cache=cr)
        return
    if which == 'wgrad_res':
        for _ in range(3):
            ops.conv2d_bwd_weight(xr, xr, (256, 256, 3, 3), 1, 1, 1)
        return
    for i, (w, d, H, s, p, cache) in enumerate(layers):
        if which != 'all' and which != str(i):
            continue
        for _ in range(3):
            ops.conv2d_bwd_data(d, w, (H, H), s, p, 0, pack_cache=cache)
partner(); torch.cuda.synchronize()
bad = 0
for trial in range(30):
    sB.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sB):
        partner()
    dw = ops.conv2d_bwd_weight(x, dy, (3, 64, 7, 7), 1, 3, 1)
    torch.cuda.synchronize()
    if not torch.equal(dw, ref):
        bad += 1
        d = (dw - ref).abs().view(3, -1)
        if bad <= 3:
            print('trial', trial, 'differs: per k', (d > 0).sum(dim=1).tolist(), 'max', float(d.max()))
print('partner', which, 'mismatching trials: %d / 30' % bad)
