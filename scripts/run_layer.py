"""runs one convolution layer (forward / data gradient on the default route, packed weights cached) a few times -- for rocprofv3 --pmc passes.
usage: run_layer.py N C H K k stride pad reps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import ops
N, C, H, K, k, stride, pad, reps = [int(v) for v in sys.argv[1:9]]
dev = torch.device('cuda:0')
x = torch.rand(N, C, H, H, device=dev) * 2 - 1
w = torch.randn(K, C, k, k, device=dev) * 0.05
P = (H + 2 * pad - k) // stride + 1
dy = torch.randn(N, K, P, P, device=dev)
ops._attach_amax(x, ops.amax_of(x)); ops._attach_amax(dy, ops.amax_of(dy))
cf, cb = {}, {}
for _ in range(reps):
    ops.conv2d_fwd(x, w, None, stride, pad, 0, pack_cache=cf)
    ops.conv2d_bwd_data(dy, w, (H, H), stride, pad, 0, pack_cache=cb)
torch.cuda.synchronize()
