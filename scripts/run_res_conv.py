"""runs the dominant convolution (256->256 3x3 reflect @32x32, bs 32) fwd / dgrad / wgrad a few times on the DEFAULT route
(packed weights cached as in the step: the bf16-split kernels unless PCGAN_BF16X6=0) -- for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import ops
dev = torch.device('cuda:0')
x = torch.rand(32, 256, 32, 32, device=dev) * 2 - 1
w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
b = torch.zeros(256, device=dev)
dy = torch.randn(32, 256, 32, 32, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cf, cb = {}, {}
for _ in range(n):
    ops.conv2d_fwd(x, w, b, 1, 1, 1, pack_cache=cf)
    ops.conv2d_bwd_data(dy, w, (32, 32), 1, 1, 1, pack_cache=cb)
    ops.conv2d_bwd_weight(x, dy, (256, 256, 3, 3), 1, 1, 1)
torch.cuda.synchronize()
