"""One residual-convolution kernel (fwd | dgrad | wgrad | wgrad_pertap | hgemm | in_bwd) in a tight loop for ~3 s, time per launch printed at the end: run beside
rocm-smi sampling (scripts/power_kernels.sh) to see at which clock / power each kernel runs when it has the chip to itself for seconds
(the 20-repetition timings of the other scripts are bursts).  usage: python scripts/power_kernel_loop.py <kind> [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgan_amd.hip import ops, lib as L
kind = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
dev = torch.device('cuda:0')
N, C, H = 32, 256, 32
x = torch.randn(N, C, H, H, device=dev).relu_()
dy = torch.randn(N, C, H, H, device=dev) * 0.01
w = torch.randn(C, C, 3, 3, device=dev) * 0.02
ops._attach_amax(x, ops.amax_of(x))
ops._attach_amax(dy, ops.amax_of(dy))
cf, cb = {}, {}
if kind == 'wgrad_pertap':
    L.set_option('wgrad_rowring', 0)
    ops.clear_plans()
if kind == 'hgemm':          # G.down2: 128 -> 256, 3x3 stride 2 at 64x64
    x2 = torch.randn(N, 128, 64, 64, device=dev)
    w2 = torch.randn(256, 128, 3, 3, device=dev) * 0.02
    ops._attach_amax(x2, ops.amax_of(x2))
fn = {'fwd': lambda: ops.conv2d_fwd(x, w, None, 1, 1, 1, pack_cache=cf),
      'dgrad': lambda: ops.conv2d_bwd_data(dy, w, (H, H), 1, 1, 1, pack_cache=cb),
      'wgrad': lambda: ops.conv2d_bwd_weight(x, dy, (C, C, 3, 3), 1, 1, 1),
      'wgrad_pertap': lambda: ops.conv2d_bwd_weight(x, dy, (C, C, 3, 3), 1, 1, 1),
      'hgemm': lambda: ops.conv2d_fwd(x2, w2, None, 2, 1, 0, pack_cache=cf),
      'in_bwd': lambda: ops.instnorm_bwd(dy, x, x, *stats, 1e-5, 1, 0.0)}[kind]
if kind == 'in_bwd':
    y, mean, m2 = ops.instnorm_fwd(x, None, 1e-5, 1, 0.0)
    stats = (mean, m2)
for _ in range(20):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    n += 200
dt = time.perf_counter() - t0
print('%s: %d launches, %.4f ms per launch (sustained over %.1f s)' % (kind, n, dt / n * 1e3, dt))
