"""bf16 tensors: the residual weight gradient in the row-ring form (one product per tap, csrc/wgrad_rowring.hip: rowring_wgrad_bf16_kernel;
library option wgrad_rowring, default on) against the per-tap kernel: error of both against float64 (of the bf16 inputs), whole-tensor difference,
time per call.  usage: python scripts/time_rowring_bf16.py [N C H W]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgan_amd.hip import ops, lib as L
N, C, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 256, 32, 32)
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
x = torch.randn(N, C, H, W, generator=g).relu_().to(dev).to(torch.bfloat16)
dy = (torch.randn(N, C, H, W, generator=g) * 0.01).to(dev).to(torch.bfloat16)
lib = L.load()
L.set_option('wgrad_rowring', 1)
d = ops.make_desc(N, C, H, W, C, 3, 3, 1, 1, 1, ops.BF16)
assert lib.pcgan_conv2d_wgrad_rowring_supported(ctypes.byref(d))
ws = torch.empty(int(lib.pcgan_conv2d_wgrad_rowring_workspace_bytes(ctypes.byref(d))), dtype=torch.uint8, device=dev)
vp = ctypes.c_void_p


def ring(acc=0, out=None):
    dw = out if out is not None else torch.empty(C, C, 3, 3, device=dev)
    L.check(lib.pcgan_conv2d_bwd_weight_rowring(ctypes.byref(d), vp(x.data_ptr()), None, 0, vp(dy.data_ptr()), None, 0, vp(dw.data_ptr()), acc, vp(ws.data_ptr()),
                                                ws.numel(), vp(torch.cuda.current_stream().cuda_stream)), 'rowring')
    return dw


def hsplit():
    L.set_option('wgrad_rowring', 0)
    ops.clear_plans()
    try:
        return ops.conv2d_bwd_weight(x, dy, (C, C, 3, 3), 1, 1, 1)
    finally:
        L.set_option('wgrad_rowring', 1)
        ops.clear_plans()


a, b = ring(), hsplit()
torch.cuda.synchronize()
ks = [0, 1, C // 3, C - 1]
w = torch.zeros(len(ks), C, 3, 3, dtype=torch.float64, requires_grad=True)
xp = torch.nn.functional.pad(x.double().cpu(), (1, 1, 1, 1), mode='reflect')
for n0 in range(0, N, 8):
    torch.nn.functional.conv2d(xp[n0:n0 + 8], w).backward(dy[n0:n0 + 8, ks].double().cpu())
ref = w.grad
for name, t in (('rowring', a), ('per-tap', b)):
    print('%s: relative L2 error against float64 on channels %s: %.3e' % (name, ks, float((t[ks].double().cpu() - ref).norm() / ref.norm())))
print('rowring vs per-tap, whole tensor: relative L2 %.3e' % float((a.double() - b.double()).norm() / b.double().norm()))
for name, fn in (('rowring', ring), ('per-tap', hsplit)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print('%s: %.4f ms per call' % (name, e0.elapsed_time(e1) / 20))
