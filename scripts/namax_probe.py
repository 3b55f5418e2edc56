import ctypes, os, sys
sys.path.insert(0, '/root/repo')
import torch
from pcgan_amd.hip import lib as L, ops
dev = torch.device('cuda:0')
x = torch.randn(32, 256, 32, 32, device=dev).relu_()
w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
d = ops.make_desc(32, 256, 32, 32, 256, 3, 3, 1, 1, 1)
y = torch.empty_like(x)
pkf = torch.empty(lib.pcgan_conv2d_hsplit_packed_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
L.check(lib.pcgan_conv2d_hsplit_pack(ctypes.byref(d), 0, w.data_ptr(), pkf.data_ptr(), st), 'pack')
full = x.abs().amax(dim=(2, 3)).reshape(-1).contiguous()
for n in (8192, 1024, 64, 1, 8192, 1):
    am = full[:n].clone() if n > 1 else full.max().reshape(1).clone()
    if n > 1: am[0] = full.max()
    fn = lambda: L.check(lib.pcgan_conv2d_fwd_hsplit(ctypes.byref(d), x.data_ptr(), am.data_ptr(), n, pkf.data_ptr(), None, y.data_ptr(), 0, 0.0, st), 'f')
    for _ in range(10): fn()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 50)
    print('x_namax %5d  %.4f ms' % (n, best))
