"""res-conv forward (256->256 3x3 reflect @32x32, bs 32): fp32 MFMA implicit GEMM vs the bf16-split experiment"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import lib as L, ops
dev = torch.device('cuda:0')
ops.BF16X6 = False      # the ops.* calls below are the fp32 MFMA reference; the bf16-split kernels are called directly
NB, HH = int(os.environ.get('TB_N', 32)), int(os.environ.get('TB_H', 32))     # TB_N=8 TB_H=64: the shape of configs 4 / 5 (256x256)
print('residual convolution 256->256 3x3 reflect, N=%d, %dx%d' % (NB, HH, HH))
GF = 2.0 * NB * HH * HH * 256 * 256 * 9 / 1e9
x = torch.randn(NB, 256, HH, HH, device=dev).relu_()
w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
b = torch.zeros(256, device=dev)
d = ops.make_desc(NB, 256, HH, HH, 256, 3, 3, 1, 1, 1)
lib = L.load()
pk = torch.empty(lib.pcgan_conv2d_bsplit_packed_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
L.check(lib.pcgan_conv2d_bsplit_pack(ctypes.byref(d), w.data_ptr(), pk.data_ptr(), st), 'pack')
y = torch.empty(NB, 256, HH, HH, device=dev)
cache = {}
def f6():
    L.check(lib.pcgan_conv2d_fwd_bsplit(ctypes.byref(d), x.data_ptr(), pk.data_ptr(), b.data_ptr(), y.data_ptr(), 0, 0.0, st), 'fwd')
def f32():
    ops.conv2d_fwd(x, w, b, 1, 1, 1, pack_cache=cache)
for name, fn in (('fp32 MFMA', f32), ('bf16x6   ', f6)):
    for _ in range(5):
        fn()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 50)
    print('%s %.4f ms  %.1f TFLOP/s fp32-equivalent' % (name, best, GF / best))
y32 = ops.conv2d_fwd(x, w, b, 1, 1, 1, pack_cache=cache)
f6(); torch.cuda.synchronize()
print('max |bf16x6 - fp32| = %.3e, rel L2 %.3e' % (float((y - y32).abs().max()), float((y - y32).norm() / y32.norm())))

# data gradient
dy = torch.randn(NB, 256, HH, HH, device=dev)
pkd = torch.empty(lib.pcgan_conv2d_bsplit_dgrad_packed_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
L.check(lib.pcgan_conv2d_bsplit_dgrad_pack(ctypes.byref(d), w.data_ptr(), pkd.data_ptr(), st), 'pack')
dx = torch.empty(NB, 256, HH, HH, device=dev)
cache2 = {}
def g6():
    L.check(lib.pcgan_conv2d_bwd_data_bsplit(ctypes.byref(d), dy.data_ptr(), pkd.data_ptr(), dx.data_ptr(), st), 'dgrad')
def g32():
    ops.conv2d_bwd_data(dy, w, (HH, HH), 1, 1, 1, pack_cache=cache2)
for name, fn in (('dgrad fp32 MFMA', g32), ('dgrad bf16x6   ', g6)):
    for _ in range(5):
        fn()
    best = 1e9
    for _ in range(3):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(50):
            fn()
        e_.record(); torch.cuda.synchronize()
        best = min(best, s_.elapsed_time(e_) / 50)
    print('%s %.4f ms  %.1f TFLOP/s fp32-equivalent' % (name, best, GF / best))

# weight gradient
wsw = torch.empty(lib.pcgan_conv2d_bsplit_wgrad_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
dw = torch.empty(256, 256, 3, 3, device=dev)
def h6():
    L.check(lib.pcgan_conv2d_bwd_weight_bsplit(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, wsw.data_ptr(), wsw.numel(), st), 'wgrad')
def h32():
    ops.conv2d_bwd_weight(x, dy, (256, 256, 3, 3), 1, 1, 1)
for name, fn in (('wgrad fp32 MFMA', h32), ('wgrad bf16x6   ', h6)):
    for _ in range(5):
        fn()
    best = 1e9
    for _ in range(3):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(50):
            fn()
        e_.record(); torch.cuda.synchronize()
        best = min(best, s_.elapsed_time(e_) / 50)
    print('%s %.4f ms  %.1f TFLOP/s fp32-equivalent' % (name, best, GF / best))
