"""residual convolution (256->256 3x3 reflect): forward / data gradient on the three routes -- fp32 MFMA, bf16 x 6 split, fp16 x 3 split"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.hip import lib as L, ops
dev = torch.device('cuda:0')
NB, HH = int(os.environ.get('TB_N', 32)), int(os.environ.get('TB_H', 32))
GF = 2.0 * NB * HH * HH * 256 * 256 * 9 / 1e9
x = torch.randn(NB, 256, HH, HH, device=dev).relu_()
dy = torch.randn(NB, 256, HH, HH, device=dev)
w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
d = ops.make_desc(NB, 256, HH, HH, 256, 3, 3, 1, 1, 1)
y = torch.empty_like(x)
SL = lib.pcgan_absmax_slots(x.numel())
amax = torch.zeros(SL, device=dev)
pkf = torch.empty(lib.pcgan_conv2d_hsplit_packed_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
pkb = torch.empty(lib.pcgan_conv2d_hsplit_packed_bytes(ctypes.byref(d), 1), dtype=torch.uint8, device=dev)
L.check(lib.pcgan_conv2d_hsplit_pack(ctypes.byref(d), 0, w.data_ptr(), pkf.data_ptr(), st), 'pack')
L.check(lib.pcgan_conv2d_hsplit_pack(ctypes.byref(d), 1, w.data_ptr(), pkb.data_ptr(), st), 'pack')
L.check(lib.pcgan_absmax(x.data_ptr(), x.numel(), 0, amax.data_ptr(), SL, st), 'absmax')
def timeit(name, fn, flop=GF):
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
    print('%-28s %.4f ms  %.1f TFLOP/s' % (name, best, flop / best))
timeit('absmax (33.5 MB)', lambda: L.check(lib.pcgan_absmax(x.data_ptr(), x.numel(), 0, amax.data_ptr(), SL, st), 'absmax'), 0.0)
timeit('fwd fp16x3', lambda: L.check(lib.pcgan_conv2d_fwd_hsplit(ctypes.byref(d), x.data_ptr(), amax.data_ptr(), SL, pkf.data_ptr(), None, y.data_ptr(), 0, 0.0, st), 'f'))
L.check(lib.pcgan_absmax(dy.data_ptr(), dy.numel(), 0, amax.data_ptr(), SL, st), 'absmax')
timeit('dgrad fp16x3', lambda: L.check(lib.pcgan_conv2d_bwd_data_hsplit(ctypes.byref(d), dy.data_ptr(), amax.data_ptr(), SL, pkb.data_ptr(), y.data_ptr(), st), 'b'))
for hs, name in ((False, 'bf16x6'), (True, 'fp16x3 via ops (with absmax)')):
    ops.HSPLIT = hs
    cf, cb = {}, {}
    timeit('fwd ' + name, lambda: ops.conv2d_fwd(x, w, None, 1, 1, 1, pack_cache=cf))
    timeit('dgrad ' + name, lambda: ops.conv2d_bwd_data(dy, w, (HH, HH), 1, 1, 1, pack_cache=cb))
wsw = torch.empty(lib.pcgan_conv2d_hsplit_wgrad_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
dw = torch.empty(256, 256, 3, 3, device=dev)
xm = x.abs().amax(dim=(2, 3)).reshape(-1).contiguous()
dm = dy.abs().amax(dim=(2, 3)).reshape(-1).contiguous()
timeit('wgrad fp16x3 (pad+main+reduce)', lambda: L.check(lib.pcgan_conv2d_bwd_weight_hsplit(ctypes.byref(d), x.data_ptr(), xm.data_ptr(), xm.numel(), dy.data_ptr(), dm.data_ptr(), dm.numel(), dw.data_ptr(), 0, wsw.data_ptr(), wsw.numel(), st), 'w'))
ops.HSPLIT = False
timeit('wgrad bf16x6 (pad+pack+main+reduce)', lambda: ops.conv2d_bwd_weight(x, dy, (256, 256, 3, 3), 1, 1, 1))
