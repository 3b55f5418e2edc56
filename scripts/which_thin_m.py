import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcgan_amd.hip import ops
dev='cuda:0'
def t(name, f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    print('%-50s %.4f ms' % (name, e0.elapsed_time(e1)/n), flush=True)
# D.c0 dgrad 4 ch and E.conv1 dgrad
w = torch.randn(64,4,4,4,device=dev)*0.05; dy = torch.randn(32,64,64,64,device=dev); c={}
t('D.c0 dgrad (4 ch)', lambda: ops.conv2d_bwd_data(dy, w, (128,128), 2, 1, 0, pack_cache=c))
w3 = torch.randn(64,3,7,7,device=dev)*0.05; dy3 = torch.randn(32,64,112,112,device=dev); c3={}
t('E.conv1 dgrad (3 ch, 7x7 s2)', lambda: ops.conv2d_bwd_data(dy3, w3, (224,224), 2, 3, 0, pack_cache=c3))
wi = torch.randn(64,3,11,11,device=dev)*0.05; dyi = torch.randn(32,64,55,55,device=dev); ci={}
t('IP.c1 dgrad (3 ch, 11x11 s4)', lambda: ops.conv2d_bwd_data(dyi, wi, (224,224), 4, 2, 0, pack_cache=ci))
