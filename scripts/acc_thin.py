import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcgan_amd.hip import ops
from oracle import ops_ref as R
dev='cuda:0'
g = torch.Generator().manual_seed(5)
for (N,C,H,K,k,st,pad,pm) in [(2,4,256,64,7,1,3,1),(2,4,128,64,7,1,3,1),(2,3,224,64,7,2,3,0)]:
    x = torch.randn(N,C,H,H,generator=g); w = torch.randn(K,C,k,k,generator=g)*0.05
    ref = R.conv2d(x.double(), w.double(), None, st, pad, pm)
    cpu32 = R.conv2d(x, w, None, st, pad, pm).double()
    out = {}
    for thin in (True, False):
        ops.THIN = thin
        y = ops.conv2d_fwd(x.to(dev), w.to(dev), None, st, pad, pm, pack_cache={}).double().cpu()
        out[thin] = y
        print((N,C,H,K,k,st), 'thin' if thin else 'fp32 MFMA', 'vs f64 %.3e' % float((y-ref).norm()/ref.norm()), 'vs CPU fp32 %.3e' % float((y-cpu32).norm()/ref.norm()),
              'max abs err / max %.3e' % float((y-ref).abs().max()/ref.abs().max()))
    print('   CPU fp32 vs f64 %.3e' % float((cpu32-ref).norm()/ref.norm()))
