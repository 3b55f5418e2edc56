import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from pcgan_amd.hip import ops
torch.manual_seed(0)
dev = torch.device('cuda:0')
for (N, C, H, K, R, pad, pm, zc) in [(4, 4, 32, 8, 7, 3, 1, True), (4, 4, 32, 8, 7, 3, 1, False), (4, 4, 32, 8, 7, 3, 0, True), (4, 8, 32, 8, 3, 1, 1, False), (2, 4, 16, 8, 7, 3, 1, False)]:
    x = torch.rand(N, C, H, H) * 2 - 1
    if zc:
        x[:, -1] = torch.randn(N, 1, 1) * 0.5
    dy = torch.randn(N, K, H, H)
    dy = dy - dy.mean(dim=(2, 3), keepdim=True)
    def ref(dt):
        xx = x.to(dt); w = torch.zeros(K, C, R, R, dtype=dt, requires_grad=True)
        xp = F.pad(xx, (pad,) * 4, mode='reflect') if pm else xx
        y = F.conv2d(xp, w, padding=0 if pm else pad)
        y.backward(dy.to(dt)); return w.grad
    g64, g32 = ref(torch.float64), ref(torch.float32)
    g = ops.conv2d_bwd_weight(x.to(dev), dy.to(dev), (K, C, R, R), 1, pad, pm).cpu().double()
    def rl2(a, b): return float((a.double() - b).norm() / b.norm())
    err = (g - g64).abs()
    idx = err.argmax()
    print('N%d C%d H%d K%d R%d pm%d z%d: rel-L2 hip %.3e cpu32 %.3e  max err %.3e at %s (|g64|max %.3e)' % (
        N, C, H, K, R, pm, zc, rl2(g, g64), rl2(g32, g64), err.max(), tuple(torch.unravel_index(idx, err.shape)), g64.abs().max()))
    # error by input channel
    print('   per-channel rel-L2:', [round(rl2(g[:, c], g64[:, c]), 6) for c in range(C)])
