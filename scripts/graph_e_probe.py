"""probe: the frozen Elo encoder's forward at batch 8 (config 4's MC-dropout passes), eager launches vs hipGraph replay"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgan_amd.models import networks
dev = torch.device('cuda:0')
torch.manual_seed(0)
for bs in (8, 32):
    e = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7).to(dev)
    x = torch.rand(bs, 3, 224, 224, device=dev) * 2 - 1
    with torch.no_grad():
        for _ in range(3):
            e(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            e(x)
        torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / 30 * 1e3
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = e(x)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            g.replay()
        torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / 30 * 1e3
    print('encoder forward, batch %d: eager %.3f ms, hipGraph replay %.3f ms' % (bs, t_eager, t_graph), flush=True)
