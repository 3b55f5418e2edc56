"""Tile / K-split sweep of the hgemm layers (every >= 16-channel convolution outside the residual trunk): forward and data gradient of
each layer under the library's heuristic and under every forced (tile, K split) -- options "hgemm_tile", "hgemm_ks" -- with HIP events.
Prints the heuristic's time, the best forced setting and what the step would gain.  Usage: python scripts/sweep_hgemm.py [filter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import torch
from pcgan_amd.hip import ops, lib
from bench_kernels import L, N, timeit, dev

TILES = (128128, 128064, 64128, 64064)
SPLITS = (1, 2, 4)


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    total_gain = 0.0
    for name, C, H, K, R, stride, pad, pm, tr, cnt in L:
        if flt not in name or C % 16 or K < 64 or name.startswith('G.res'):
            continue
        P = (H + 2 * pad - R) // stride + 1
        x = torch.rand(N, C, H, H, device=dev) * 2 - 1
        w = torch.randn(K, C, R, R, device=dev) * 0.05
        b = torch.zeros(K, device=dev)
        dy = torch.randn(N, K, P, P, device=dev)
        ops._attach_amax(x, ops.amax_of(x))
        ops._attach_amax(dy, ops.amax_of(dy))
        flop = 2.0 * N * P * P * K * C * R * R
        iters = 10 if flop > 5e9 else 20

        def run(which):
            cache = {}
            if which == 'fwd':
                return timeit(lambda: ops.conv2d_fwd(x, w, b, stride, pad, pm, pack_cache=cache), iters)
            return timeit(lambda: ops.conv2d_bwd_data(dy, w, (H, H), stride, pad, pm, pack_cache=cache), iters)
        for which, n_per_step in (('fwd', cnt[1] if tr else cnt[0]), ('dgrad', cnt[0] if tr else cnt[1])):
            lib.set_option('hgemm_tile', 0)
            lib.set_option('hgemm_ks', 0)
            ops.clear_plans()
            base = run(which)
            res = []
            for t in TILES:
                if K <= 64 and t // 1000 == 128:
                    continue
                for ks in SPLITS:
                    lib.set_option('hgemm_tile', t)
                    lib.set_option('hgemm_ks', ks)
                    ops.clear_plans()
                    res.append((run(which), t, ks))
            res.sort()
            bt, t, ks = res[0]
            gain = max(0.0, base - bt) * n_per_step
            total_gain += gain
            print('%-22s %-5s heuristic %.4f ms | best %.4f ms tile %dx%d ks %d (%+.1f %%) | x%d per step: %.3f ms | runners-up %s' % (
                name, which, base, bt, t // 1000, t % 1000, ks, 100 * (bt - base) / base, n_per_step, gain,
                ' '.join('%dx%d/%d:%.4f' % (tt // 1000, tt % 1000, kk, v) for v, tt, kk in res[1:3])), flush=True)
    lib.set_option('hgemm_tile', 0)
    lib.set_option('hgemm_ks', 0)
    print('sum of gains if every layer took its best setting: %.2f ms per step' % total_gain)


if __name__ == '__main__':
    main()
