"""Forces every implicit-GEMM tile shape in turn (PCGAN_TILE) over the kernel dashboard and prints, per layer and pass, the
default choice against the best forced one -- where choose_tile() leaves time on the table.  (Split-K layers ignore the
switch.)  Usage on the GPU box: python scripts/tile_sweep.py > gpurun_out/tile_sweep.log"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tiles = [None, '128,128', '128,64', '64,128', '64,64', '32,128']
res = {}
for t in tiles:
    env = dict(os.environ)
    if t:
        env['PCGAN_TILE'] = t
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'bench_kernels.py')], env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        if '|' not in line or line.startswith('layer'):
            continue
        name = line[:24].strip()
        f = line[24:].split('|')[0].split()
        res.setdefault(name, {})[t or 'default'] = (float(f[0]), float(f[1]))
    print('done', t, file=sys.stderr, flush=True)
print('%-24s %-6s %8s | %s' % ('layer', 'pass', 'default', '  '.join('%9s' % t for t in tiles[1:])))
for name, r in res.items():
    for i, pas in enumerate(('fwd', 'dgrad')):
        d = r['default'][i]
        best = min((r[t][i], t) for t in tiles[1:] if t in r)
        mark = '  <-- %s %.0f %%' % (best[1], 100 * (1 - best[0] / d)) if best[0] < 0.93 * d else ''
        print('%-24s %-6s %8.3f | %s%s' % (name, pas, d, '  '.join('%9.3f' % r[t][i] for t in tiles[1:] if t in r), mark))
