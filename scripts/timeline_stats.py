"""GPU busy fraction and concurrency from a rocprofv3 --kernel-trace CSV (one row per kernel launch with start / end timestamps):
usage: timeline_stats.py <kernel_trace.csv> <steps to skip> <steps to analyse>.  Prints, for the steady part of the trace, the wall time,
the union of kernel intervals (GPU busy), the time with 1 / 2 / 3+ kernels in flight, the largest idle gaps and which kernels end
before / start after them."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda t: t[0])
t0, t1 = ks[0][0], max(k[1] for k in ks)
# steps are delimited by the discriminator's Adam launch (every second adam_kernel): skip argv[2] steps, analyse argv[3]
adam = [e for s, e, n in ks if 'adam_kernel' in n][1::2]
skip, take = int(sys.argv[2]), int(sys.argv[3])
lo, hi = adam[skip - 1], adam[skip - 1 + take]
print('steps %d .. %d of %d: %.3f ms per step' % (skip, skip + take, len(adam), (hi - lo) / take / 1e6))
ev = []
for s, e, n in ks:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, lo, {}
for t, d in ev:
    hist[min(depth, 4)] = hist.get(min(depth, 4), 0) + (t - last)
    last = t
    depth += d
hist[0] = hist.get(0, 0) + (hi - last)
wall = hi - lo
print('window %.1f ms' % (wall / 1e6))
for k in sorted(hist):
    print('  %s kernels in flight: %6.2f %%' % (('%d' % k) if k < 4 else '4+', 100.0 * hist[k] / wall))
# idle gaps
gaps, cur_end, last_name = [], lo, None
for s, e, n in ks:
    if e <= lo or s >= hi:
        continue
    if s > cur_end:
        gaps.append((s - cur_end, last_name, n))
    if e > cur_end:
        cur_end, last_name = e, n
gaps.sort(key=lambda g: -g[0])
print('idle gaps > 20 us: %d, total %.2f ms; > 5 us: %d, total %.2f ms; all: %d, total %.2f ms' % (
    sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6, sum(1 for g in gaps if g[0] > 5000),
    sum(g[0] for g in gaps if g[0] > 5000) / 1e6, len(gaps), sum(g[0] for g in gaps) / 1e6))
for g in gaps[:12]:
    print('  %7.1f us  after %-60s before %s' % (g[0] / 1e3, (g[1] or '')[:60], g[2][:60]))
