"""Where do the batch uploads sit in the step?  usage: upload_timeline.py <dir with rocprofv3 csv output> [step index]
Reads *_kernel_trace.csv and *_memory_copy_trace.csv (rocprofv3 --kernel-trace --memory-copy-trace --output-format csv), cuts one
step (between two discriminator Adam launches) and prints every memory copy in it and, per stream / queue, the first and last kernel and
the busy time -- all relative to the start of the step."""
import csv, glob, os, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
mt = glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True)
rows = list(csv.DictReader(open(kt)))
print('kernel trace columns:', list(rows[0].keys()))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?')), r.get('Queue_Id', '?')) for r in rows), key=lambda t: t[0])
adam = [e for s, e, n, q, _ in ks if 'adam_kernel' in n][1::2]
i = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) // 2
lo, hi = adam[i - 1], adam[i]
print('step %d of %d: %.3f ms' % (i, len(adam), (hi - lo) / 1e6))
per = {}
for s, e, n, st, q in ks:
    if e <= lo or s >= hi:
        continue
    p = per.setdefault((st, q), [s, e, 0, 0, n, n])
    p[0] = min(p[0], s); p[1] = max(p[1], e); p[2] += min(e, hi) - max(s, lo); p[3] += 1
    if s == p[0]: p[4] = n
    p[5] = n
for (st, q), p in sorted(per.items(), key=lambda kv: kv[1][0]):
    print('  stream %-4s queue %-3s: %4d kernels, first at %8.3f ms, last ends %8.3f ms, busy %7.3f ms   first: %s | last: %s' % (
        st, q, p[3], (p[0] - lo) / 1e6, (p[1] - lo) / 1e6, p[2] / 1e6, p[4][:40], p[5][:40]))
if mt:
    mrows = list(csv.DictReader(open(mt[0])))
    print('memory copy columns:', list(mrows[0].keys()) if mrows else None)
    for r in mrows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e <= lo or s >= hi:
            continue
        print('  copy %-22s %10s bytes  stream %-4s start %8.3f ms  duration %8.1f us' % (r.get('Direction', '?'), r.get('Bytes', r.get('Size', '?')), r.get('Stream_Id', '?'),
                                                                                  (s - lo) / 1e6, (e - s) / 1e3))
# blit copies show up as kernels
for s, e, n, st, q in ks:
    if lo <= s < hi and ('copyBuffer' in n or 'fillBuffer' in n):
        print('  blit %-40s stream %-4s start %8.3f ms  duration %8.1f us' % (n[:40], st, (s - lo) / 1e6, (e - s) / 1e3))
