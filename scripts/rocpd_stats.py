"""Per-kernel summary (the columns of rocprofv3's kernel_stats.csv) out of a rocpd database -- what
`rocprofv3 --kernel-trace --stats` writes on this image when no --output-format is given.
Usage: python scripts/rocpd_stats.py <results.db> > profiles/<name>.csv"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration), '
                   'avg(duration * duration) from kernels group by name order by sum(duration) desc').fetchall()
total = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
for name, calls, tot, avg, mn, mx, sq in rows:
    print('"%s",%d,%d,%.6f,%.2f,%d,%d,%.6f' % (name, calls, tot, avg, 100.0 * tot / total, mn, mx, max(sq - avg * avg, 0.0) ** 0.5))
