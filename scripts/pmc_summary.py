"""Summarise a rocprofv3 --pmc results db: mean counter value per kernel.  usage: pmc_summary.py <db> [name-filter]"""
import collections
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
rows = c.execute('select kernel_name, counter_name, value, duration from counters_collection').fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for k, cn, v, d in rows:
    if flt in k:
        agg[k[:70]][cn].append(v)
        dur[k[:70]].append(d)
for k, v in agg.items():
    print(k, ' mean duration %.1f us' % (sum(dur[k]) / len(dur[k]) / 1e3))
    for cn, vals in sorted(v.items()):
        print('   %-28s n=%d mean=%.5g' % (cn, len(vals), sum(vals) / len(vals)))
