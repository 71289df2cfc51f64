#!/usr/bin/env python3
"""Average PMC counters per kernel from a rocprofv3 rocpd .db.  Usage: pmc_summary.py results.db [kernel-substring]"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); sub = sys.argv[2] if len(sys.argv) > 2 else ''
cols = [r[1] for r in con.execute("pragma table_info('counters_collection')")]
rows = con.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r[ix['kernel_name']] if 'kernel_name' in ix else r[ix['name']]
    if sub in name:
        agg[name.split('(')[0][:70]][r[ix['counter_name']]].append(r[ix['value']])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})')
